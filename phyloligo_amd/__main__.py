import sys

from .phyloligo import main

main()
sys.exit(0)      # the reference always exits 0 (phyloligo.py:1075)

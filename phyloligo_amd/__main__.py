import os
import sys

from . import _lib

if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
    _lib.PREFER_NO_TORCH = True      # one process, one GPU: numpy + the host-pointer entry points of the C ABI are enough

from .phyloligo import main

main()
sys.exit(0)      # the reference always exits 0 (phyloligo.py:1075)

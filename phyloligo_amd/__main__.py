import os
import sys

from . import _lib, launch


def _gpus_asked(argv):
    """--gpus N / --gpus=N without argparse (the full parser lives behind imports the launcher should not need)"""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith("--gpus="):
            return a.split("=", 1)[1]
    return "1"


try:
    _n = int(_gpus_asked(sys.argv[1:]))
except ValueError:
    _n = 1                               # get_cmd() below reports the malformed value
if launch.needs_launcher(_n):
    # --gpus N > 1: this process only starts one rank per GPU (python -m torch.distributed.run ... -m phyloligo_amd <same
    # arguments>) and waits for them - the reference fans out to its own workers the same way (bin/phyloligo.py:386-390).
    # It has touched neither torch nor the HIP runtime.  A failed launch is reported; the reference's exit status 0 (:1075)
    # is kept for a job that ran.
    sys.exit(launch.spawn_ranks(_n, ["-m", "phyloligo_amd"], sys.argv[1:],
                                timeout_s=float(os.environ.get("PO_CLI_LAUNCH_TIMEOUT", "0")) or None))

if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
    _lib.PREFER_NO_TORCH = True      # one process, one GPU: numpy + the host-pointer entry points of the C ABI are enough

from .phyloligo import main  # noqa: E402

main()
sys.exit(0)      # the reference always exits 0 (phyloligo.py:1075)

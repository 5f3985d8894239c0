"""Start one process per GPU for the multi-GPU forms of the path (`bench.py --gpus N`, `python -m phyloligo_amd --gpus N`).

The reference starts its own workers: `Parallel(n_jobs=threads_max)` over `gen_even_slices`
(/root/reference/phylopackage/bin/phyloligo.py:386-390, :424) - the caller gives a number, the program fans out.  Here the
workers are ranks of `torch.distributed.run`, one per GPU.  The process that calls `spawn_ranks` is only a launcher: it must
not have imported torch or touched the HIP runtime (nothing in this module does), it starts the ranks as a FRESH child
process in its own process group, lets them write straight to its stdout / stderr (rank 0 prints the result), waits, kills
the whole group on timeout or when it is itself told to stop, and hands back the child's return code.  No exec, no restart of
a process that has initialised a GPU.
"""
import os
import signal
import socket
import subprocess
import sys


def needs_launcher(gpus):
    """True when this process was asked for several GPUs and is not already one of the ranks."""
    return int(gpus) > 1 and "WORLD_SIZE" not in os.environ


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_command(n_ranks, target, argv, port=None):
    """argv of the child: python -m torch.distributed.run ... <target> <argv>.  target = ["script.py"] or ["-m", "module"]."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port())] + list(target) + list(argv)


def spawn_ranks(n_ranks, target, argv, timeout_s=None, env=None):
    """Run `target argv` as n_ranks ranks; returns the launcher's exit status (124 after a timeout, as timeout(1) does)."""
    assert "torch" not in sys.modules, "the launcher process must not import torch (it would initialise the HIP runtime)"
    cmd = rank_command(n_ranks, target, argv)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    child_env.setdefault("OMP_NUM_THREADS", "1")                      # torchrun would set (and announce) it otherwise
    sys.stdout.flush()
    sys.stderr.flush()
    proc = subprocess.Popen(cmd, env=child_env, start_new_session=True)          # own process group: killable as a whole

    def stop(signum=signal.SIGTERM, grace=15.0):
        try:
            os.killpg(proc.pid, signum)
        except ProcessLookupError:
            return
        try:
            proc.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()

    def on_signal(signum, _frame):
        stop(signum)
        sys.exit(128 + signum)

    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        try:
            return proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            sys.stderr.write("phyloligo_amd.launch: %d ranks did not finish within %.0f s - killing the process group\n"
                             % (n_ranks, timeout_s))
            stop()
            return 124
    finally:
        for s, h in old.items():
            signal.signal(s, h)
        if proc.poll() is None:          # an exception on the way out must not leave ranks behind
            stop()

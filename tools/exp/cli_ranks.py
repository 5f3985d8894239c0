"""One-off: the multi-GPU CLI with 4 ranks sharing one GPU over gloo, files compared with the single-process run
(at most 6 processes may hold the GPU on a gpurun box: the reference run is a child process that exits first)."""
import os, subprocess, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
rng = np.random.default_rng(5)
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "asm.fa")
    with open(fa, "wb") as fh:
        for i in range(1203):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(300, 3000)))].tobytes()
            fh.write(b">c%d\n" % i + s + b"\n")
    env = dict(os.environ, PO_CLI_REHEARSAL="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=root)
    for metric, large in (("JSD", "None"), ("BC", "memmap")):
        args = ["-i", fa, "-k", "4", "-d", metric, "--method", "joblib", "--large", large]
        ref = os.path.join(tmp, "ref")
        subprocess.run([sys.executable, "-m", "phyloligo_amd"] + args + ["-o", ref], check=True, cwd=root, env=env, capture_output=True)
        for ranks in (4,):
            got = os.path.join(tmp, "got%d" % ranks)
            out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                                  "--master-port", str(29560 + ranks), "-m", "phyloligo_amd"] + args + ["-o", got], capture_output=True, text=True, timeout=600, cwd=root, env=env)
            same = out.returncode == 0 and open(ref, "rb").read() == open(got, "rb").read()
            print(metric, large, ranks, "ranks:", "identical" if same else "DIFFERENT rc=%d %s" % (out.returncode, out.stderr[-500:]), flush=True)

"""one-off: one seed of tests.test_gpu_fuzz_pairwise.test_random_problem (for a seed the long fuzz runs reported)"""
import sys
sys.path.insert(0, ".")
import phyloligo_amd as pa
from tests import test_gpu_fuzz_pairwise as t
ctx = pa.Context(0)
for seed in [int(a) for a in sys.argv[1:]]:
    try:
        t.test_random_problem(ctx, seed)
        print("seed", seed, "ok")
    except Exception as exc:      # noqa: BLE001
        print("seed", seed, "FAIL", str(exc)[:400].replace("\n", " | "))

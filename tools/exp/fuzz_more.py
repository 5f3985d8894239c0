"""one-off: the seeded fuzz tests of the GPU suite over MANY more seeds than the committed ranges (not part of the suite)"""
import sys, importlib
sys.path.insert(0, '.')
import numpy as np
import phyloligo_amd as pa
lo, hi = int(sys.argv[1]), int(sys.argv[2])
ctx = pa.Context(0)
fails = 0
for modname, fns in (("tests.test_gpu_fuzz_pairwise", None), ("tests.test_gpu_fuzz_blocks", None), ("tests.test_gpu_fuzz_counts", None)):
    mod = importlib.import_module(modname)
    for name in dir(mod):
        if not name.startswith("test_"):
            continue
        fn = getattr(mod, name)
        import inspect
        params = list(inspect.signature(fn).parameters)
        if params[:2] != ["ctx", "seed"] or len(params) != 2:
            continue
        import time
        t_grp = time.time()
        for seed in range(lo, hi):
            if (seed - lo) % 25 == 0:
                print("  %s seed %d (%.0f s so far)" % (name, seed, time.time() - t_grp), flush=True)
            try:
                t_one = time.time()
                fn(ctx, seed)
                if time.time() - t_one > 5:
                    print("  SLOW %s seed %d: %.1f s" % (name, seed, time.time() - t_one), flush=True)
            except Exception as exc:      # noqa: BLE001
                fails += 1
                print("FAIL %s.%s seed %d: %r" % (modname, name, seed, str(exc)[:300]), flush=True)
        print("%s.%s: seeds %d..%d done" % (modname, name, lo, hi - 1), flush=True)
print("failures:", fails, flush=True)

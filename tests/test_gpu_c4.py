"""BASELINE config 4 (200 000 contigs x 2 kb, seed 200001, k=4, -d JSD, 8 row blocks) on ONE GPU: the work lists of
rank 0 and rank 7 of the 8-rank plan are run with their real buffers (40 GB row slab + 17.6 GB of mirror blocks each,
64-bit indexing throughout) through po_pairwise_blocks_dev, exactly as bench.py --gpus 8 runs them on every rank.

Checked: (1) the plan evaluates every unordered pair exactly once (host arithmetic over all 8 ranks, and on the device:
every entry a rank must produce is written, nothing else is); (2) complete rows against the oracle - a row of R_0 is
assembled from rank 0's slab and rank 7's mirror block, i.e. across the two ranks; (3) each mirror block is the exact
transpose of the slab block it mirrors.  Reference shard shape: even row slices, /root/reference/phylopackage/bin/
phyloligo.py:424 (gen_even_slices) - the oracle side never reads that path at run time."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle
from phyloligo_amd import synthetic
from phyloligo_amd.dist import RowBlockPlan

N, WORLD, LENGTH = 200_000, 8, 2000


def test_c4_plan_covers_every_pair_once_host():
    plan = RowBlockPlan(N, WORLD)
    assert plan.bounds == [0, 25088, 50176, 75264, 100352, 125440, 150528, 175616, 200000]      # 128-aligned row blocks
    assert sum(plan.pair_evaluations(g) for g in range(WORLD)) == N * (N + 1) // 2
    # block-level exactly-once: every unordered pair of row blocks {a, b} is owned by one rank, or split into two
    # disjoint, complete halves by the two partners a, a + G/2 of an even world
    seen = {}
    for g in range(WORLD):
        for rows, cols, kind, peer in plan.work(g):
            seen.setdefault((g, g) if kind == "diag" else tuple(sorted((g, peer))), []).append((g, rows, cols, kind))
    assert len(seen) == WORLD * (WORLD + 1) // 2
    for (a, b), items in seen.items():
        if len(items) == 1:
            g, rows, cols, kind = items[0]
            assert kind in ("diag", "full") and rows == plan.rows(g) and cols == plan.rows(b if g == a else a)
            continue
        assert b == a + WORLD // 2 and len(items) == 2 and all(it[3] == "half" for it in items)
        low = next(it for it in items if it[0] == a)
        up = next(it for it in items if it[0] == b)
        assert low[2] == plan.rows(b) and up[1] == plan.rows(b)          # both span all of R_b
        assert low[1][0] == plan.rows(a)[0] and low[1][1] == up[2][0] and up[2][1] == plan.rows(a)[1]   # R_a = [lo,mid) + [mid,hi)


@pytest.fixture(scope="module")
def c4():
    import torch
    import phyloligo_amd as pa
    torch.cuda.empty_cache()                                   # what earlier tests left in torch's cache is not "used"
    free, total = torch.cuda.mem_get_info()
    if total < 100e9:
        pytest.skip("needs ~60 GB of HBM")
    ctx = pa.Context(0)
    seq, off = synthetic.contig_bytes(N, LENGTH, seed=synthetic.SEEDS["C4"])
    dseq = torch.from_numpy(seq).cuda()
    doff = torch.from_numpy(off.astype(np.int64)).cuda()
    counts, totals = ctx.count_profiles(dseq, doff, "1111", "both")
    del dseq
    freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
    yield ctx, counts, totals, freq
    ctx.close()


def _nan_count(t, chunk=2048):
    import torch
    return sum(int(torch.isnan(t[i:i + chunk]).sum()) for i in range(0, t.shape[0], chunk))


@pytest.mark.gpu
def test_c4_rank0_and_rank7_on_one_gpu(c4):
    import torch
    ctx, counts, totals, freq = c4
    plan = RowBlockPlan(N, WORLD)
    kept = {}
    for g in (7, 0):
        lo, hi = plan.rows(g)
        slab, mirrors = plan.allocate(g, counts.device, torch.float64)
        slab.fill_(float("nan"))
        for m in mirrors:
            if m is not None:
                m.fill_(float("nan"))
        st = plan.compute(ctx, counts, totals, "JSD", g, slab, mirrors, want_stats=True)
        torch.cuda.synchronize()
        assert st["kernel_id"] == 6 and st["rc_folded"]               # equal-total table kernel on folded operands
        # (1) written exactly where the plan says
        expect_written = 0
        for ((r0, r1), (c0, c1), kind, peer), m in zip(plan.work(g), mirrors):
            expect_written += (r1 - r0) * (c1 - c0)
            assert _nan_count(slab[r0 - lo:r1 - lo, c0:c1]) == 0
            if m is not None:
                assert _nan_count(m) == 0
                # (3) the mirror block is the exact transpose (sampled rows of the mirror: full 25 000-wide lines)
                for jj in (0, (c1 - c0) // 2, c1 - c0 - 1):
                    assert torch.equal(m[jj], slab[r0 - lo:r1 - lo, c0 + jj])
        assert slab.numel() - _nan_count(slab) == expect_written
        assert st["pairs"] * 2 >= plan.pair_evaluations(g)          # entries (with mirrors) / 2
        # diagonal block: symmetric, zero diagonal
        d = slab[:, lo:hi]
        assert bool((torch.diagonal(d) == 0).all())
        assert torch.equal(d[:4096, :4096], d[:4096, :4096].T)
        # (2) rows against the oracle, over every column this rank produced
        for i in (lo, lo + (hi - lo) // 2 + 3, hi - 1):
            want = oracle.pairwise_block(np.vstack([freq[i:i + 1], freq]), "JSD", 0, 1)[0, 1:]
            want[i] = 0.0
            got = slab[i - lo].cpu().numpy()
            mask = ~np.isnan(got)
            assert mask.sum() == sum((c1 - c0) for (r0, r1), (c0, c1), kind, peer in plan.work(g) if r0 <= i < r1)
            np.testing.assert_allclose(got[mask], want[mask], rtol=1e-6, atol=1e-12)
        if g == 7:       # keep rank 7's mirror of R_7 x R_0 (rows of R_0, columns R_7): completes rank 0's rows below
            for ((r0, r1), (c0, c1), kind, peer), m in zip(plan.work(7), mirrors):
                if peer == 0:
                    kept["m70"] = (m[:512].clone(), c0, r0, r1)      # first 512 rows of R_0 x all of R_7
        else:
            m70, c0, r0, r1 = kept["m70"]
            assert c0 == lo == 0
            # a row of R_0 completed across ranks: rank 0's slab + rank 7's mirror block, against the oracle
            for i in (0, 300, 511):
                want = oracle.pairwise_block(np.vstack([freq[i:i + 1], freq]), "JSD", 0, 1)[0, 1:]
                want[i] = 0.0
                row = slab[i].clone()
                assert bool(torch.isnan(row[r0:r1]).all())            # rank 0 did not evaluate R_0 x R_7 ...
                row[r0:r1] = m70[i]                                   # ... rank 7 did, once
                got = row.cpu().numpy()
                mask = ~np.isnan(got)
                np.testing.assert_allclose(got[mask], want[mask], rtol=1e-6, atol=1e-12)
                assert mask[:plan.bounds[4]].all() and mask[r0:r1].all()
        del slab, mirrors
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_config4_assembly_on_one_gpu_float32():
    """The north star's own size on ONE GPU: 200 000 contigs x 2 kb.  The float64 matrix (320 GB) does not fit, the float32 one
    (160 GB - the container type of --large memmap / h5py) does: JSD through the table kernel with float32 stores, 64-bit
    indexing over 4e10 entries.  Exact symmetry of far corners, zero diagonal, three complete rows against the oracle (float32
    rounding of a float64 result: rtol 1e-6), and the forced float64 matrix-core Euclidean on the same 160 GB buffer."""
    import torch
    import phyloligo_amd as pa
    from phyloligo_amd import synthetic
    from oracle import phyloligo_oracle as po
    n = 200_000
    torch.cuda.empty_cache()                                   # what earlier tests left in torch's cache is not "used"
    free, _ = torch.cuda.mem_get_info()
    if free < 175 * (1 << 30):
        pytest.skip("needs 175 GB of free HBM")
    seq, off = synthetic.contig_bytes(n, 2000, seed=synthetic.SEEDS["C4"])
    with pa.Context(0) as ctx:
        counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
        out = torch.empty((n, n), dtype=torch.float32, device="cuda")
        freq = None
        for metric, kw, kid in (("JSD", {}, 6), ("Eucl", {"table_path": False}, 3)):
            out.fill_(-1.0)
            _, st = ctx.pairwise(counts, totals, metric, dtype="float32", out=out, want_stats=True, **kw)
            assert st["kernel_id"] == kid and st["pairs"] == n * n // 2
            for a0, b0 in ((0, n - 4096), (100_000, 60_000), (n - 4096, 0)):
                assert torch.equal(out[a0:a0 + 4096, b0:b0 + 4096], out[b0:b0 + 4096, a0:a0 + 4096].T)
            assert bool((torch.diagonal(out) == 0).all()) and float(out.min()) >= 0.0
            if freq is None:
                freq = po.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
            for r in (0, 123_456, n - 1):
                want = po.pairwise_block(np.vstack([freq[r:r + 1], freq]), metric, 0, 1)[0, 1:]
                want[r] = 0.0
                np.testing.assert_allclose(out[r].cpu().numpy(), want.astype(np.float32), rtol=1e-6, atol=1e-7)
        del out
        torch.cuda.empty_cache()

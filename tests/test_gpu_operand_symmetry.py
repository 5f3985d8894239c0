"""Row blocks evaluated independently (no mirroring: what --large memmap does above its block budget, and what a
multi-GPU run's transposed blocks are compared with) must give the same bits for (i, j) and (j, i): every tile kernel
has to be exactly operand-symmetric, or "the multi-rank container equals the single-process one byte for byte" would
only hold below one row block.  Reference: per-pair metric calls are symmetric functions of their two
arguments up to floating-point order, /root/reference/phylopackage/core/phylodist.py:36-85."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def _profiles(ctx, ragged, pattern, n=650):
    rng = np.random.default_rng(11 if ragged else 12)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    lens = rng.integers(300, 6000, size=n) if ragged else np.full(n, 1800)
    seq = acgt[rng.integers(0, 4, size=int(lens.sum()))].copy()
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return ctx.count_profiles(seq, off, pattern, "both")


@pytest.mark.parametrize("metric", ["JSD", "BC", "Eucl", "KT", "SC"])
@pytest.mark.parametrize("ragged", [True, False])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_row_blocks_without_mirroring_are_exactly_symmetric(ctx, metric, ragged, dtype):
    pattern = "11011" if metric == "BC" else "1111"
    counts, totals = _profiles(ctx, ragged, pattern)
    n = counts.shape[0]
    full = np.empty((n, n), dtype=dtype)
    for lo in range(0, n, 128):                                   # independent row blocks, every (i, j) evaluated as asked
        hi = min(n, lo + 128)
        ctx.pairwise(counts, totals, metric, lo, hi, dtype=dtype, symmetric=False, out=full[lo:hi])
    assert np.array_equal(full, full.T, equal_nan=True)
    whole = ctx.pairwise(counts, totals, metric, dtype=dtype)     # one symmetric call: mirrored tiles
    assert np.array_equal(full, whole, equal_nan=True)


def test_memmap_container_in_row_blocks_equals_one_call(ctx, tmp_path, monkeypatch):
    """the container written block by block (forced here) and the one written from one symmetric call: same bytes"""
    from phyloligo_amd import phyloligo as P
    from phyloligo_amd import synthetic
    rng = np.random.default_rng(5)
    fa = tmp_path / "asm.fa"
    with open(fa, "wb") as fh:
        for i in range(700):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(400, 4000)))].tobytes()
            fh.write(b">c%d\n" % i + s + b"\n")
    for metric in ("JSD", "BC"):
        freq, _ = P.compute_frequencies("joblib", "None", str(fa), "1111", "both", 250, 4, ".")
        one, blk = tmp_path / ("one_%s.f32" % metric), tmp_path / ("blk_%s.f32" % metric)
        assert P.compute_distances("joblib", "memmap", freq, None, str(one), metric, 4, 250, ".") is None
        with monkeypatch.context() as m:
            m.setattr(P, "_row_chunk", lambda n, itemsize, budget=0: 128)
            assert P.compute_distances("joblib", "memmap", freq, None, str(blk), metric, 4, 250, ".") is None
        assert one.read_bytes() == blk.read_bytes(), metric

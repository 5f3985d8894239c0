"""Seeded fuzz of po_pairwise_blocks_dev (the multi-GPU work-list entry point): random rectangular blocks with and
without mirror buffers, triangular blocks, unaligned edges - every block must equal the corresponding part of the
full matrix bit for bit (a pair's arithmetic does not depend on the tile it falls into)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(30))
def test_random_block_lists(ctx, seed):
    import torch
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(2, 700))
    dim = int(rng.choice([4, 16, 64, 256]))
    metric = str(rng.choice(["Eucl", "JSD", "BC", "SC", "KT"]))
    top = int(rng.choice([3, 60, 300]))
    counts = rng.integers(0, top + 1, size=(n, dim)).astype(np.int32)
    if rng.random() < 0.5:                                   # some equal-total stretch so that both kernel families take part
        m = min(n, 300)
        target = int(counts[:m].sum(1).max())
        counts[:m, 0] += (target - counts[:m].sum(1)).astype(np.int32)
    totals = counts.sum(1).astype(np.int64)
    dc, dt = torch.from_numpy(counts).cuda(), torch.from_numpy(totals).cuda()
    dtype = torch.float64 if rng.random() < 0.7 else torch.float32
    full = ctx.pairwise(dc, dt, metric, dtype=dtype)
    blocks, checks = [], []
    for _ in range(int(rng.integers(1, 5))):
        if rng.random() < 0.3:
            r0 = int(rng.integers(0, n)); r1 = int(rng.integers(r0 + 1, n + 1))
            out = torch.full((r1 - r0, r1 - r0), float("nan"), dtype=dtype, device="cuda")
            blocks.append({"rows": (r0, r1), "cols": (r0, r1), "out": out, "triangular": True})
            checks.append((out, None, r0, r1, r0, r1))
        else:
            r0 = int(rng.integers(0, n)); r1 = int(rng.integers(r0 + 1, n + 1))
            c0 = int(rng.integers(0, n)); c1 = int(rng.integers(c0 + 1, n + 1))
            pad = int(rng.integers(0, 3)) * 2                  # leading dimension larger than the block
            out = torch.full((r1 - r0, c1 - c0 + pad), float("nan"), dtype=dtype, device="cuda")
            b = {"rows": (r0, r1), "cols": (c0, c1), "out": out}
            mir = None
            if rng.random() < 0.6:
                mir = torch.full((c1 - c0, r1 - r0 + pad), float("nan"), dtype=dtype, device="cuda")
                b["mirror"] = mir
            blocks.append(b)
            checks.append((out, mir, r0, r1, c0, c1))
    ctx.pairwise_blocks(dc, dt, metric, blocks, dtype=dtype)
    torch.cuda.synchronize()

    def same(a, b):
        return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))

    for out, mir, r0, r1, c0, c1 in checks:
        want = full[r0:r1, c0:c1]
        assert same(out[:, :c1 - c0], want), (metric, n, dim, (r0, r1, c0, c1))
        if mir is not None:
            assert same(mir[:, :r1 - r0], want.T.contiguous()), (metric, "mirror", (r0, r1, c0, c1))


@pytest.mark.parametrize("seed", range(6))
def test_random_block_lists_with_plane_classes(ctx, seed):
    """The same from 8 192 records on, where Eucl deals the tiles of every block to the one- / two- / three-plane kernels by the
    classes of their record blocks (host-read block maxima, device-built tile lists): random class patterns - a few 128-record
    blocks with counts to 5 000, to 500 000, or none at all - random blocks, float64 and float32, against the full matrix."""
    import torch
    rng = np.random.default_rng(9500 + seed)
    n = int(rng.integers(8192, 9700))
    dim = int(rng.choice([64, 256]))
    counts = rng.integers(0, 50, size=(n, dim)).astype(np.int32)
    for top, how_many in ((5000, int(rng.integers(0, 6))), (500_000, int(rng.integers(0, 3)))):
        for r in rng.integers(0, n, size=how_many):
            counts[r] = rng.integers(0, top, size=dim)
    if rng.random() < 0.3:
        counts[int(rng.integers(0, n))] = 0
    totals = counts.astype(np.int64).sum(1)
    dc, dt = torch.from_numpy(counts).cuda(), torch.from_numpy(totals).cuda()
    dtype = torch.float64 if seed % 2 == 0 else torch.float32
    full = torch.full((n, n), float("nan"), dtype=dtype, device="cuda")
    ctx.pairwise(dc, dt, "Eucl", out=full, dtype=dtype)
    assert not bool(torch.isnan(full).any()) and bool(torch.equal(full, full.T))
    blocks, checks = [], []
    for _ in range(3):
        r0 = int(rng.integers(0, n - 1)); r1 = int(rng.integers(r0 + 1, min(n, r0 + 3000) + 1))
        if rng.random() < 0.3:
            out = torch.full((r1 - r0, r1 - r0), float("nan"), dtype=dtype, device="cuda")
            blocks.append({"rows": (r0, r1), "cols": (r0, r1), "out": out, "triangular": True})
            checks.append((out, None, r0, r1, r0, r1))
        else:
            c0 = int(rng.integers(0, n - 1)); c1 = int(rng.integers(c0 + 1, min(n, c0 + 3000) + 1))
            pad = int(rng.integers(0, 5))
            out = torch.full((r1 - r0, c1 - c0 + pad), float("nan"), dtype=dtype, device="cuda")
            mir = torch.full((c1 - c0, r1 - r0 + pad), float("nan"), dtype=dtype, device="cuda") if rng.random() < 0.7 else None
            b = {"rows": (r0, r1), "cols": (c0, c1), "out": out}
            if mir is not None:
                b["mirror"] = mir
            blocks.append(b)
            checks.append((out, mir, r0, r1, c0, c1))
    ctx.pairwise_blocks(dc, dt, "Eucl", blocks, dtype=dtype)
    for out, mir, r0, r1, c0, c1 in checks:
        assert bool(torch.equal(out[:, :c1 - c0], full[r0:r1, c0:c1])), (seed, (r0, r1, c0, c1))
        assert bool(torch.isnan(out[:, c1 - c0:]).all())
        if mir is not None:
            assert bool(torch.equal(mir[:, :r1 - r0], full[c0:c1, r0:r1])), (seed, "mirror", (r0, r1, c0, c1))

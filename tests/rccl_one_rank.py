"""Child process of tests/test_gpu_rccl.py: the torch.distributed calls of the multi-GPU path on the `nccl` backend
(= RCCL) with ONE rank on cuda:0 -- the most a one-GPU box can exercise.  The process group is created before any other
GPU call; nothing is exec'ed afterwards.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    from phyloligo_amd.dist import RowBlockPlan
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    # (1) the single exchange of the path: int32 counts / int64 totals through all_gather_into_tensor
    n, dim = 1000, 256
    g = torch.Generator().manual_seed(3)
    counts = torch.randint(0, 40, (n, dim), dtype=torch.int32, generator=g).to(dev)
    totals = counts.sum(dim=1, dtype=torch.int64)
    plan = RowBlockPlan(n, 1)
    c2, t2 = plan.all_gather_profiles(counts, totals, dist, force=True)
    torch.cuda.synchronize()
    out["allgather_new_buffers"] = c2.data_ptr() != counts.data_ptr() and t2.data_ptr() != totals.data_ptr()
    out["allgather_equal"] = bool(torch.equal(c2, counts) and torch.equal(t2, totals))
    # (2) the row-completing exchange's primitive: batched isend / irecv of float64 row chunks, self-addressed
    src = torch.arange(64 * 300, dtype=torch.float64, device=dev).reshape(64, 300)
    dst = torch.full_like(src, float("nan"))
    ops = [dist.P2POp(dist.isend, src[:32], 0), dist.P2POp(dist.irecv, dst[:32], 0),
           dist.P2POp(dist.isend, src[32:], 0), dist.P2POp(dist.irecv, dst[32:], 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    out["p2p_equal"] = bool(torch.equal(src, dst))
    # complete_rows itself is a no-op for one rank (no peers) and must leave the slab untouched
    slab = torch.ones((n, n), dtype=torch.float64, device=dev)
    same = plan.complete_rows(0, slab, [None], dist)
    out["complete_rows_noop"] = bool(same.data_ptr() == slab.data_ptr() and float(slab.sum().item()) == n * n)
    # (3) the small collectives of bench.py
    t = torch.tensor([2.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one)
    lst = [torch.zeros(1, dtype=torch.float64, device=dev)]
    dist.all_gather(lst, t)
    dist.barrier()
    torch.cuda.synchronize()
    out["all_reduce"] = [float(t.item()), int(one.item()), float(lst[0].item())]
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

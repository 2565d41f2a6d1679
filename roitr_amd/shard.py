"""Pair sharding over the GPUs of one node.

Every pair is independent (the reference runs batch size 1 per forward, main.py:122-127, no cross-pair
state, read-only weights), so the partition is the one a DistributedSampler would make (main.py:106):
pair i -> rank i mod W.  No data-path collective exists; the only exchange is the gather of the small
per-rank result record to rank 0 over RCCL/xGMI (KB-scale: a direct exchange, not a ring reduction).
"""
import torch
import torch.distributed as dist


def pairs_for_rank(n_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes."""
    return list(range(rank, n_pairs, world))


def gather_counts(value):
    """All ranks contribute one integer (e.g. the number of correspondences found); every rank gets the list.
    Uses the default process group's backend: RCCL ('nccl') on GPUs, gloo in the CPU tests."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(value)]
    world = dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def gather_result_records(records):
    """Gather a per-rank list of small result records (pair_id, n_corr, scores tensor) to rank 0.

    records: list of (pair_id:int, scores: 1-D float tensor).  Padded to the longest record so that ONE
    all_gather carries everything.  Returns {pair_id: scores} on rank 0, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()):
        return {pid: s.detach().cpu() for pid, s in records}
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    n_local = torch.tensor([len(records), max([s.numel() for _, s in records] + [0])], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    max_rec = int(max(int(s[0]) for s in sizes))
    max_len = int(max(int(s[1]) for s in sizes))
    buf = torch.zeros((max_rec, max_len + 2), dtype=torch.float32, device=dev)
    for i, (pid, s) in enumerate(records):
        buf[i, 0] = float(pid)
        buf[i, 1] = float(s.numel())
        buf[i, 2:2 + s.numel()] = s.to(dev, torch.float32)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    if rank != 0:
        return None
    merged = {}
    for r in range(world):
        for i in range(int(sizes[r][0])):
            row = out[r][i].cpu()
            n = int(row[1].item())
            merged[int(row[0].item())] = row[2:2 + n].clone()
    return merged

"""The timed loop of bench.py and its cross-rank aggregation, separated from the script so that the N > 1 path -- pair
sharding, record ids, the one gather, max-over-ranks timing -- runs under `torch.distributed.run` on CPU (gloo, a stub
engine) in tests/test_shard_cpu.py exactly as it runs on 8 GPUs over RCCL.

Engine interface used here (roitr_amd.riga.RIGA_v2 provides it; the tests pass a stub):
    launch_batch(pairs, want_gt=True) -> handle          enqueue one batched forward, never waits for the device
    finish_batch(handle) -> [result dict per pair]        the one host synchronisation of the path ('corr_scores' is read)
    batch_records(handle, pair_ids) -> shard.RecordBatch  result records of a finished batch
"""
import sys
import time

import torch
import torch.distributed as dist

from .shard import assemble_block, block_words, gather_result_records


def record_id(step, slot, pairs_per_step, rank, world):
    """Global id of the pair in `slot` of `step` on `rank`: unique over steps, slots and ranks, and id % world == rank --
    the partition pair i -> rank i mod W (main.py:105-106 DistributedSampler) read backwards."""
    return (step * pairs_per_step + slot) * world + rank


def run_steps(model, batch, pairs_per_step, first, steps, rank=0, world=1, gather=True, scores_per_pair=1020, trace=False):
    """`steps` forwards, two batches in flight: batch s+1 is enqueued before the host unpacks batch s (launch_batch never waits
    for the GPU), so the device does not idle during the per-pair unpacking.  Every step's result records are packed (headers
    + one slice of the engine's scores); gather=True ends with the one collective of the path carrying all of them.

    batch(step) -> list of `pairs_per_step` forward() argument dicts.  Returns (correspondences found, GatheredRecords | None)."""
    B = pairs_per_step
    n_corr = 0
    handle = model.launch_batch(batch(first), want_gt=True)
    packed = []
    t_prev = time.perf_counter()
    for s in range(steps):
        nxt = model.launch_batch(batch(first + s + 1), want_gt=True) if s + 1 < steps else None
        t_l = time.perf_counter()
        res = model.finish_batch(handle)
        n_corr += sum(int(r["corr_scores"].shape[0]) for r in res)
        if trace:
            t_now = time.perf_counter()
            print(f"[bench trace] step {s}: launch {1e3 * (t_l - t_prev):.1f} ms, finish {1e3 * (t_now - t_l):.1f} ms", file=sys.stderr)
            t_prev = t_now
        if gather:
            packed.append(model.batch_records(handle, [record_id(s, j, B, rank, world) for j in range(B)]))
        handle = nxt
    recs = None
    if gather:
        slots = steps * B
        recs = gather_result_records(assemble_block(packed, slots, scores_per_pair), slots, scores_per_pair)
    return n_corr, recs


def aggregate(dt, n_corr, pairs_per_step, steps):
    """Whole-job numbers from the per-rank timed regions: time = MAX over ranks, work = the pairs of ALL ranks.
    Returns dict(dt, n_corr, total_pairs, value, world)."""
    world = 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt, float(n_corr)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, n_corr = float(tmax[0].item()), int(t[1].item())
    total = pairs_per_step * steps * world
    return {"dt": dt, "n_corr": int(n_corr), "total_pairs": total, "value": total / dt, "world": world}


def gather_summary(records, pairs_per_step, steps, scores_per_pair):
    """The `result_gather` object of the bench line (rank 0)."""
    return {"backend": {"nccl": "rccl"}.get(records.backend, records.backend), "rccl_ranks_seen": records.ranks_seen,
            "records": len(records), "scores": int(sum(records.n_scores.values())), "truncated_pairs": len(records.truncated),
            "record_bytes_per_rank": 4 * block_words(pairs_per_step * steps, scores_per_pair),
            "records_cover": f"every pair of all {steps} timed steps", "collectives": 1 if records.backend != "local" else 0}

#!/usr/bin/env python3
"""Headline benchmark: RoITr test-mode forward throughput in point-cloud pairs/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs-per-step B] [--n-points 5000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one engine forward over B independent synthetic 3DMatch-sized pairs (BASELINE.json
configs[1]: ~5000 points per cloud, fp32, 3DMatch test settings) -- the complete path of
model/RIGA_v2.py:58-175: FPS, kNN/PPF, local PPF-attention encoder/decoder, global geometric transformer,
partition, coarse matching, optimal transport, fine matching.  Inputs are resident in HBM before the timed
region.  Pairs shard over ranks with no data-path collective (weak scaling: every rank runs the same
per-step work on its own pairs); the only collective is the final gather of the per-rank correspondence
counts (the KB-scale result record of SURVEY.md 8e), outside the per-pair path.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle timed on this host, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--pairs-per-step", type=int, default=512)
    ap.add_argument("--n-points", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true", help="skip the one-pair-per-call measurement (profiling passes)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in roitr_amd)")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")  # RCCL on ROCm

    from roitr_amd.shard import gather_counts, pairs_for_rank
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device

    model = build_model("3DMatch")
    B, N = args.pairs_per_step, args.n_points
    # distinct resident pairs, cycled; pair ids are sharded over ranks exactly like the test loop would
    n_resident = max(B + B // 2, 16)
    ids = pairs_for_rank(n_resident * world, rank, world)
    pool = [pair_to_device(make_pair(N, config=2, pair_index=i)) for i in ids]

    def batch(step):
        return [pool[(step * B + j) % len(pool)] for j in range(B)]

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for s in range(args.warmup):
            model.forward_batch(batch(s), want_gt=True)
        barrier()
        import gc
        gc.collect()
        gc.disable()   # a generation-2 collection of the result dicts costs ~40 ms every dozen steps
        if not os.environ.get("ROITR_BENCH_NOPROF"):
            model.profile_reset()
        t0 = time.perf_counter()
        n_corr_total = 0
        # two batches in flight: batch s+1 is enqueued before the host unpacks batch s (launch_batch never waits for the
        # GPU), so the device does not idle during the per-pair unpacking; every step's full work is inside the region
        handle = model.launch_batch(batch(args.warmup), want_gt=True)
        for s in range(args.steps):
            nxt = model.launch_batch(batch(args.warmup + s + 1), want_gt=True) if s + 1 < args.steps else None
            res = model.finish_batch(handle)
            n_corr_total += sum(int(r["corr_scores"].shape[0]) for r in res)
            handle = nxt
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
    prof = model.profile_read()

    # the reference's own loop feeds ONE pair per forward (DataLoader batch_size 1, lib/tester.py:24-53): report that mode
    # too (rank 0, outside the timed region above), so the batched headline can be read against it
    single = None
    if rank == 0 and not distributed and not args.no_single_pair:
        with torch.no_grad():
            for s in range(3):
                model.forward_batch([pool[s]], want_gt=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n1 = 60
            h1 = model.launch_batch([pool[0]], want_gt=True)
            for s in range(n1):
                nx1 = model.launch_batch([pool[(s + 1) % len(pool)]], want_gt=True) if s + 1 < n1 else None
                model.finish_batch(h1)
                h1 = nx1
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
        single = {"pairs_per_step": 1, "pairs_per_s": round(n1 / d1, 2), "ms_per_pair": round(1e3 * d1 / n1, 3),
                  "note": "one pair per engine call, two calls in flight; bound by the ~800 dependent kernel dispatches of a forward (a HIP-graph replay of the same forward measures the same, scripts/bench_graph.py)"}

    # max over ranks of the timed region; total work = pairs of all ranks
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        counts = gather_counts(n_corr_total)
    else:
        counts = [n_corr_total]
    total_pairs = B * args.steps * world
    value = total_pairs / dt

    out = {
        "metric": "point-cloud pairs/s",
        "value": round(value, 3),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"3DMatch-sized synthetic pairs: {N} pts/cloud src+tgt, fp32, 3DMatch test settings "
                        f"(P=256 patches x 64 pts, 100 Sinkhorn iterations), full RIGA_v2 forward",
            "pairs_per_step": B,
            "n_points": N,
            "sharding": f"pairs over {world} rank(s), no data-path collective",
            "correspondences_found": int(sum(counts)),
        },
    }
    if rank == 0:
        out["roofline"] = roofline(prof, B, N)
        attach_traffic(out["roofline"], B)
        if out["roofline"] and "share_of_forward_time" in out["roofline"]:
            k = out["roofline"]["kernel"]
            out["roofline"]["share_of_forward_time"] = round(prof[k]["ms"] / args.steps / (1e3 * dt / args.steps), 4)
        out["kernel_ms_per_step"] = {k: round(v["ms"] / max(args.steps, 1), 4) for k, v in prof.items()}
        if single:
            out["single_pair_mode"] = single
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


def roofline(prof, B, N):
    """Dominant kernel family of the timed region (by summed HIP-event time, events on the launch stream) against
    its roofline.  gemm_kernel (every dense layer) is MFMA-bound: achieved = algorithmic FLOPs (2*M*N*K per launch) /
    kernel time vs the fp32-input MFMA peak.  The geometry kernels are priced on algorithmic HBM bytes
    (SURVEY.md 8d / DESIGN.md): FPS n->m: 12n + 4m + 8n (in/out `tmp`); kNN+PPF: 24R + 24M[queries != refs] + 20MK."""
    if not prof:
        return None
    name = max(prof, key=lambda k: prof[k]["ms"])
    p = prof[name]
    launches = max(p["launches"], 1)
    avg_ms = p["ms"] / launches
    per_launch = p["bytes"] / launches
    if name in ("gemm_kernel", "geo_embed_kernel"):
        achieved = p["bytes"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                "algorithmic_flops_per_launch": int(per_launch), "launches_timed": int(p["launches"]),
                "share_of_forward_time": None}
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
            "algorithmic_bytes_per_launch": int(per_launch), "launches_timed": int(p["launches"])}


def attach_traffic(roof, B):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE collected
    in their own rocprofv3 runs of this same command by scripts/collect_profiles.sh, gfx950 corrections applied in
    scripts/pmc_summary.py); only attached when the committed summary was taken at the same pairs-per-step."""
    if not roof:
        return
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return
    k = pmc.get("kernels", {}).get(roof.get("kernel"))
    if k and pmc.get("pairs_per_step") == B:
        roof["traffic"] = k["hbm_bytes_per_launch"]
        roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"


def cpu_baseline(N, budget_s):
    """The CPU oracle (oracle/, 'port' kind) on this host: full forward of ONE pair of the same workload."""
    try:
        from oracle import roitr_ref
    except Exception as e:  # oracle model restatement not available
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return roitr_ref.timed_baseline(N, budget_s)


if __name__ == "__main__":
    main()

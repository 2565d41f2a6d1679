#!/usr/bin/env python3
"""Headline benchmark: RoITr test-mode forward throughput in point-cloud pairs/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4] [--pairs-per-step B] [--n-points N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one engine forward over B independent synthetic pairs -- the complete path of model/RIGA_v2.py:58-175:
FPS, kNN/PPF, local PPF-attention encoder/decoder, global geometric transformer, partition, coarse matching, optimal
transport, fine matching, ground-truth side outputs.  Workloads (BASELINE.json `configs`):
    --config 2 (default)  3DMatch settings, 5000 pts/cloud, fp32            (configs[1], the config the metric is quoted on)
    --config 3            the same with the test-time rotation of dataset/tdmatch.py:99-112 (3DLoMatch rotated, configs[2])
    --config 4            4DMatch settings (factor 2, adaptive coarse matching, top-2), 8000 pts/cloud, bf16 operand
                          storage for the dense layers (configs[3])
`--pairs-per-step 1` times the reference's own one-pair-per-forward loop verbatim.
Inputs are resident in HBM before the timed region.  Pairs shard over ranks with no data-path collective (weak scaling:
every rank runs the same per-step work on its own pairs); the one collective of the path -- the gather of the per-pair
result records (match scores) to rank 0, shard.gather_result_records -- runs once at the end, INSIDE the timed region.

Timing: K steps between barrier + synchronize on both sides with the HIP-event instrumentation OFF -> `value`.
Rooflines: the same K steps are then run once more with the events of csrc/prof.cpp ON (events recorded on the launch
stream around each instrumented kernel) -> `roofline` (dominant kernel family) and `rooflines` (+ the two north-star
entries: kNN+PPF against HBM on algorithmic bytes, the global transformer phase against the MFMA peak).
Prints ONE JSON line (rank 0); `cpu_baseline` = the CPU oracle timed on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16

WORKLOADS = {
    2: dict(benchmark="3DMatch", n_points=5000, pairs=512, dtype="f32", seed_config=2,
            text="3DMatch-sized synthetic pairs: {N} pts/cloud src+tgt, fp32, 3DMatch test settings (P=256 patches x 64 pts, "
                 "100 Sinkhorn iterations), full RIGA_v2 forward"),
    3: dict(benchmark="3DLoMatch", n_points=5000, pairs=512, dtype="f32", seed_config=3,
            text="3DLoMatch-rotated synthetic pairs: {N} pts/cloud, seeded test-time SO(3) rotation of one cloud "
                 "(dataset/tdmatch.py:99-112), fp32, 3DMatch test settings, full RIGA_v2 forward"),
    4: dict(benchmark="4DMatch", n_points=8000, pairs=32, dtype="bf16", seed_config=4,
            text="4DMatch-sized synthetic pairs: {N} pts/cloud, 4DMatch test settings (factor 2 widths, adaptive coarse matching "
                 "min 128 / thr 0.75, top-2 fine matching), bf16 operand storage in the dense layers (fp32 accumulate; FPS / kNN / "
                 "PPF / OT in fp32), full RIGA_v2 forward"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--pairs-per-step", type=int, default=None)
    ap.add_argument("--n-points", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"], help="operand storage of the dense layers (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true", help="skip the one-pair-per-call measurement (profiling passes)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the instrumented repeat of the timed steps (no rooflines)")
    ap.add_argument("--no-rccl-selftest", action="store_true",
                    help="N=1 without torch.distributed.run: do NOT create the 1-rank RCCL group the result gather otherwise runs through")
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
    elif not args.no_rccl_selftest:
        # one GPU, no launcher: a 1-rank RCCL group, so that the path's collective goes through RCCL here as well
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        try:
            dist.init_process_group("nccl", rank=0, world_size=1)
        except Exception as e:   # e.g. the port is taken: the gather then stays local (reported as backend "local")
            print(f"[bench] 1-rank RCCL group not created: {e}", file=sys.stderr)

    from roitr_amd.harness import build_model, pair_to_device
    from roitr_amd.shard import gather_result_records, pairs_for_rank
    from roitr_amd.synthetic import make_pair

    wl = WORKLOADS[args.config]
    dtype = args.dtype or wl["dtype"]
    B = args.pairs_per_step or wl["pairs"]
    N = args.n_points or wl["n_points"]
    model = build_model(wl["benchmark"], operand_dtype=dtype)
    # distinct resident pairs, cycled; pair ids are sharded over ranks exactly like the test loop would
    n_resident = max(B + B // 2, 16)
    ids = pairs_for_rank(n_resident * world, rank, world)
    pool = [pair_to_device(make_pair(N, config=wl["seed_config"], pair_index=i)) for i in ids]

    def batch(step):
        return [pool[(step * B + j) % len(pool)] for j in range(B)]

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    max_scores = model.max_scores_per_pair()

    def run_steps(first, steps, gather):
        """`steps` forwards, two batches in flight: batch s+1 is enqueued before the host unpacks batch s (launch_batch never
        waits for the GPU), so the device does not idle during the per-pair unpacking.  gather: finish with the one collective
        of the path carrying the last step's records."""
        n_corr = 0
        handle = model.launch_batch(batch(first), want_gt=True)
        recs = None
        trace = os.environ.get("ROITR_BENCH_TRACE")
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
            if gather and s + 1 == steps:
                # unique slot ids for the record block: global pair id of the pool entry, made unique per slot of the step
                rec_ids = [rank + world * j for j in range(B)]
                block = model.batch_records(handle, rec_ids)
                recs = gather_result_records(block, B, max_scores)
            handle = nxt
        return n_corr, recs

    import gc
    with torch.no_grad():
        # warm-up with the SAME loop as the timed region (two batches in flight, the collective at the end): the caching
        # allocator then already owns both sets of output buffers, the RCCL communicator exists and the packing kernels are loaded
        run_steps(0, max(args.warmup, 1), gather=True)
        barrier()
        gc.collect()
        gc.disable()   # a generation-2 collection of the result dicts costs ~40 ms every dozen steps
        t0 = time.perf_counter()
        n_corr_total, records = run_steps(args.warmup, args.steps, gather=True)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()

        # ---- the same steps once more with the HIP-event instrumentation on (not part of `value`)
        prof, prof_steps = {}, 0
        if not args.no_profile_pass:
            prof_steps = args.steps
            gc.disable()
            model.profile_reset()
            run_steps(args.warmup, prof_steps, gather=False)
            torch.cuda.synchronize()
            gc.enable()
            prof = model.profile_read(kernels_only=False)

    # the reference's own loop feeds ONE pair per forward (DataLoader batch_size 1, lib/tester.py:24-53): report that mode
    # too (rank 0, outside the timed region above), so a batched headline can be read against it
    single = None
    if rank == 0 and not distributed and not args.no_single_pair and B != 1:
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
                  "note": "one pair per engine call (the reference's DataLoader batch size), two calls in flight"}

    # max over ranks of the timed region; total work = pairs of all ranks
    if distributed:
        t = torch.tensor([dt, float(n_corr_total)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0].item())
        n_corr_all = int(t[1].item())
    else:
        n_corr_all = n_corr_total
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
        "dtype": dtype,
        "data": "synthetic",
        "config": {
            "workload": wl["text"].format(N=N),
            "baseline_config": args.config,
            "pairs_per_step": B,
            "n_points": N,
            "sharding": f"pairs over {world} rank(s), no data-path collective; one gather of the result records at the end",
            "correspondences_found": int(n_corr_all),
        },
    }
    gt = model.geo_table_info()
    out["config"]["geometric_embedding"] = (
        "function table (csrc/geo_table.hip): degree-7 polynomial per channel on intervals of %g, %d distance + %d angle intervals, "
        "float64 fit error %.1e / %.1e of the amplitude" % (gt["interval"], gt["n_int_d"], gt["n_int_a"], gt["fit_d"] / max(gt["amp_d"], 1e-30),
                                                             gt["fit_a"] / max(gt["amp_a"], 1e-30))
        if gt else "fp32 MFMA GEMM form (geo_embed_kernel)")
    if rank == 0:
        if records is not None:
            out["result_gather"] = {"backend": {"nccl": "rccl"}.get(records.backend, records.backend), "rccl_ranks_seen": records.ranks_seen,
                                    "records": len(records), "scores": int(sum(records.n_scores.values())),
                                    "record_bytes_per_rank": int(B * (4 + max_scores) * 4), "collectives": 1 if records.backend != "local" else 0}
        roofs = rooflines(prof, prof_steps, dtype)
        attach_traffic(roofs, B, args.config)
        out["roofline"] = roofs[0] if roofs else None
        out["rooflines"] = roofs
        if prof_steps:
            out["kernel_ms_per_step"] = {k: round(v["ms"] / prof_steps, 4) for k, v in prof.items() if k != "geo_embed_reference_flops"}
            out["profile_pass"] = {"steps": prof_steps, "note": "the timed steps repeated with HIP events on; `value` is timed with them off"}
        if single:
            out["single_pair_mode"] = single
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_baseline_seconds, wl["benchmark"], wl["seed_config"])
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line goes out LAST: librccl prints its version banner through C stdio, which is flushed here first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


def rooflines(prof, steps, dtype):
    """[dominant kernel family, kNN+PPF vs HBM, global transformer phase vs MFMA] from the instrumented pass.

    gemm_kernel / geo_embed_kernel are MFMA-bound: achieved = algorithmic FLOPs (2*M*N*K per launch; 2*(1+k)*rows*C^2 for the
    embedding) / kernel time vs the MFMA peak of the operand dtype.  Geometry kernels are priced on algorithmic HBM bytes
    (SURVEY.md 8d / DESIGN.md): FPS n->m: 12n + 4m + 8n; kNN+PPF: 24R + 24M[queries != refs] + 20MK.  The global-transformer
    entry divides the FLOPs of every GEMM / embedding launch inside the phase by the WHOLE phase time (attention, softmax and
    LayerNorm kernels included)."""
    if not prof or not steps:
        return []
    mfma_peak = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    fwd_ms = prof.get("phase.forward", {}).get("ms", 0.0)

    def entry(name, label=None):
        p = prof[name]
        launches = max(p["launches"], 1)
        avg_ms = p["ms"] / launches
        per_launch = p["bytes"] / launches
        share = round(p["ms"] / fwd_ms, 4) if fwd_ms > 0 else None
        if name in ("gemm_kernel", "geo_embed_kernel") or name.startswith("phase."):
            achieved = p["bytes"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
            return {"bound": "mfma", "kernel": label or name, "achieved": round(achieved, 3), "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": round(achieved / mfma_peak, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                    "algorithmic_flops_per_launch": int(per_launch), "launches_timed": int(p["launches"]),
                    "share_of_forward_time": share}
        achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        return {"bound": "hbm", "kernel": label or name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                "algorithmic_bytes_per_launch": int(per_launch), "launches_timed": int(p["launches"]), "share_of_forward_time": share}

    kernels = {k: v for k, v in prof.items() if not k.startswith("phase.") and k != "geo_embed_reference_flops"}
    roofs = []
    if kernels:
        roofs.append(entry(max(kernels, key=lambda k: kernels[k]["ms"])))
    if "knn_query_kernel" in prof:
        roofs.append(entry("knn_query_kernel", "knn+ppf (every knn_*_kernel launch of the forward, PPF fused)"))
    if "phase.global_transformer" in prof:
        e = entry("phase.global_transformer", "global_transformer phase (FLOPs of the MFMA launches inside the phase over the whole phase time)")
        if "geo_embed_reference_flops" in prof:
            # the geometric embedding is evaluated from a function table (csrc/geo_table.hip), not by the reference's four
            # (rows, C) x (C, C) products: `achieved` above counts EXECUTED matrix work only; this is the same phase priced with
            # the FLOPs of the reference formulation (what the round-1 / GEMM-form numbers counted)
            p = prof["phase.global_transformer"]
            algo = (p["bytes"] + prof["geo_embed_reference_flops"]["bytes"]) / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
            e["achieved_reference_formulation"] = round(algo, 3)
            e["frac_reference_formulation"] = round(algo / mfma_peak, 5)
        roofs.append(e)
    if "geo_table_kernel" in prof:
        roofs.append(entry("geo_table_kernel", "geo_table_kernel (geometric embedding from the LDS function table; bytes = E written once + index rows)"))
    return roofs


def attach_traffic(roofs, B, config):
    """roofline.traffic: HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in their own rocprofv3
    runs of this same command by scripts/collect_profiles.sh, gfx950 corrections applied in scripts/pmc_summary.py); only
    attached when the committed summary was taken at the same workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return
    if pmc.get("pairs_per_step") != B or pmc.get("baseline_config", 2) != config:
        return
    src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    for roof in roofs:
        name = roof["kernel"].split(" ")[0]
        if name == "knn+ppf":   # the instrumented class spans every knn_*_kernel: launch-weighted mean over them
            ks = [v for n, v in pmc.get("kernels", {}).items() if n.startswith("knn_") and "replay" not in n]
            n = sum(v["launches"] for v in ks)
            if n:
                roof["traffic"] = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks) / n)
                roof["traffic_source"] = src
            continue
        k = pmc.get("kernels", {}).get(name)
        if k:
            roof["traffic"] = k["hbm_bytes_per_launch"]
            roof["traffic_source"] = src


def cpu_baseline(N, budget_s, benchmark, seed_config):
    """The CPU oracle (oracle/, 'port' kind) on this host: full forwards of pairs of the same workload."""
    try:
        from oracle import roitr_ref
    except Exception as e:  # oracle model restatement not available
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return roitr_ref.timed_baseline(N, budget_s, benchmark=benchmark, seed_config=seed_config)


if __name__ == "__main__":
    main()

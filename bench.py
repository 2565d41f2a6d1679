#!/usr/bin/env python3
"""Headline benchmark: RoITr test-mode forward throughput in point-cloud pairs/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--pairs-per-step B] [--n-points N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one engine forward over B independent synthetic pairs -- the complete path of model/RIGA_v2.py:58-175:
FPS, kNN/PPF, local PPF-attention encoder/decoder, global geometric transformer, partition, coarse matching, optimal
transport, fine matching, ground-truth side outputs.  Workloads (BASELINE.json `configs`):
    --config 2 (default)  3DMatch settings, 5000 pts/cloud, fp32            (configs[1], the config the metric is quoted on)
    --config 3            the same with the test-time rotation of dataset/tdmatch.py:99-112 (3DLoMatch rotated, configs[2])
    --config 4            4DMatch settings (factor 2, adaptive coarse matching, top-2), 8000 pts/cloud, bf16 operand
                          storage for the dense layers (configs[3])
    --config 5            kernel micro-benchmark of configs[4]: kNN(64 neighbours) + fused PPF on 32 clouds x 30000 points, one
                          call per step (roitr_amd.pointops.knn_ppf: grid build + query + PPF); its line reports queries/s, the
                          HBM fraction on algorithmic bytes and the VALU-issue fraction of the committed SQ pass
`--pairs-per-step 1` times the reference's own one-pair-per-forward loop verbatim.
Inputs are resident in HBM before the timed region.  Pairs shard over ranks with no data-path collective (weak scaling:
every rank runs the same per-step work on its own pairs); the result records (match scores) of EVERY timed step are packed
as the steps finish, and the one collective of the path -- their gather to rank 0, shard.gather_result_records -- runs once at
the end, INSIDE the timed region.  The loop and the cross-rank aggregation live in roitr_amd/benchloop.py (tested on CPU with
two gloo ranks and a stub engine).

Timing: K steps between barrier + synchronize on both sides with the HIP-event instrumentation OFF -> `value`.
Rooflines: the same K steps are then run once more with the events of csrc/prof.cpp ON (events recorded on the launch
stream around each instrumented kernel) -> `roofline` (dominant kernel family) and `rooflines` (+ the two north-star
entries: kNN+PPF against HBM on algorithmic bytes, the global transformer phase against HBM -- it is the stream of the
geometric embedding E -- with its executed matrix work against the MFMA peak beside it; + `whole_forward`).
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
    2: dict(benchmark="3DMatch", n_points=5000, pairs=512, dtype="f32", seed_config=2, weights="selective",
            text="3DMatch-sized synthetic pairs: {N} pts/cloud src+tgt, fp32, 3DMatch test settings (P=256 patches x 64 pts, "
                 "100 Sinkhorn iterations), full RIGA_v2 forward"),
    3: dict(benchmark="3DLoMatch", n_points=5000, pairs=512, dtype="f32", seed_config=3, weights="selective",
            text="3DLoMatch-rotated synthetic pairs: {N} pts/cloud, seeded test-time SO(3) rotation of one cloud "
                 "(dataset/tdmatch.py:99-112), fp32, 3DMatch test settings, full RIGA_v2 forward"),
    4: dict(benchmark="4DMatch", n_points=8000, pairs=64, dtype="bf16", seed_config=4, weights="selective", record_scores=12288,
            text="4DMatch-sized synthetic pairs: {N} pts/cloud, 4DMatch test settings (factor 2 widths, adaptive coarse matching "
                 "min 128 / thr 0.75, top-2 fine matching), bf16 operand storage in the dense layers (fp32 accumulate; FPS / kNN / "
                 "PPF / OT in fp32), full RIGA_v2 forward"),
    5: dict(benchmark="3DMatch", n_points=30000, pairs=32, dtype="f32", seed_config=5,
            text="kNN + PPF stress (BASELINE configs[4]): {B} synthetic clouds x {N} points, k = 64 neighbours (knnquery nsample 65, "
                 "column 0 dropped) + fused PPF, fp32 distances / int32 indices"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (one process per GPU).  Without a launcher (no WORLD_SIZE in the environment) and N > 1 the "
                         "script starts the N ranks itself through torch.distributed.run; under a launcher it must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--pairs-per-step", type=int, default=None)
    ap.add_argument("--n-points", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "f32x3"],
                    help="operand storage of the dense layers (default: the config's); f32x3 = fp32 with the K >= 256 linear layers on the bf16 "
                         "matrix cores by a three-way operand split (csrc/gemm_x3.hip)")
    ap.add_argument("--weights", default=None, choices=["plain", "selective"],
                    help="closed-form weight variant (roitr_amd/weights.py); default: selective -- descriptors that discriminate, thousands of "
                         "correspondences per pair (round 4; 'plain' ends in ~34 per pair, i.e. times the matching tail on near-empty outputs)")
    ap.add_argument("--cloud", default="uniform", choices=["uniform", "surface"],
                    help="synthetic geometry (roitr_amd/synthetic.py): points ~ U[0,2)^3, or room-like piecewise-planar surfaces with sensor noise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true", help="skip the one-pair-per-call measurement (profiling passes)")
    ap.add_argument("--no-sampling-ahead", action="store_true",
                    help="order the inputs of every forward on the main stream (the pool is resident, so by default the engine gets an "
                         "event instead and starts the first sampling level of call s+1 beside call s: RoitrForwardIO::inputs_ready)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the instrumented repeat of the timed steps (no rooflines)")
    ap.add_argument("--no-rccl-selftest", action="store_true",
                    help="N=1 without torch.distributed.run: do NOT create the 1-rank RCCL group the result gather otherwise runs through")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--record-scores-per-pair", type=int, default=None,
                    help="average score capacity per pair of the gathered result block (shard.py); the block carries every timed step "
                         "(default 4092; config 4: 12288 -- the adaptive matching emits ~10 k correspondences per pair)")
    return ap.parse_args()


def test_engine_hook():
    """ROITR_BENCH_TEST_ENGINE=/path/to/file.py (tests only): a module with `make(rank, world) -> (model, make_pool)` that stands in for
    the HIP engine on CPU tensors, so that the launcher / sharding / aggregation logic of THIS script runs where there is no GPU
    (tests/test_bench_launch_cpu.py; backend gloo).  The line it prints is marked `"data": "stub engine (test hook)"` and carries no
    roofline -- never a measurement.  Unset (always, outside tests/): the HIP engine, and no GPU is an error."""
    path = os.environ.get("ROITR_BENCH_TEST_ENGINE")
    if not path:
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("roitr_bench_test_engine", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks (one process per GPU) through
    torch.distributed.run on this node -- the same command line the driver's torchrun form uses -- and exit with its status.
    The reference does the same job with mp/torchrun around main.py:27-30 (init_process_group per local rank)."""
    import subprocess
    if not os.environ.get("ROITR_BENCH_TEST_ENGINE"):
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s); refusing to run a mislabeled job")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launched and args.gpus > 1:
        launch_ranks(args)          # does not return
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree "
                         "(n_gpus of the JSON line is the number of ranks that really ran)")
    distributed = world > 1
    hook = test_engine_hook()
    if hook is not None:
        args.test_engine = hook
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        out = forward_bench(args, rank, world, distributed)
        if distributed:
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    args.test_engine = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in roitr_amd)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")  # RCCL on ROCm
    elif not args.no_rccl_selftest and args.config != 5:
        # one GPU, no launcher: a 1-rank RCCL group, so that the path's collective goes through RCCL here as well
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        try:
            dist.init_process_group("nccl", rank=0, world_size=1)
        except Exception as e:   # e.g. the port is taken: the gather then stays local (reported as backend "local")
            print(f"[bench] 1-rank RCCL group not created: {e}", file=sys.stderr)

    if args.config == 5:
        out = knn_stress(args, rank, world, distributed)
    else:
        out = forward_bench(args, rank, world, distributed)
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


def forward_bench(args, rank, world, distributed):
    import gc

    import torch
    import torch.distributed as dist

    from roitr_amd import benchloop
    from roitr_amd.harness import build_model, pair_to_device
    from roitr_amd.shard import pairs_for_rank
    from roitr_amd.synthetic import make_pair

    wl = WORKLOADS[args.config]
    dtype = args.dtype or wl["dtype"]
    B = args.pairs_per_step or wl["pairs"]
    N = args.n_points or wl["n_points"]
    weights = args.weights or wl.get("weights", "plain")
    normals = "field" if weights == "selective" else "random"
    stub = args.test_engine is not None
    # distinct resident pairs, cycled; pair ids are sharded over ranks exactly like the test loop would
    n_resident = max(B + B // 2, 16)
    ids = pairs_for_rank(n_resident * world, rank, world)
    if stub:     # tests only (test_engine_hook): CPU stand-in for the engine, launcher / sharding / aggregation are the real ones
        model, make_pool = args.test_engine.make(rank, world)
        pool = make_pool(ids)
        args.no_profile_pass = args.no_single_pair = args.no_cpu_baseline = True
    else:
        model = build_model(wl["benchmark"], operand_dtype=dtype, weights=weights)
        pool = [pair_to_device(make_pair(N, config=wl["seed_config"], pair_index=i, normals=normals, cloud=args.cloud)) for i in ids]
        torch.cuda.synchronize()                                   # the pool is resident: nothing pending on any stream
        model.inputs_resident = not args.no_sampling_ahead
        model.weights_frozen = True                                # an inference loop: the weights were registered by build_model

    def batch(step):
        return [pool[(step * B + j) % len(pool)] for j in range(B)]

    def barrier():
        if distributed:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    spp = args.record_scores_per_pair or wl.get("record_scores", 4092)
    trace = bool(os.environ.get("ROITR_BENCH_TRACE"))
    with torch.no_grad():
        # warm-up with the SAME loop as the timed region (two batches in flight, the collective at the end): the caching
        # allocator then already owns both sets of output buffers, the RCCL communicator exists and the packing kernels are loaded
        benchloop.run_steps(model, batch, B, 0, max(args.warmup, 1), rank, world, True, spp)
        barrier()
        gc.collect()
        gc.disable()   # a generation-2 collection of the result dicts costs ~40 ms every dozen steps
        if not stub:
            model.host_ms.update(launch=0.0, unpack=0.0, calls=0)
        t0 = time.perf_counter()
        n_corr_total, records = benchloop.run_steps(model, batch, B, args.warmup, args.steps, rank, world, True, spp, trace)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        host_ms = None if stub else {k: round(v / max(args.steps, 1), 3) for k, v in model.host_ms.items() if k != "calls"}

        # ---- the same steps once more with the HIP-event instrumentation on (not part of `value`)
        prof, prof_steps, ot_counts = {}, 0, [0, 0, 0]
        if not args.no_profile_pass:
            prof_steps = args.steps
            gc.disable()
            model.profile_reset()
            ot_counts = ot_stats(enable=1)      # data-dependent work of the optimal-transport stage, counted over this pass only
            benchloop.run_steps(model, batch, B, args.warmup, prof_steps, rank, world, False, spp)
            torch.cuda.synchronize()
            gc.enable()
            prof = model.profile_read(kernels_only=False)
            ot_counts = ot_stats(enable=0)

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
            # and the latency of a call with nothing else in flight (launch, wait, unpack; no overlap between calls)
            t2 = time.perf_counter()
            n2 = 30
            for s in range(n2):
                model.forward_batch([pool[s % len(pool)]], want_gt=True)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
        single = {"pairs_per_step": 1, "pairs_per_s": round(n1 / d1, 2), "ms_per_pair": round(1e3 * d1 / n1, 3),
                  "ms_per_pair_one_call_in_flight": round(1e3 * d2 / n2, 3), "sampling_ahead": bool(model.inputs_resident),
                  "note": "one pair per engine call (the reference's DataLoader batch size), two calls in flight; "
                          "ms_per_pair_one_call_in_flight = latency of a call issued alone"}

    # max over ranks of the timed region; total work = pairs of all ranks
    agg = benchloop.aggregate(dt, n_corr_total, B, args.steps)
    dt = agg["dt"]

    out = {
        "metric": "point-cloud pairs/s",
        "value": round(agg["value"], 3),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "stub engine (test hook), not a measurement" if stub else "synthetic",
        "config": {
            "workload": wl["text"].format(N=N, B=B),
            "baseline_config": args.config,
            "pairs_per_step": B,
            "n_points": N,
            "weights": f"closed-form, variant '{weights}' (roitr_amd/weights.py); normals: {normals}",
            "cloud": args.cloud,
            "sharding": f"pairs over {world} rank(s), no data-path collective; one gather of the result records of all timed steps at the end",
            "correspondences_found": agg["n_corr"],
        },
    }
    gt = None if stub else model.geo_table_info()
    out["config"]["geometric_embedding"] = (
        "function table (csrc/geo_table.hip): degree-7 polynomial per channel on intervals of %g, %d distance + %d angle intervals, "
        "largest per-channel relative fit error %.1e / %.1e (gate 3.0e-08), %d bytes of LDS per workgroup"
        % (gt["interval"], gt["n_int_d"], gt["n_int_a"], gt["rel_d"], gt["rel_a"], gt["lds_bytes"])
        if gt else "fp32 MFMA GEMM form (geo_embed_kernel)")
    if gt:
        out["config"]["geometric_embedding_table"] = {"interval": gt["interval"], "lds_bytes": gt["lds_bytes"], "rel_fit_d": gt["rel_d"], "rel_fit_a": gt["rel_a"]}
    if host_ms:
        # host-side Python per step on THIS rank inside the timed region: launch = packing + allocation + the engine call (its ~350 kernel
        # launches), unpack = finish_batch after the device answered; both run while the device works on the other batch in flight
        out["host_ms_per_step"] = host_ms
    if rank == 0:
        if records is not None:
            out["result_gather"] = benchloop.gather_summary(records, B, args.steps, spp)
        roofs = rooflines(prof, prof_steps, dtype)
        pmc = attach_traffic(roofs, B, args.config)
        out["roofline"] = roofs[0] if roofs else None
        out["rooflines"] = roofs
        if prof_steps:
            out["whole_forward"] = whole_forward(prof, prof_steps, dtype, out["ms_per_step"], pmc)
            out["kernel_ms_per_step"] = {k: round(v["ms"] / prof_steps, 4) for k, v in prof.items() if k != "geo_embed_reference_flops"}
            out["profile_pass"] = {"steps": prof_steps, "note": "the timed steps repeated with HIP events on; `value` is timed with them off"}
            live, skipped, logdom = ot_counts
            # what the data-dependent tail of the forward actually did: patches that went through Sinkhorn, the share of
            # their 100 iterations skipped by the bit-exact fixed-point exit, patches the exponential form handed to the log-domain kernel
            out["optimal_transport"] = {"live_patches_per_step": round(live / prof_steps, 1), "log_domain_patches_per_step": round(logdom / prof_steps, 1),
                                        "sinkhorn_iterations_skipped_frac": round(skipped / max(1.0, 100.0 * live), 4)}
        if single:
            out["single_pair_mode"] = single
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_baseline_seconds, wl["benchmark"], wl["seed_config"], weights, normals, args.cloud)
    return out


def knn_stress(args, rank, world, distributed):
    """--config 5 (BASELINE configs[4]): kNN(64) + fused PPF, 32 clouds x 30000 points per call, clouds sharded over ranks (every
    rank its own clouds, no collective).  One step = one roitr_amd.pointops.knn_ppf call: grid build, query, column-0 drop, PPF."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from roitr_amd import pointops as P

    wl = WORKLOADS[5]
    B = args.pairs_per_step or wl["pairs"]
    N = args.n_points or wl["n_points"]
    K = 64
    rng = np.random.default_rng(5000 + rank)
    xyz = torch.from_numpy((rng.random((N * B, 3)) * 2.0).astype(np.float32)).cuda()
    nrm = rng.standard_normal((N * B, 3))
    nrm = torch.from_numpy((nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)).cuda()
    off = (torch.arange(1, B + 1, dtype=torch.int32) * N).cuda()

    def step():
        return P.knn_ppf(K, xyz, xyz, nrm, nrm, off, off)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # per-launch duration of the call on its stream (HIP events on torch's current stream = the stream pointops launches on)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1) / args.steps
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
    queries = N * B * args.steps * world
    algo = (24.0 * N + 20.0 * N * K) * B          # 24 R + 4 M K (idx) + 16 M K (ppf), queries == refs
    achieved = algo / (call_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": "knn+ppf call (grid_build_kernel + knn_cell_kernel / knn_gridsel_kernel / knn_replay_kernel, PPF fused)",
            "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
            "avg_launch_ms": round(call_ms, 5), "algorithmic_bytes_per_launch": int(algo), "launches_timed": args.steps,
            "pair_evaluations_brute_force": int(float(N) * N * B),
            "note": "an exact kNN is bound by VALU issue (selecting 64 of ~700 staged candidates per query), not by the 39 MB per cloud "
                    "it has to move (SURVEY.md finding 3): see valu_issue"}
    sq = load_profile_json("sq_knn_config5.json")
    from roitr_amd.build import source_hash
    if sq and sq.get("clouds") == B and sq.get("n_points") == N and sq.get("kernel_source_sha16") == source_hash():
        roof["valu_issue"] = sq
    out = {"metric": "kNN+PPF queries/s (k = 64, N = 30000 per cloud)", "value": round(queries / dt, 1), "unit": "queries/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": wl["text"].format(N=N, B=B), "baseline_config": 5, "clouds_per_step": B, "n_points": N, "k": K},
           "roofline": roof, "rooflines": [roof]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_knn(N, K, args.cpu_baseline_seconds)
    return out


def ot_stats(enable):
    """roitr_ot_stats: (live patches, Sinkhorn iterations skipped by the bit-exact fixed-point exit, patches served by the log-domain
    kernel) since counting was switched on; then switches counting on (zeroed) or off."""
    import ctypes
    from roitr_amd import _lib as L
    out = (ctypes.c_ulonglong * 3)()
    L.check(L.lib().roitr_ot_stats(ctypes.c_int(enable), out), "ot_stats")
    return [int(v) for v in out]


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def rooflines(prof, steps, dtype):
    """[dominant kernel family, kNN+PPF vs HBM, global transformer phase vs HBM, geo_table_kernel vs HBM] from the instrumented pass.

    gemm_kernel / geo_embed_kernel are MFMA-bound: achieved = algorithmic FLOPs (2*M*N*K per launch; 2*(1+k)*rows*C^2 for the
    embedding) / kernel time vs the MFMA peak of the operand dtype; the entry also carries the algorithmic HBM bytes per launch
    (every operand element read once, every result written once) so that the PMC `traffic` can be read against them.
    Geometry kernels are priced on algorithmic HBM bytes (SURVEY.md 8d / DESIGN.md): FPS n->m: 12n + 4m + 8n; kNN+PPF:
    24R + 24M[queries != refs] + 20MK.  The global-transformer phase is priced against HBM on the algorithmic bytes of the
    instrumented launches inside it (the stream of the geometric embedding E dominates: written once, read once per self layer)
    over the WHOLE phase time; its executed matrix work against the MFMA peak rides along as `mfma_frac_executed`."""
    if not prof or not steps:
        return []
    mfma_peak = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    fwd_ms = prof.get("phase.forward", {}).get("ms", 0.0)

    def entry(name, label=None, bound=None):
        p = prof[name]
        launches = max(p["launches"], 1)
        avg_ms = p["ms"] / launches
        share = round(p["ms"] / fwd_ms, 4) if fwd_ms > 0 else None
        mfma_class = name in ("gemm_kernel", "geo_embed_kernel", "gemm_kernel.mfma_roofed")
        if (bound or ("mfma" if mfma_class else "hbm")) == "mfma":
            achieved = p["bytes"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
            e = {"bound": "mfma", "kernel": label or name, "achieved": round(achieved, 3), "peak": mfma_peak, "unit": "TFLOP/s",
                 "frac": round(achieved / mfma_peak, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                 "algorithmic_flops_per_launch": int(p["bytes"] / launches), "launches_timed": int(p["launches"]),
                 "share_of_forward_time": share}
            if p.get("aux"):
                e["algorithmic_bytes_per_launch"] = int(p["aux"] / launches)
                e["hbm_gbs_on_algorithmic_bytes"] = round(p["aux"] / (p["ms"] * 1e-3) / 1e9, 1) if p["ms"] > 0 else 0.0
            return e
        nbytes = p["aux"] if name.startswith("phase.") else p["bytes"]
        achieved = nbytes / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
        return {"bound": "hbm", "kernel": label or name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                "algorithmic_bytes_per_launch": int(nbytes / launches), "launches_timed": int(p["launches"]), "share_of_forward_time": share}

    kernels = {k: v for k, v in prof.items() if not k.startswith("phase.") and k != "geo_embed_reference_flops" and "." not in k}
    roofs = []
    if kernels:
        top = max(kernels, key=lambda k: kernels[k]["ms"])
        roofs.append(entry(top))
        if top == "gemm_kernel":
            # the family priced honestly (round 5): its launches are split by the roof the roofline model gives them -- algorithmic FLOPs
            # per algorithmic byte above / below the machine balance (157.3 TFLOP/s over 8 TB/s = 19.7 FLOP per byte in fp32) -- and each
            # half is priced against its own roof.  `frac` of the family (above) stays the MFMA fraction of rounds 1 - 4.
            split = {}
            if "gemm_kernel.mfma_roofed" in prof:
                e = entry("gemm_kernel.mfma_roofed", "gemm_kernel launches above the machine balance (levels 3-4, global transformer, K >= 128 with wide N)", bound="mfma")
                split["mfma_roofed"] = {k: e[k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_timed", "share_of_forward_time", "hbm_gbs_on_algorithmic_bytes") if k in e}
            if "gemm_kernel.hbm_roofed" in prof:
                p = prof["gemm_kernel.hbm_roofed"]
                gbs = p["aux"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
                split["hbm_roofed"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                                       "avg_launch_ms": round(p["ms"] / max(p["launches"], 1), 5), "launches_timed": int(p["launches"]),
                                       "share_of_forward_time": round(p["ms"] / fwd_ms, 4) if fwd_ms > 0 else None,
                                       "tflops": round(p["bytes"] / (p["ms"] * 1e-3) / 1e12, 2) if p["ms"] > 0 else 0.0,
                                       "note": "level-1/2 layers (K = 64 / 128 over 1.3 - 5.1 M rows): algorithmic bytes (every operand read once, every result written once) over their launch time"}
            roofs[0]["split_by_roof"] = split
    if "knn_query_kernel" in prof:
        roofs.append(entry("knn_query_kernel", "knn+ppf (every knn_*_kernel launch of the forward, PPF fused)"))
    if "phase.global_transformer" in prof:
        p = prof["phase.global_transformer"]
        e = entry("phase.global_transformer", "global_transformer phase (algorithmic HBM bytes of the instrumented launches inside the phase -- "
                  "the E stream dominates -- over the whole phase time)", bound="hbm")
        tf = p["bytes"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        e["mfma_tflops_executed"] = round(tf, 3)
        e["mfma_frac_executed"] = round(tf / mfma_peak, 5)
        if "geo_embed_reference_flops" in prof:
            e["note"] = ("the geometric embedding is a function table (csrc/geo_table.hip): the %.2f TFLOP per step of the reference's four "
                         "(rows, C) x (C, C) projections are not executed and not priced here" % (prof["geo_embed_reference_flops"]["bytes"] / steps / 1e12))
        roofs.append(e)
    if "geo_table_kernel" in prof:
        roofs.append(entry("geo_table_kernel", "geo_table_kernel (geometric embedding from the LDS function table; bytes = E written once + index rows)"))
    return roofs


def whole_forward(prof, steps, dtype, ms_per_step, pmc):
    """The complete forward against both roofs: matrix FLOPs EXECUTED per step (every GEMM / embedding launch) and HBM bytes per
    step (PMC: FETCH_SIZE + WRITE_SIZE summed over every kernel of a step, profiles/pmc_traffic.json) over the UNINSTRUMENTED step time."""
    mfma_peak = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    fwd = prof.get("phase.forward")
    if not fwd or not steps or ms_per_step <= 0:
        return None
    flops = fwd["bytes"] / steps
    tf = flops / (ms_per_step * 1e-3) / 1e12
    out = {"executed_flops_per_step": int(flops), "tflops": round(tf, 3), "mfma_peak": mfma_peak, "mfma_frac": round(tf / mfma_peak, 5),
           "algorithmic_bytes_per_step_instrumented_kernels": int(fwd.get("aux", 0.0) / steps), "ms_per_step": ms_per_step}
    if pmc and pmc.get("total_hbm_bytes_per_step"):
        gbs = pmc["total_hbm_bytes_per_step"] / (ms_per_step * 1e-3) / 1e9
        out.update(pmc_hbm_bytes_per_step=int(pmc["total_hbm_bytes_per_step"]), hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 5),
                   traffic_source="profiles/pmc_traffic.json")
    return out


def attach_traffic(roofs, B, config):
    """roofline.traffic: HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in their own rocprofv3
    runs of this same command by scripts/collect_profiles.sh, gfx950 corrections applied in scripts/pmc_summary.py); only
    attached when the committed summary was taken at the same workload.  Returns the summary (or None)."""
    pmc = load_profile_json("pmc_traffic.json")
    if not pmc or pmc.get("pairs_per_step") != B or pmc.get("baseline_config", 2) != config:
        return None
    from roitr_amd.build import source_hash
    have = source_hash()
    if pmc.get("kernel_source_sha16") != have:
        # counters of another build must not ride on this build's timing: the entries keep `traffic: null` and say why
        why = ("profiles/pmc_traffic.json was collected from kernel sources %s, this run is %s: re-collect with scripts/collect_profiles.sh"
               % (pmc.get("kernel_source_sha16", "<unstamped>"), have))
        for roof in roofs:
            roof["traffic_note"] = why
        return None
    src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    for roof in roofs:
        name = roof["kernel"].split(" ")[0]
        if name == "knn+ppf":   # the instrumented class spans every knn_*_kernel: launch-weighted mean over them
            ks = [v for n, v in pmc.get("kernels", {}).items() if n.startswith("knn_") and "replay" not in n]
            n = sum(v["launches"] for v in ks)
            if n:
                roof["traffic"] = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks) / n)
                roof["traffic_source"] = src
            # what an exact kNN is bound by: the share of the chip's VALU issue slots its kernels use (SQ pass of this command,
            # kernels serialised; scripts/sq_pass.sh -> scripts/sq_forward_json.py)
            sq = load_profile_json("sq_forward.json")
            if sq and sq.get("knn_family") and sq.get("kernel_source_sha16") == have:
                roof["valu_issue_frac"] = sq["knn_family"]["valu_issue_frac"]
                roof["valu_issue_per_kernel"] = {n: v["valu_issue_frac"] for n, v in sq.get("kernels", {}).items() if n.startswith("knn_")}
                roof["valu_issue_source"] = "profiles/sq_forward.json (" + sq.get("definition", "") + ")"
            continue
        # the instrumented class `gemm_kernel` spans both kernels of csrc/gemm.hip (64 x 64 tiles and the small-grid gemm_small_kernel):
        # launch-weighted mean over the two, like `achieved` and `algorithmic_bytes_per_launch`
        names = ("gemm_kernel", "gemm_small_kernel") if name == "gemm_kernel" else (name,)
        ks = [pmc.get("kernels", {}).get(n) for n in names]
        ks = [k for k in ks if k]
        n = sum(k["launches"] for k in ks)
        if n:
            roof["traffic"] = round(sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / n)
            roof["traffic_source"] = src
            if roof.get("algorithmic_bytes_per_launch"):
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 3)
    return pmc


def cpu_baseline(N, budget_s, benchmark, seed_config, weights="plain", normals="random", cloud="uniform"):
    """The CPU oracle (oracle/, 'port' kind) on this host: full forwards of pairs of the same workload, per-stage ms included."""
    try:
        from oracle import roitr_ref
    except Exception as e:  # oracle model restatement not available
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return roitr_ref.timed_baseline(N, budget_s, benchmark=benchmark, seed_config=seed_config, weights=weights, normals=normals, cloud=cloud)


def cpu_baseline_knn(N, K, budget_s):
    """--config 5: the C restatement of knnquery_cuda_kernel.cu:65-108 (oracle/pointops_ref.c, brute force like the reference)
    + the numpy PPF on single clouds of the same size, on all host cores, for about `budget_s` seconds."""
    try:
        from oracle import roitr_ref
    except Exception as e:
        return {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return roitr_ref.timed_knn_baseline(N, K, budget_s)


if __name__ == "__main__":
    main()

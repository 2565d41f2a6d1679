"""CPU, world_size 2 over gloo: the pair partition, the single result gather of SURVEY.md 8e, and bench.py's timed loop +
cross-rank aggregation (roitr_amd/benchloop.py) with a stub engine -- the code that differs between N = 1 and N > 1."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from roitr_amd.shard import pairs_for_rank, gather_counts, gather_result_records, pack_records, slots_per_rank
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = pairs_for_rank(7, rank, world)
    counts = gather_counts(sum(mine))
    # the tester's path: records of a finished engine batch packed from (pair ids, row offsets, flat scores, aux)
    lens = [i + 1 for i in mine]
    starts = [0]
    for n in lens:
        starts.append(starts[-1] + n)
    flat = torch.cat([torch.arange(n, dtype=torch.float32) * (rank + 1) for n in lens])
    aux = torch.tensor([[0.5 * i, float("nan") if i == 3 else 0.25 * i] for i in mine], dtype=torch.float32)
    # average capacity 3 scores per pair, 4 slots per rank -> a pool of 12 words: rank 0 holds 1 + 3 + 5 + 7 = 16 scores (pair 6,
    # the last one in its pool, is cut to 3 and flagged), rank 1 holds 2 + 4 + 6 = 12 (exactly full: pairs above the AVERAGE fit)
    PER = 3
    batch = pack_records(mine, starts, flat, aux)
    merged = gather_result_records(batch, slots_per_rank(7, world), PER)
    out = {"rank": rank, "mine": mine, "counts": counts,
           "merged": None if merged is None else {str(k): v.tolist() for k, v in sorted(merged.items())},
           "meta": None if merged is None else {"n": {str(k): v for k, v in merged.n_scores.items()}, "trunc": merged.truncated,
                                                "ranks": merged.ranks_seen, "backend": merged.backend,
                                                "aux": {str(k): [a if a == a else None for a in v] for k, v in merged.aux.items()}}}
    open(os.path.join(os.environ["RESULT_DIR"], f"rank{rank}.json"), "w").write(json.dumps(out))
    dist.destroy_process_group()
""") % ROOT


def test_pairs_for_rank_partition():
    sys.path.insert(0, ROOT)
    from roitr_amd.shard import pairs_for_rank
    for world in (1, 2, 4, 8):
        parts = [pairs_for_rank(1623, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(1623))                      # every pair exactly once (3DMatch has 1623 pairs)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        assert all(i % world == r for r, p in enumerate(parts) for i in p)


def _torchrun(script, port):
    # every rank writes its result to its own file (two ranks printing to one pipe can interleave their lines)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", RESULT_DIR=str(script.parent))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [json.load(open(script.parent / f"rank{k}.json")) for k in (0, 1)]
    return {x["rank"]: x for x in res}


def test_two_process_gather_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    by_rank = _torchrun(script, 29533)
    assert sorted(by_rank) == [0, 1]
    assert by_rank[0]["mine"] == [0, 2, 4, 6] and by_rank[1]["mine"] == [1, 3, 5]
    assert by_rank[0]["counts"] == by_rank[1]["counts"] == [12, 9]
    assert by_rank[1]["merged"] is None
    merged = by_rank[0]["merged"]
    assert sorted(map(int, merged)) == list(range(7))
    for i in range(7):
        scale = 1 if i % 2 == 0 else 2
        assert merged[str(i)] == [float(v * scale) for v in range(3 if i == 6 else i + 1)]
    meta = by_rank[0]["meta"]
    assert meta["ranks"] == 2 and meta["backend"] == "gloo" and meta["trunc"] == [6]
    assert meta["n"] == {str(i): i + 1 for i in range(7)}            # true counts survive the cut
    assert meta["aux"]["2"] == [1.0, 0.5] and meta["aux"]["3"] == [1.5, None]   # IR / PIR ride in the header (nan kept)


def test_single_process_records_round_trip():
    """No process group: the same call returns the local records (what bench.py --gpus 1 and a 1-GPU tester run use)."""
    import torch
    sys.path.insert(0, ROOT)
    from roitr_amd.shard import DEFAULT_SCORES_PER_PAIR, block_words, gather_result_records, max_scores_per_pair
    assert max_scores_per_pair(256, 64, 3) == 49152 and max_scores_per_pair(256, 64, 3, mutual=False) == 98304
    recs = [(5, torch.tensor([0.25, 0.5])), (9, torch.zeros(0)), (2, torch.arange(4.0), (0.75, 0.125))]
    got = gather_result_records(recs, 4, 8)
    assert got.backend == "local" and got.ranks_seen == 1 and len(got) == 3 and sorted(got.keys()) == [2, 5, 9]
    assert got[5].tolist() == [0.25, 0.5] and got[9].numel() == 0 and got[2].tolist() == [0.0, 1.0, 2.0, 3.0]
    assert got.aux[2] == (0.75, 0.125) and got.n_scores == {5: 2, 9: 0, 2: 4} and got.truncated == []
    # the block is sized from the AVERAGE capacity: 512 pairs at the default are 8 MiB, not the 100 MB of the exact bound
    assert block_words(512, DEFAULT_SCORES_PER_PAIR) * 4 == 8 * 1024 * 1024
    # a full pool cuts the tail in slot order and says so; the true counts survive
    got = gather_result_records(recs, 3, 1)
    assert got[5].tolist() == [0.25, 0.5] and got[2].tolist() == [0.0] and got.truncated == [2] and got.n_scores[2] == 4


BENCH_WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from roitr_amd import benchloop
    from roitr_amd.shard import pack_records
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    class StubEngine:
        # the three calls bench.py makes on the model, on CPU tensors: a pair emits (its seed mod 5) scores, all equal to the seed
        def launch_batch(self, pairs, want_gt=True):
            return {"pairs": pairs}
        def finish_batch(self, h):
            time.sleep(0.002 * (rank + 1))     # rank 1 is the slow rank: the job time is ITS time
            res, starts, flat = [], [0], []
            for p in h["pairs"]:
                n = p["seed"] %% 5
                sc = torch.full((n,), float(p["seed"]))
                res.append({"corr_scores": sc}); flat.append(sc); starts.append(starts[-1] + n)
            h["starts"], h["flat"] = starts, torch.cat(flat) if flat else torch.zeros(0)
            return res
        def batch_records(self, h, ids, aux=None):
            return pack_records(ids, h["starts"], h["flat"], aux)

    B, STEPS, WARM = 3, 4, 2
    pool = [{"seed": 10 * i + rank} for i in range(7)]
    batch = lambda step: [pool[(step * B + j) %% len(pool)] for j in range(B)]
    model = StubEngine()
    benchloop.run_steps(model, batch, B, 0, WARM, rank, world, True, 8)
    dist.barrier()
    t0 = time.perf_counter()
    n_corr, recs = benchloop.run_steps(model, batch, B, WARM, STEPS, rank, world, True, 8)
    dist.barrier()
    dt = time.perf_counter() - t0
    agg = benchloop.aggregate(dt, n_corr, B, STEPS)
    out = {"rank": rank, "dt": dt, "n_corr": n_corr, "agg": agg, "recs": None}
    if recs is not None:
        out["recs"] = {"ids": sorted(recs.keys()), "n": {str(k): v for k, v in recs.n_scores.items()}, "ranks": recs.ranks_seen,
                       "summary": benchloop.gather_summary(recs, B, STEPS, 8),
                       "scores": {str(k): v.tolist() for k, v in recs.items()}}
    open(os.path.join(os.environ["RESULT_DIR"], f"rank{rank}.json"), "w").write(json.dumps(out))
    dist.destroy_process_group()
""") % ROOT


def test_bench_loop_two_ranks_gloo(tmp_path):
    """bench.py's timed loop + aggregation (roitr_amd/benchloop.py) under torch.distributed.run with 2 gloo ranks and a stub
    engine: value = pairs of ALL ranks / MAX time over ranks, record ids unique across ranks and steps and congruent to the
    rank (pair i -> rank i mod W), every rank's records arrive through the one gather, only rank 0 holds them."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER)
    res = _torchrun(script, 29534)
    assert sorted(res) == [0, 1]
    B, STEPS, W, WARM = 3, 4, 2, 2
    a0, a1 = res[0]["agg"], res[1]["agg"]
    assert a0 == a1                                                     # every rank computes the same whole-job numbers
    assert a0["world"] == 2 and a0["total_pairs"] == B * STEPS * W
    assert abs(a0["dt"] - max(res[0]["dt"], res[1]["dt"])) < 1e-9         # MAX over ranks, not this rank's time
    assert abs(a0["value"] - B * STEPS * W / a0["dt"]) < 1e-6
    assert a0["n_corr"] == res[0]["n_corr"] + res[1]["n_corr"]           # SUM over ranks
    assert res[1]["recs"] is None                                        # only rank 0 receives
    rec = res[0]["recs"]
    ids = rec["ids"]
    assert len(ids) == len(set(ids)) == B * STEPS * W                    # unique over steps, slots and ranks
    assert sum(1 for i in ids if i % W == 0) == B * STEPS                # id % world == rank: the DistributedSampler partition
    s = rec["summary"]
    assert rec["ranks"] == 2 and s["rccl_ranks_seen"] == 2 and s["backend"] == "gloo" and s["collectives"] == 1
    assert s["records"] == B * STEPS * W and s["truncated_pairs"] == 0
    assert sum(rec["n"].values()) == a0["n_corr"] == s["scores"]
    # a record's scores are the ones its rank produced in that step / slot (the score value is the pool entry's seed)
    for pid, sc in rec["scores"].items():
        pid = int(pid)
        rank, local = pid % W, pid // W
        step, slot = local // B, local % B
        seed = 10 * (((step + WARM) * B + slot) % 7) + rank
        assert sc == [float(seed)] * (seed % 5), (pid, sc)

"""CPU, world_size 2 over gloo: the pair partition and the single result gather of SURVEY.md 8e."""
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
    MAXS = 6   # pair 6 has 7 scores: its tail is cut and flagged
    block = pack_records(mine, starts, flat, MAXS, aux)
    merged = gather_result_records(block, slots_per_rank(7, world), MAXS)
    out = {"rank": rank, "mine": mine, "counts": counts,
           "merged": None if merged is None else {str(k): v.tolist() for k, v in sorted(merged.items())},
           "meta": None if merged is None else {"n": {str(k): v for k, v in merged.n_scores.items()}, "trunc": merged.truncated,
                                                "ranks": merged.ranks_seen, "backend": merged.backend,
                                                "aux": {str(k): [a if a == a else None for a in v] for k, v in merged.aux.items()}}}
    print("RESULT " + json.dumps(out), flush=True)
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


def test_two_process_gather_gloo(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l]
    assert len(res) == 2
    by_rank = {x["rank"]: x for x in res}
    assert by_rank[0]["mine"] == [0, 2, 4, 6] and by_rank[1]["mine"] == [1, 3, 5]
    assert by_rank[0]["counts"] == by_rank[1]["counts"] == [12, 9]
    assert by_rank[1]["merged"] is None
    merged = by_rank[0]["merged"]
    assert sorted(map(int, merged)) == list(range(7))
    for i in range(7):
        scale = 1 if i % 2 == 0 else 2
        assert merged[str(i)] == [float(v * scale) for v in range(min(i + 1, 6))]
    meta = by_rank[0]["meta"]
    assert meta["ranks"] == 2 and meta["backend"] == "gloo" and meta["trunc"] == [6]
    assert meta["n"] == {str(i): i + 1 for i in range(7)}            # true counts survive the cut
    assert meta["aux"]["2"] == [1.0, 0.5] and meta["aux"]["3"] == [1.5, None]   # IR / PIR ride in the header (nan kept)


def test_single_process_records_round_trip():
    """No process group: the same call returns the local records (what bench.py --gpus 1 and a 1-GPU tester run use)."""
    import torch
    sys.path.insert(0, ROOT)
    from roitr_amd.shard import gather_result_records, max_scores_per_pair
    assert max_scores_per_pair(256, 64, 3) == 49152 and max_scores_per_pair(256, 64, 3, mutual=False) == 98304
    recs = [(5, torch.tensor([0.25, 0.5])), (9, torch.zeros(0)), (2, torch.arange(4.0), (0.75, 0.125))]
    got = gather_result_records(recs, 4, 8)
    assert got.backend == "local" and got.ranks_seen == 1 and len(got) == 3 and sorted(got.keys()) == [2, 5, 9]
    assert got[5].tolist() == [0.25, 0.5] and got[9].numel() == 0 and got[2].tolist() == [0.0, 1.0, 2.0, 3.0]
    assert got.aux[2] == (0.75, 0.125) and got.n_scores == {5: 2, 9: 0, 2: 4}

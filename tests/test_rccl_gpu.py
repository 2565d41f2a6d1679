"""GPU: the one collective of the path through RCCL.  A 1-rank 'nccl' process group is created on the box in a child
process (bounded by a timeout so that a wedged RCCL init cannot hang the suite) and the product's own
shard.gather_result_records round-trips the records of a real engine batch through it."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from roitr_amd.harness import build_model, pair_to_device
    from roitr_amd.shard import gather_result_records
    from roitr_amd.synthetic import make_pair
    model = build_model("3DMatch", weights="selective")     # thousands of correspondences per pair (the plain weights emit ~30)
    pairs = [pair_to_device(make_pair(5000, config=2, pair_index=i, normals="field")) for i in range(3)]
    with torch.no_grad():
        h = model.launch_batch(pairs)
        outs = model.finish_batch(h)
    aux = torch.tensor([[0.5, 0.25], [1.0, float("nan")], [0.0, 0.75]], device="cuda")
    block = model.batch_records(h, [10, 11, 12], aux)
    rec = gather_result_records(block, 4, model.record_scores_per_pair())
    ok = all(torch.equal(rec[10 + i], outs[i]["corr_scores"].cpu()) for i in range(3))
    ok = ok and rec.aux[10] == (0.5, 0.25) and rec.aux[11][0] == 1.0 and rec.aux[11][1] != rec.aux[11][1] and rec.aux[12] == (0.0, 0.75)
    print("RESULT " + json.dumps({"backend": rec.backend, "ranks": rec.ranks_seen, "n": len(rec), "equal": ok,
                                  "counts": [rec.n_scores[10 + i] for i in range(3)], "lens": [int(o["corr_scores"].shape[0]) for o in outs],
                                  "trunc": rec.truncated}), flush=True)
    dist.destroy_process_group()
""") % ROOT


def test_result_gather_through_rccl_world1(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l]
    assert res and res[0]["backend"] == "nccl" and res[0]["ranks"] == 1 and res[0]["n"] == 3
    assert res[0]["equal"] and res[0]["trunc"] == [] and res[0]["counts"] == res[0]["lens"] and min(res[0]["lens"]) > 100

"""Stress of calls in flight with the sampling-ahead mode mixed in (a race shows as a bitwise mismatch):
    python scripts/stress_inflight.py [rounds]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from roitr_amd.harness import build_model, pair_to_device
KEYS = ("src_point_feats", "tgt_point_feats", "src_node_feats", "tgt_node_feats", "corr_scores", "src_corr_points", "tgt_corr_points",
        "matching_scores", "gt_src_node_occ", "gt_node_corr_overlaps", "src_node_corr_indices")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(7)
model = build_model("3DMatch", weights="selective")
sizes = (1024, 1500, 2048, 3000, 4000, 5000)
pool = [pair_to_device(make_pair(sizes[i % len(sizes)], config=2, pair_index=i, normals="field")) for i in range(24)]
torch.cuda.synchronize()
batches = [[pool[j] for j in rng.sample(range(24), rng.choice((1, 1, 2, 3, 6, 12)))] for _ in range(16)]
with torch.no_grad():
    refs = []
    for b in batches:
        refs.append(model.forward_batch(b)); torch.cuda.synchronize()
    bad = calls = 0
    for r in range(rounds):
        order = [rng.randrange(len(batches)) for _ in range(8)]
        hs = [(i, model.launch_batch(batches[i], want_gt=rng.random() < 0.8, inputs_resident=rng.random() < 0.7)) for i in order]
        for i, h in hs:
            got = model.finish_batch(h)
            calls += 1
            for x, y in zip(got, refs[i]):
                for k in KEYS:
                    if k.startswith("gt_") and not h["have_gt"]:
                        continue
                    if not torch.equal(x[k], y[k]):
                        bad += 1; print("MISMATCH round", r, "batch", i, k)
print(f"stress_inflight: {calls} calls in flight, {bad} mismatches")
sys.exit(1 if bad else 0)

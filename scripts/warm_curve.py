"""Per-step time of the bench workload from a cold process (how long does the box take to reach steady state?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from tests.gpu_util import build_model, pair_to_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = build_model("3DMatch")
pool = [pair_to_device(make_pair(5000, config=2, pair_index=i)) for i in range(B)]
ts = []
with torch.no_grad():
    for s in range(60):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.forward_batch(pool)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("ms per step:", " ".join(f"{t:.1f}" for t in ts))

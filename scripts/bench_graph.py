"""Plain vs HIP-graph forward at small batches: python scripts/bench_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from tests.gpu_util import build_model, pair_to_device
model = build_model("3DMatch")
pool = [pair_to_device(make_pair(5000, config=2, pair_index=i)) for i in range(64)]
for B in (1, 2, 8, 32):
    for graph in (False, True):
        batches = [pool[(i * B) % 64:(i * B) % 64 + B] for i in range(64 // B)] if B < 64 else [pool]
        with torch.no_grad():
            for i in range(8):
                model.forward_batch(batches[i % len(batches)], graph=graph)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 40
            h = model.launch_batch(batches[0], graph=graph)
            for s in range(n):
                nx = model.launch_batch(batches[(s + 1) % len(batches)], graph=graph) if s + 1 < n else None
                model.finish_batch(h); h = nx
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} graph={graph}: {1e3*dt/n:.3f} ms per forward, {B*n/dt:.1f} pairs/s")

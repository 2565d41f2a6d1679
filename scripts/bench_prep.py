"""Input-prep / evaluator microbenchmarks: normals for B clouds of N points on the GPU vs the CPU oracle.

    python scripts/bench_prep.py [B] [N]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from roitr_amd import prep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = np.random.default_rng(0)
xyz = rng.uniform(0, 2, (B * N, 3)).astype(np.float32)
off = (np.arange(1, B + 1) * N).astype(np.int32)
x, o = torch.from_numpy(xyz).cuda(), torch.from_numpy(off).cuda()
for _ in range(2):
    nrm = prep.estimate_normals(x, o, 33)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    nrm = prep.estimate_normals(x, o, 33)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
# algorithmic bytes: xyz in (12 B/pt) + normals out (12 B/pt); the kNN(33) index list (132 B/pt) stays on chip in the ideal
print(f"GPU estimate_normals: {B} clouds x {N} pts: {ms:.3f} ms  = {B / ms * 1e3:.0f} clouds/s, {B * N / ms / 1e6:.2f} Gpts/s, "
      f"{B * N * 24 / ms / 1e6:.1f} GB/s algorithmic")
from oracle import prep_ref
cores = len(os.sched_getaffinity(0))
t0 = time.perf_counter(); nc = 0
while time.perf_counter() - t0 < 10.0 and nc < B:
    ref = prep_ref.estimate_normals_fast(xyz[nc * N:(nc + 1) * N], 33, threads=cores); nc += 1
dt = time.perf_counter() - t0
print(f"CPU oracle (C kNN on {cores} threads + numpy eigh): {nc} clouds in {dt:.2f} s = {nc / dt:.1f} clouds/s")
g = nrm[:N].cpu().numpy().astype(np.float64)
r0 = prep_ref.estimate_normals_fast(xyz[:N], 33, threads=cores)
print("agreement on cloud 0: min |dot| =", float(np.abs((g * r0).sum(1)).min()), " sign agreement =", float(((g * r0).sum(1) > 0).mean()))

"""Micro-benchmark: FPS kernel time vs workgroup shape (ROITR_FPS_BLOCK) for one level-1 3DMatch-sized cloud set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roitr_amd import pointops as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(0)
xyz = torch.from_numpy((rng.random((n * b, 3)) * 2).astype(np.float32)).cuda()
off = torch.arange(1, b + 1, dtype=torch.int32).cuda() * n
noff = torch.arange(1, b + 1, dtype=torch.int32).cuda() * (n // 4)
for _ in range(2):
    P.furthestsampling(xyz, off, noff)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    idx = P.furthestsampling(xyz, off, noff)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print(f"block={os.environ.get('ROITR_FPS_BLOCK','auto')} n={n} b={b}: {dt*1e3:.3f} ms  ({dt/(n//4)*1e9:.0f} ns/iteration)  checksum {int(idx.sum())}")

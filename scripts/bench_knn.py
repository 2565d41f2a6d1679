"""Micro-benchmark (BASELINE.json configs[4]): knnquery(k+1)+drop-self+PPF on synthetic clouds, HBM roofline fraction.
Algorithmic bytes: 24*R (xyz+normals) + 20*M*K (idx int32 + ppf 4xf32) for self queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roitr_amd import pointops as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(0)
xyz = torch.from_numpy((rng.random((n * b, 3)) * 2).astype(np.float32)).cuda()
nrm = torch.nn.functional.normalize(torch.randn(n * b, 3, device="cuda"), dim=1).contiguous()
off = (torch.arange(1, b + 1, dtype=torch.int32) * n).cuda()
for _ in range(2):
    g, ppf = P.knn_ppf(k, xyz, xyz, nrm, nrm, off, off)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 10
e0.record()
for _ in range(R):
    g, ppf = P.knn_ppf(k, xyz, xyz, nrm, nrm, off, off)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / R
by = 24.0 * n * b + 20.0 * n * b * k
print(f"knn+ppf n={n} k={k} clouds={b}: {ms*1e3:.1f} us/call (incl. grid build + alloc)  algorithmic {by/1e6:.2f} MB -> {by/ms/1e6:.1f} GB/s = {by/ms/1e6/8000*100:.2f}% of 8 TB/s")

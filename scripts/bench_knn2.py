import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roitr_amd import pointops as P
n, k, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
xyz = torch.from_numpy((rng.random((n * b, 3)) * 2).astype(np.float32)).cuda()
nrm = torch.nn.functional.normalize(torch.randn(n * b, 3, device="cuda"), dim=1).contiguous()
off = (torch.arange(1, b + 1, dtype=torch.int32) * n).cuda()
def timeit(fn, R=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3
print("knn only (idx+dist2):", round(timeit(lambda: P.knnquery_raw(k + 1, xyz, xyz, off, off)), 1), "us")
print("knn + group + ppf   :", round(timeit(lambda: P.knn_ppf(k, xyz, xyz, nrm, nrm, off, off)), 1), "us")

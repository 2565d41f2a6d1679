"""kNN (+ fused PPF) calls of a 512-pair forward, one by one, outputs preallocated: ms per call (grid build apart).
usage: python scripts/bench_knn_shapes.py [clouds=1024] [points per cell=6]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roitr_amd import _lib as L
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
occ2 = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
lib = L.lib()
g = torch.Generator(device="cuda").manual_seed(0)
n1, n2, n3 = 5000, 1250, 312
p1 = (torch.rand((NC * n1, 3), device="cuda", generator=g) * 2).contiguous()
nr1 = torch.nn.functional.normalize(torch.randn((NC * n1, 3), device="cuda", generator=g), dim=1).contiguous()
def sub(p, nr, n, m):   # m points of every cloud (a stand-in for the FPS picks: spread subsets of the cloud)
    idx = (torch.arange(NC, device="cuda")[:, None] * n + torch.stack([torch.randperm(n, device="cuda", generator=g)[:m] for _ in range(8)])[torch.arange(NC, device="cuda") % 8]).reshape(-1)
    return p[idx].contiguous(), nr[idx].contiguous()
p2, nr2 = sub(p1, nr1, n1, n2)
p3, nr3 = sub(p2, nr2, n2, n3)
other = (torch.rand((NC * n1, 3), device="cuda", generator=g) * 2).contiguous()
off = lambda n: (torch.arange(1, NC + 1, dtype=torch.int32, device="cuda") * n).contiguous()
o1, o2, o3 = off(n1), off(n2), off(n3)
def ws_for(n, m): return torch.empty(lib.roitr_knn_workspace_bytes(NC, NC * n, NC * m), dtype=torch.uint8, device="cuda")
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
st = L.stream_ptr()
def grid(p, o, n, mcap, ws, occ): return lambda: L.check(lib.roitr_knn_build_grid_ex(NC, NC * n, NC * mcap, L.ptr(p), L.ptr(o), L.ptr(ws), ctypes.c_float(occ), st), "grid")
def query(ns, p, o, n, q, qo, m, ws, nr=None, qnr=None, mcap=None, idx=False):
    M = NC * m
    gi = torch.empty((M, ns - 1), dtype=torch.int32, device="cuda") if nr is not None else None
    pf = torch.empty((M, ns - 1, 4), dtype=torch.float32, device="cuda") if nr is not None else None
    ii = torch.empty((M, ns), dtype=torch.int32, device="cuda") if idx else None
    dd = torch.empty((M, ns), dtype=torch.float32, device="cuda") if idx else None
    return lambda: L.check(lib.roitr_knnquery_ex(NC, NC * n, M, ns, L.ptr(p), L.ptr(q), L.ptr(o), L.ptr(qo), L.ptr(ii), L.ptr(dd), L.ptr(gi), L.ptr(pf),
                                                 L.ptr(nr), L.ptr(qnr), 1, NC * (mcap or m), L.ptr(ws), st), "knn")
ws1, ws2 = ws_for(n1, n1), ws_for(n2, n1)
rows = []
rows.append(("grid L1 (5000/cloud) occ %.1f" % occ2, timeit(grid(p1, o1, n1, n1, ws1, occ2))))
rows.append(("self L1 ns=9 +ppf   M=%d" % (NC * n1), timeit(query(9, p1, o1, n1, p1, o1, n1, ws1, nr1, nr1))))
rows.append(("TD 1->2 ns=17 +ppf  M=%d" % (NC * n2), timeit(query(17, p1, o1, n1, p2, o2, n2, ws1, nr1, nr2, mcap=n1))))
rows.append(("grid L2 (1250/cloud) occ %.1f" % occ2, timeit(grid(p2, o2, n2, n1, ws2, occ2))))
rows.append(("self L2 ns=17 +ppf  M=%d" % (NC * n2), timeit(query(17, p2, o2, n2, p2, o2, n2, ws2, nr2, nr2, mcap=n1))))
rows.append(("TD 2->3 ns=17 +ppf  M=%d" % (NC * n3), timeit(query(17, p2, o2, n2, p3, o3, n3, ws2, nr2, nr3, mcap=n1))))
rows.append(("3-NN L1 in L2 ns=3  M=%d" % (NC * n1), timeit(query(3, p2, o2, n2, p1, o1, n1, ws2, mcap=n1, idx=True))))
# (round 5: a finer grid of its own for the 3-NN query -- 1.5 / 2 / 3 points per cell -- 0.88 / 0.74 / 0.61 ms against 0.63 on the shared
#  6-point grid: that query is not bound by its candidate scan)
d2 = torch.empty((NC * n1,), dtype=torch.float32, device="cuda")
cap2 = ctypes.c_float(0.0375 * 0.0375 * 1.01)
rows.append(("within 0.0375       M=%d" % (NC * n1), timeit(lambda: L.check(lib.roitr_knn_within(NC, NC * n1, NC * n1, L.ptr(p1), L.ptr(other), L.ptr(o1), L.ptr(o1), cap2,
                                                                                               L.ptr(d2), 1, NC * n1, L.ptr(ws1), st), "within"))))
tot = 0.0
for name, ms in rows:
    print(f"{name:40s} {ms:8.3f} ms"); tot += ms
print(f"{'total':40s} {tot:8.3f} ms   (ROITR_KNN_X={os.environ.get('ROITR_KNN_X', 'default')})")

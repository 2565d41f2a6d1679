"""GEMM microbenchmark through the C ABI: python scripts/bench_gemm.py M N K [reps] [pad]  -> TFLOP/s (HIP events).
pad: extra floats per row of A and W (leading dimension K + pad) -- probes power-of-two row-stride effects."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd import _lib as L
from roitr_amd.ops import _Gemm
M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
pad = int(sys.argv[5]) if len(sys.argv) > 5 else 0
a = torch.randn(M, K + pad, device="cuda"); w = torch.randn(N, K + pad, device="cuda"); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
g = _Gemm(M, N, K, L.ptr(a), L.ptr(None), K + pad, L.ptr(None), 0, L.ptr(w), K + pad, L.ptr(None), 0, L.ptr(b), 1.0, 0,
          L.ptr(out), N, 1, 0, 0, 0, 0, 0, 0, L.ptr(None), 0, 0)
run = lambda: L.check(L.lib().roitr_gemm(ctypes.byref(g), L.stream_ptr()), "gemm")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
ref = a[:257, :K] @ w[:, :K].T + b
err = float((out[:257] - ref).abs().max())
print(f"M {M} N {N} K {K} pad {pad}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s  (in+out {(M*K+M*N)*4/ms/1e6:.0f} GB/s)  max|err| vs torch {err:.2e}")

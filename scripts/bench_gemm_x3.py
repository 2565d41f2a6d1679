"""fp32 MFMA GEMM (csrc/gemm.hip) vs the three-way bf16 split (csrc/gemm_x3.hip) on the K >= 256 shapes of a 512-pair forward:
time per launch, TFLOP/s (fp32-equivalent 2 M N K), error against float64 of both, row-count invariance of the split kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd import ops

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

g = torch.Generator(device="cpu").manual_seed(7)
shapes = [(319488, 768, 256, 0), (319488, 256, 512, 0), (319488, 256, 256, 0), (319488, 256, 256, 1), (79872, 768, 256, 0), (79872, 256, 512, 1), (79872, 512, 256, 0),
          (39936, 256, 256, 0), (39936, 512, 256, 0), (156, 256, 256, 0), (156, 768, 256, 0)]
for M, N, K, ln in shapes:
    x = torch.randn((M, K), generator=g).cuda()
    w = (torch.randn((N, K), generator=g) / K ** 0.5).cuda()
    b = torch.randn((N,), generator=g).cuda()
    gam, bet = torch.randn((N,), generator=g).cuda(), torch.randn((N,), generator=g).cuda()
    if ln:
        res = torch.randn((M, N), generator=g).cuda()
        f32 = lambda: ops.linear_layernorm(x, w, b, gam, bet, res=res, relu=True)
        w3 = ops.split_bf16x3(w)
        x3 = lambda: ops.linear_layernorm(x, w, b, gam, bet, res=res, relu=True, x3=True)
    else:
        f32 = lambda: ops.linear(x, w, b, relu=True)
        x3 = lambda: ops.linear(x, w, b, relu=True, x3=True)
    t32, t3 = timeit(f32), timeit(x3)   # (the x3 wrapper re-splits the weight every call: ~N*K elements, negligible at these M)
    a, c = f32(), x3()
    rows = torch.randint(0, M, (min(M, 2048),), generator=g).cuda()
    ref = x[rows].double() @ w.double().T + b.double()
    if ln:
        t = ref + res[rows].double()
        ref = ((t - t.mean(1, keepdim=True)) / torch.sqrt(t.var(1, unbiased=False, keepdim=True) + 1e-5) * gam.double() + bet.double())
    ref = ref.clamp_min(0)
    e32, e3 = (a[rows].double() - ref).abs().max().item(), (c[rows].double() - ref).abs().max().item()
    m32, m3 = (a[rows].double() - ref).abs().mean().item(), (c[rows].double() - ref).abs().mean().item()
    small = (ops.linear_layernorm(x[:156].contiguous(), w, b, gam, bet, res=res[:156].contiguous(), relu=True, x3=True) if ln
             else ops.linear(x[:156].contiguous(), w, b, relu=True, x3=True))
    inv = torch.equal(small, c[:156])
    fl = 2.0 * M * N * K
    print(f"M {M:7d} N {N:4d} K {K:4d} ln {ln}: fp32 {t32:7.4f} ms {fl / t32 / 1e9:6.1f} TF | x3 {t3:7.4f} ms {fl / t3 / 1e9:6.1f} TF  x{t32 / t3:4.2f} | "
          f"max err fp32 {e32:.2e} x3 {e3:.2e} mean {m32:.2e} / {m3:.2e} | rows invariant {inv}")

"""Micro-benchmark of csrc/local_block.hip local_td_kernel at the level-2 shape of the 512-pair step (tuning aid):
    python scripts/bench_local_td.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd import ops

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
I, H, n_in, n_out = 64, 128, 5000, 1250
NC = 2 * pairs
N_in, M = NC * n_in, NC * n_out
x = torch.randn((N_in, I), device=dev, generator=g)
cloud = torch.arange(NC, device=dev).repeat_interleave(n_out)
local = torch.randint(0, n_in, (M,), device=dev, generator=g)
node_idx = (cloud * n_in + local).to(torch.int32)
# neighbours: nearby rows of the same cloud (what cell-ordered kNN groups look like to the caches)
off = torch.randint(-60, 61, (M, 16), device=dev, generator=g)
grp = (cloud[:, None] * n_in + (local[:, None] + off).remainder(n_in)).to(torch.int32)
ppf = torch.rand((M, 16, 4), device=dev, generator=g)
r = lambda *s: torch.randn(s, device=dev, generator=g) / (s[-1] ** 0.5)
w = dict(wqqt=r(H + 4 * I, I), bqqt=r(H + 4 * I), wv=r(H, I), bv=r(H), wpe=r(H, 4), wvpe=r(H, 4), bvpe=r(H), wcat=r(H, H + I), bcat=r(H),
         norm_w=1 + 0.1 * r(H), norm_b=0.1 * r(H), wout=r(H, H), bout=r(H))
for _ in range(2):
    ops.local_td(x, node_idx, grp, ppf, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.local_td(x, node_idx, grp, ppf, w)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
flops = 2.0 * M * ((H + 4 * I) * I + H * I + H * (H + I) + H * H)
print(f"local_td M={M} N_in={N_in}: {ms:7.3f} ms per launch ({flops / ms / 1e9:6.1f} TFLOP/s on the on-chip GEMMs)")

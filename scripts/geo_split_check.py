"""fp32 MFMA vs split-bf16 geometric embedding: error against float64 and kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roitr_amd import ops
rng = np.random.default_rng(4)
C, rows = 256, 1_500_000
d = torch.from_numpy((rng.uniform(0, 3, rows) / 0.2).astype(np.float32)).cuda()
a = torch.from_numpy((rng.uniform(0, np.pi, (rows, 3)) * 180 / (15 * np.pi)).astype(np.float32)).cuda()
div = torch.exp(torch.arange(0, C, 2).float() * (-np.log(10000.0) / C)).cuda()
wd, wa = torch.randn(C, C).cuda() / 16, torch.randn(C, C).cuda() / 16
bd, ba = torch.randn(C).cuda(), torch.randn(C).cuda()
n = 4096
def emb(v):
    om = v.double()[..., None] * div.double()
    return torch.stack([torch.sin(om), torch.cos(om)], -1).reshape(*v.shape, C)
ref = emb(d[:n]) @ wd.double().T + bd.double() + (emb(a[:n]) @ wa.double().T + ba.double()).max(1).values
for split in (False, True):
    out = ops.geo_embed(d, a, div, wd, bd, wa, ba, split=split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        out = ops.geo_embed(d, a, div, wd, bd, wa, ba, split=split)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    err = (out[:n].double() - ref).abs()
    print(f"split={split}: {ms:.2f} ms  {2.0*rows*4*C*C/ms/1e9:.1f} TFLOP/s-equivalent  max abs err {float(err.max()):.3e}  rms {float(err.pow(2).mean().sqrt()):.3e}  (|E| max {float(ref.abs().max()):.2f})")

"""Micro-benchmark of csrc/local_block.hip at the level-1 / level-2 shapes of the 512-pair step (tuning aid).
    python scripts/bench_local_block.py [pairs]
Prints ms per launch of the kernel, of the kernel without its attention phase, and of the attention phase alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roitr_amd import ops

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for H, K, n_cloud in ((64, 8, 5000), (128, 16, 1250)):
    M = 2 * pairs * n_cloud
    x = torch.randn((M, H), device=dev, generator=g)
    kv = torch.randn((M, 2 * H), device=dev, generator=g)
    # neighbours: nearby rows of the same cloud (what cell-ordered kNN groups look like to the caches)
    base = torch.arange(M, device=dev).unsqueeze(1)
    off = torch.randint(-40, 41, (M, K), device=dev, generator=g)
    cloud0 = (base // n_cloud) * n_cloud
    grp = (cloud0 + (base - cloud0 + off).remainder(n_cloud)).to(torch.int32)
    ppf = torch.rand((M, K, 4), device=dev, generator=g)
    r = lambda *s: torch.randn(s, device=dev, generator=g) / (s[-1] ** 0.5)
    w = dict(wq=r(H, H), bq=r(H), wpe=r(H, 4), bpe=r(H), wvpe=r(H, 4), bvpe=r(H), wcat=r(H, 2 * H), bcat=r(H), norm_w=1 + 0.1 * r(H),
             norm_b=0.1 * r(H), wout=r(H, H), bout=r(H), bn2_w=1 + 0.1 * r(H), bn2_b=0.1 * r(H))
    for variant, name in ((None, "kernel"), (1, "no attention"), (2, "attention only"), (10, "kernel, 2x rows"), (11, "no attention, 2x"), (12, "attn only, 2x")):
        for _ in range(2):
            ops.local_block(x, kv, grp, ppf, w, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.local_block(x, kv, grp, ppf, w, variant=variant)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flops = 8.0 * M * H * H
        print(f"H={H} K={K} M={M}: {name:20s} {ms:7.3f} ms   ({flops / ms / 1e9:6.1f} TFLOP/s on the on-chip GEMMs)")

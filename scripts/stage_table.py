"""Per-stage times, GPU engine vs CPU oracle, for the SURVEY.md 8(d) table (N = 1024 and 5000).

    python scripts/stage_table.py            # on the GPU box: both columns;  --cpu-only works anywhere
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import roitr_ref
from roitr_amd.synthetic import make_pair

cpu_only = "--cpu-only" in sys.argv
cores = len(os.sched_getaffinity(0))
sd = roitr_ref.closed_form_state()
for N in (1024, 5000):
    t = {}
    t0 = time.perf_counter()
    roitr_ref.forward(sd, make_pair(N, config=2, pair_index=0), threads=cores, timings=t)
    tot = time.perf_counter() - t0
    print(f"CPU oracle N={N} ({cores} cores): total {tot:.2f} s | " + " | ".join(f"{k} {v:.3f}" for k, v in t.items()))
    if cpu_only:
        continue
    import torch
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch")
    for B in (1, 128):
        pairs = [pair_to_device(make_pair(N, config=2, pair_index=i)) for i in range(B)]
        with torch.no_grad():
            for _ in range(2):
                model.forward_batch(pairs)
            model.profile_reset()
            reps = 5
            for _ in range(reps):
                model.forward_batch(pairs)
        ph = model.profile_read(kernels_only=False)
        row = {k.replace("phase.", ""): v["ms"] / reps for k, v in ph.items() if k.startswith("phase.")}
        print(f"GPU N={N} B={B}: " + " | ".join(f"{k} {v:.3f} ms" for k, v in row.items()) + f" | per pair {row.get('forward', 0) / B:.3f} ms")

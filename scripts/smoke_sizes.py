"""Forward smoke at the other BASELINE.json sizes: 3DMatch N=30000 and 4DMatch (factor 2) N=8000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from tests.gpu_util import build_model, pair_to_device
for bench, N, B, cfg in (("3DMatch", 30000, 2, 2), ("4DMatch", 8000, 8, 4)):
    model = build_model(bench)
    pairs = [pair_to_device(make_pair(N, config=cfg, pair_index=i)) for i in range(B)]
    with torch.no_grad():
        res = model.forward_batch(pairs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            res = model.forward_batch(pairs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{bench} N={N} B={B}: {dt*1e3:.1f} ms per forward = {B/dt:.1f} pairs/s; corr per pair {[int(r['corr_scores'].shape[0]) for r in res]}; "
          f"nodes {res[0]['src_nodes'].shape[0]}, coarse {res[0]['src_node_corr_indices'].shape[0]}, finite {bool(torch.isfinite(res[0]['src_point_feats']).all())}")
    del model

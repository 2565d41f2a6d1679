"""Measured error of the bf16 operand mode against the fp32 CPU oracle (4DMatch settings): prints the statistics the
tolerances in tests/test_bf16_gpu.py are set from.  python scripts/bf16_report.py [n_points] [pairs]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import roitr_ref as R
from roitr_amd.harness import build_model, pair_to_device
from roitr_amd.synthetic import make_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cores = len(os.sched_getaffinity(0))
mb = build_model("4DMatch", operand_dtype="bf16", weights="selective")
mf = build_model("4DMatch", weights="selective")
for i in range(pairs):
    pair = make_pair(n, config=4, pair_index=2 + i, normals="field")
    ref = R.forward(R.closed_form_state(2, "selective"), pair, cfg=dict(R.FDMATCH_CFG), threads=cores)
    with torch.no_grad():
        ob = mb.forward(**pair_to_device(pair))
        of = mf.forward(**pair_to_device(pair))
    rep = {"pair": i, "n": n}
    for tag, o in (("bf16", ob), ("f32", of)):
        for k in ("src_node_feats", "tgt_node_feats", "src_point_feats", "tgt_point_feats"):
            a, b = o[k].cpu().numpy(), ref[k]
            e = np.abs(a - b)
            cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-30)
            rep[f"{tag}.{k}"] = {"max": float(e.max()), "mean": float(e.mean()), "ref_absmean": float(np.abs(b).mean()), "cos_min": float(cos.min())}
        got = set(zip(o["tgt_node_corr_indices"].tolist(), o["src_node_corr_indices"].tolist()))
        want = set(zip(ref["tgt_node_corr_indices"].tolist(), ref["src_node_corr_indices"].tolist()))
        rep[f"{tag}.coarse"] = {"got": len(got), "want": len(want), "common": len(got & want)}
        rep[f"{tag}.fine"] = {"got": int(o["corr_scores"].shape[0]), "want": int(ref["corr_scores"].shape[0])}
        rep[f"{tag}.nodes_equal"] = bool(np.array_equal(o["src_nodes"].cpu().numpy(), ref["src_nodes"]) and np.array_equal(o["tgt_nodes"].cpu().numpy(), ref["tgt_nodes"]))
    print(json.dumps(rep))

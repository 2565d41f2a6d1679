"""Shared helpers for the GPU parity tests."""
import numpy as np
import torch


def build_model(benchmark="3DMatch"):
    from roitr_amd.config import test_config
    from roitr_amd.riga import create_model, state_dict_layout
    from roitr_amd.weights import closed_form_param
    cfg = test_config(benchmark)
    model = create_model(cfg)
    sd = model.state_dict()
    for k, shape, kind in state_dict_layout(model.factor, model.architecture):  # factor 2 for 4DMatch
        if kind == "param":
            sd[k].copy_(torch.from_numpy(closed_form_param(k, tuple(shape))))
    model = model.cuda().eval()
    model.sync_engine()
    return model


def pair_to_device(pair):
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in pair.items()}
    return dict(src_pcd=t["src_points"], tgt_pcd=t["tgt_points"], src_feats=t["src_feats"], tgt_feats=t["tgt_feats"],
                src_normals=t["src_normals"], tgt_normals=t["tgt_normals"], rot=t["rot"], trans=t["trans"],
                src_raw_pcd=t["raw_src_pcd"])


def golden_pair_inputs(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("in.")}

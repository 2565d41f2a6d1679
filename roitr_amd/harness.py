"""Small host-side helpers shared by bench.py, the tester entry point and the tests: a model filled with the
deterministic closed-form weights (there are no checkpoints in this image) and numpy pair -> device arguments."""
import numpy as np
import torch


def build_model(benchmark="3DMatch", operand_dtype="f32", weights="plain"):
    """create_model(test config of `benchmark`) with closed-form weights on the current ROCm device.
    operand_dtype: 'f32' (reference arithmetic) or 'bf16' (BASELINE config 4: bf16 operand storage for the dense layers).
    weights: closed-form variant, 'plain' | 'selective' (roitr_amd/weights.py)."""
    from .config import test_config
    from .riga import create_model, state_dict_layout
    from .weights import closed_form_param
    cfg = test_config(benchmark)
    if operand_dtype != "f32":
        cfg["operand_dtype"] = operand_dtype
    model = create_model(cfg)
    sd = model.state_dict()
    for k, shape, kind in state_dict_layout(model.factor, model.architecture):  # factor 2 for 4DMatch
        if kind == "param":
            sd[k].copy_(torch.from_numpy(closed_form_param(k, tuple(shape), weights)))
    model = model.cuda().eval()
    model.sync_engine()
    return model


def pair_to_device(pair):
    """synthetic.make_pair() dict -> the keyword arguments of RIGA_v2.forward, resident on the device."""
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in pair.items()}
    return dict(src_pcd=t["src_points"], tgt_pcd=t["tgt_points"], src_feats=t["src_feats"], tgt_feats=t["tgt_feats"],
                src_normals=t["src_normals"], tgt_normals=t["tgt_normals"], rot=t["rot"], trans=t["trans"],
                src_raw_pcd=t["raw_src_pcd"])

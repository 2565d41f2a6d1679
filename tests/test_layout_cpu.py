"""CPU: the engine-side parameter layout equals the reference's state_dict (keys, shapes, buffers)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_layout_matches_reference_capture():
    from roitr_amd.riga import state_dict_layout
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))
    mine = state_dict_layout(1)
    assert len(mine) == len(ref) == 522
    for (k, s, t), (rk, rs, rt) in zip(mine, ref):
        assert k == rk and list(s) == list(rs) and t == rt, (k, rk)
    assert sum(int(np.prod(s)) for k, s, t in mine if t == "param") == 10102531


def test_module_state_dict_keys_are_reference_keys():
    from roitr_amd.config import test_config
    from roitr_amd.riga import create_model
    m = create_model(test_config("3DMatch"))
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in ref]
    for k, s, _ in ref:
        assert list(sd[k].shape) == s
    # div_term buffers follow positional_encoding.py:43-45
    dt = sd["backbone.global_transformer.embedding.embedding.div_term"].numpy()
    np.testing.assert_allclose(dt, np.exp(np.arange(0, 256, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / 256)), rtol=1e-6)


def test_closed_form_weights_are_stable():
    from roitr_amd.weights import closed_form_param
    a = closed_form_param("coarse_proj.weight", (256, 256))
    assert a.dtype == np.float32 and a.shape == (256, 256)
    # pinned values: the golden fixtures were generated with exactly these weights
    np.testing.assert_allclose(a[0, :3], closed_form_param("coarse_proj.weight", (256, 256))[0, :3])
    assert abs(float(a.mean())) < 1e-3 and 0.03 < float(a.std()) < 0.04
    assert closed_form_param("OT.alpha", ()).shape == ()


def test_weight_signature_sees_every_kind_of_weight_change():
    """RIGA_v2._weights_signature() decides in front of every forward whether the engine's registered pointers / derived
    weights are stale (riga.py _ensure_engine).  It must cover every tensor of the state_dict and change on an in-place
    update (load_state_dict / copy_), a moved storage (what .cuda() / .to() do: param.data replaced) and a replaced
    Parameter object."""
    import torch
    from roitr_amd.config import test_config
    from roitr_amd.riga import create_model
    m = create_model(test_config("3DMatch"))
    m._holders = list(m.modules())
    sig0 = m._weights_signature()
    assert len(sig0) == len(m.state_dict()) == 522
    assert m._weights_signature() == sig0                       # stable while nothing changes
    sd = m.state_dict()
    k = "coarse_proj.weight"
    sd[k].copy_(torch.zeros_like(sd[k]))                        # in place (load_state_dict does exactly this)
    sig1 = m._weights_signature()
    assert sig1 != sig0
    m.coarse_proj.weight.data = torch.ones_like(m.coarse_proj.weight.data)   # new storage under the same Parameter
    sig2 = m._weights_signature()
    assert sig2 != sig1
    m.coarse_proj.weight = torch.nn.Parameter(torch.zeros(256, 256))         # a new Parameter object
    assert m._weights_signature() != sig2
    m.load_state_dict(m.state_dict())                           # a full reload bumps every version
    assert m._weights_signature() != sig2


def test_weights_frozen_skips_only_the_signature_walk():
    """weights_frozen = True (inference loops, bench.py) must not hide a missing engine: _ensure_engine still synchronises when
    there is no engine yet, and goes back to checking as soon as the flag is cleared."""
    from roitr_amd.config import test_config
    from roitr_amd.riga import create_model
    m = create_model(test_config("3DMatch"))
    calls = []
    m.sync_engine = lambda: calls.append(1) or setattr(m, "_engine", object()) or setattr(m, "_engine_sig", m._weights_signature())
    m._holders = list(m.modules())
    m.weights_frozen = True
    m._ensure_engine()                       # no engine yet: synchronises even when frozen
    assert len(calls) == 1
    import torch
    with torch.no_grad():
        m.coarse_proj.weight.add_(1.0)       # (an update through `.data` would not bump the version counter: invisible to any such check)
    m._ensure_engine()                       # frozen: the change is the caller's responsibility
    assert len(calls) == 1
    m.weights_frozen = False
    m._ensure_engine()                       # checking again: the stale registration is noticed
    assert len(calls) == 2

"""Shared helpers for the GPU parity tests."""
from roitr_amd.harness import build_model, pair_to_device  # noqa: F401


def golden_pair_inputs(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("in.")}

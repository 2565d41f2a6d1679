"""Shared helpers for the GPU parity tests."""
import os

from roitr_amd.harness import build_model as _build_model, pair_to_device  # noqa: F401


def build_model(benchmark="3DMatch", operand_dtype="f32", weights="plain"):
    """harness.build_model; ROITR_TEST_OPERAND_DTYPE=f32x3 (test infrastructure only) runs every fp32 model of the suite in the
    three-way-split mode instead (RoitrEngineConfig.operand_dtype = 2): the whole parity suite, unchanged tolerances, is that mode's test."""
    if operand_dtype == "f32" and os.environ.get("ROITR_TEST_OPERAND_DTYPE"):
        operand_dtype = os.environ["ROITR_TEST_OPERAND_DTYPE"]
    return _build_model(benchmark, operand_dtype=operand_dtype, weights=weights)


def golden_pair_inputs(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("in.")}

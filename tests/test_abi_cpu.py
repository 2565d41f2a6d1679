"""CPU: the C-ABI library loads and exports every symbol the headers declare (no compute calls)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_exports_declared_symbols():
    import __graft_entry__ as G
    from roitr_amd import _lib
    lib = _lib.lib()
    names = G.declared_symbols()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.roitr_abi_version() == 4   # bumped with every struct change of include/*.h (round 6: batch_live, the compacted patch layout; RoitrLocalBlock::w*_h)
    # the reference's own launcher names are present verbatim (cpp_wrappers/pointops/src/*/*_cuda_kernel.h)
    for n in ("furthestsampling_cuda_launcher", "knnquery_cuda_launcher", "grouping_forward_cuda_launcher",
              "grouping_backward_cuda_launcher", "interpolation_forward_cuda_launcher", "interpolation_backward_cuda_launcher",
              "subtraction_forward_cuda_launcher", "subtraction_backward_cuda_launcher", "aggregation_forward_cuda_launcher",
              "aggregation_backward_cuda_launcher"):
        assert hasattr(lib, n), n


def test_level_sizes_and_workspace_queries_are_host_only():
    from roitr_amd import _lib
    lib = _lib.lib()
    out = (ctypes.c_int * 4)()
    lib.roitr_level_sizes(5000, out)
    assert list(out) == [5000, 1250, 312, 78]       # model/model.py:59-62 floor rule
    lib.roitr_knn_workspace_bytes.restype = ctypes.c_size_t
    assert lib.roitr_knn_workspace_bytes(2, 10000, 10000) > 10000 * 16


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from roitr_amd import _lib, pointops
    x = torch.zeros(8, 3)
    o = torch.tensor([8], dtype=torch.int32)
    with pytest.raises(_lib.RoitrError):
        pointops.furthestsampling(x, o, torch.tensor([2], dtype=torch.int32))
    with pytest.raises(_lib.RoitrError):
        pointops.knnquery(3, x, x, o, o)


def test_product_never_imports_the_oracle():
    import re
    pkg = os.path.join(ROOT, "roitr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text and "pointops_cpu" not in text and "roitr_ref" not in text, f

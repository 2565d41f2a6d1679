import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_pair():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "pair_n1024.npz"))


@pytest.fixture(scope="session")
def golden_stages():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "stages.npz"))


_ORACLE_CACHE = {}


def oracle_forward(benchmark, n, config, pair_index, weights="selective", normals="field", cloud="uniform"):
    """(pair, CPU-oracle forward of it), cached for the session: several GPU test modules check the engine against the same
    full-size oracle result (an N = 8000 4DMatch forward is ~1 min of host time)."""
    key = (benchmark, n, config, pair_index, weights, normals, cloud)
    if key not in _ORACLE_CACHE:
        from oracle import roitr_ref as R  # checker only
        from roitr_amd.synthetic import make_pair
        fd = benchmark in ("4DMatch", "4DLoMatch")
        pair = make_pair(n, config=config, pair_index=pair_index, normals=normals, cloud=cloud)
        ref = R.forward(R.closed_form_state(2 if fd else 1, weights), pair, cfg=dict(R.FDMATCH_CFG) if fd else None,
                        threads=len(os.sched_getaffinity(0)))
        _ORACLE_CACHE[key] = (pair, ref)
    return _ORACLE_CACHE[key]

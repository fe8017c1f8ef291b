import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-depth x full-length parity runs (minutes each); they run only with MC_RUN_SLOW=1 (tools/gpu_session.sh <tag> pytest_slow)")


def tolerance_probe(key, value):
    """record the MEASURED margin behind a tolerance bar (gpurun_out/tolerance_probe.json on the GPU box; the round's copy is
    committed as profiles/rNN/tolerance_probe.json and the bars in the tests are set from it)"""
    import json
    try:
        path = os.path.join(ROOT, "gpurun_out", "tolerance_probe.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        json.dump(data, open(path, "w"), indent=1)
    except OSError:
        pass


def calib_errors(got, want):
    """max |difference| and max relative difference per calibration statistic (lists per step)"""
    import numpy as np
    out = {}
    for k in ("norm_ratio", "norm_std", "cos_dis"):
        g, w = np.asarray(got[k], dtype=np.float64), np.asarray(want[k], dtype=np.float64)
        out[k] = dict(max_abs=float(np.abs(g - w).max()), max_rel=float((np.abs(g - w) / np.maximum(np.abs(w), 1e-12)).max()),
                      min_abs_value=float(np.abs(w).min()))
    return out


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need the MI355X: without one they are skipped, not errored."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu-marked tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)

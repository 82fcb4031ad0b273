import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def stereo():
    z = np.load(os.path.join(GOLDEN, "stereo_pair_u8.npz"))
    return z["left"].astype(np.float32), z["right"].astype(np.float32)


@pytest.fixture(scope="session")
def ctx():
    from cudasift_amd import capi
    if capi.device_count() < 1:
        pytest.fail("no HIP device visible — GPU tests must run on the MI355X box (no CPU fallback)")
    c = capi.Context(0)
    yield c
    c.close()


_DIAG = {}


def record(name, **kw):
    """Collect per-test diagnostics; written to gpurun_out/parity_report.json at session end."""
    def conv(v):
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        if isinstance(v, np.ndarray):
            return v.tolist()
        return v
    _DIAG.setdefault(name, {}).update({k: conv(v) for k, v in kw.items()})


def pytest_sessionfinish(session, exitstatus):
    if _DIAG:
        try:                                   # diagnostics only: never let the report fail a green session
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_report.json"), "w") as f:
                json.dump(_DIAG, f, indent=1, sort_keys=True)
        except OSError:
            pass

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


def _cpu_budget():
    """CPUs this process may actually use: the GPU boxes show 256 logical CPUs behind a cgroup quota of 16.  The oracle's
    OpenMP team defaults to one spinning thread per LOGICAL CPU — 256 threads on 16 CPUs' worth of time made every
    orc.extract call take ~4 s whatever the image (r06: 55 of the 56 s of a tiny-image test; half the GPU suite's time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return n


# before liboracle.so / the emulated reference (libgomp) are loaded
os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_budget()))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("SIMT_THREADS", str(_cpu_budget()))


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


@pytest.fixture(autouse=True)
def _guard_bands_intact():
    """With MISIFT_GUARD=1 in the environment EVERY device allocation of the run is guarded (64 KiB bands, NaN-poisoned payload):
    the bands are verified after each test (buffers freed meanwhile were verified as they went), so the whole GPU suite doubles as
    an out-of-bounds check:  MISIFT_GUARD=1 python -m pytest tests -m gpu"""
    yield
    if os.environ.get("MISIFT_GUARD") == "1":
        from cudasift_amd import capi
        if capi.device_count() >= 1:
            capi.check_guards()


def pytest_sessionfinish(session, exitstatus):
    if _DIAG:
        try:                                   # diagnostics only: never let the report fail a green session
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_report.json"), "w") as f:
                json.dump(_DIAG, f, indent=1, sort_keys=True)
        except OSError:
            pass

"""The in-launch wait of the single-call path (dog_scan_all_kernel: coarse-level workgroups wait for the ScaleDown chain
that runs in the first workgroups of the SAME launch) and the completion contract of the synchronous calls.

  * the wait is bounded: with a bound of zero the flag is never raised (test mode), every waiting workgroup gives up, the
    host sees CNT_CHAINTMO with the counts, re-runs the call with a stand-alone chain launch and returns the right records;
  * under a reduced CU mask (8 / 32 of the 256 CUs) the embedded chain still completes — workgroups are dispatched in
    index order, so the chain (lowest indices) is resident before anything waits for it — and no fallback happens;
  * embedded vs stand-alone chain under a memory-bound co-runner, many iterations: same records every time (the ticket is
    drawn only after s_waitcnt vmcnt(0), advisor r04);
  * misift_extract's default completion is a full synchronisation: a plain copy on ANOTHER stream right after the call sees
    every record (r04 returned at the host flag by default)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import record
from synth import synth_frame

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, hashlib
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from cudasift_amd import capi
from synth import synth_frame
ctx = capi.Context(0)
out = []
for seed, w, h in ((3, 1920, 1080), (4, 1280, 960), (5, 1920, 1080)):
    img = synth_frame(seed, w, h)
    pts, n, cnt = ctx.extract(img, num_octaves=5, thresh=3.0)
    k = [pts[:n][f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")]
    out.append({"n": int(n), "cnt": cnt.tolist(), "sha": hashlib.sha256(pts[:n][np.lexsort(k)].tobytes()).hexdigest()})
print("RESULT " + json.dumps({"frames": out, "fallbacks": ctx.chain_fallbacks(), "fuse_fallbacks": ctx.fuse_fallbacks()}))
"""


def _child(env_extra):
    env = dict(os.environ, MISIFT_QUIET="0", MISIFT_TUNABLES="1", **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (env_extra, r.stdout[-1500:], r.stderr[-1500:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:]), r.stderr


@pytest.fixture(scope="module")
def baseline():
    res, _ = _child({"MISIFT_CHAIN_EMBED": "0"})
    assert res["fallbacks"] == 0
    return res


def test_expired_wait_falls_back_to_a_stand_alone_chain(baseline):
    res, err = _child({"MISIFT_CHAIN_WAIT_US": "0"})
    assert res["fallbacks"] == 1, res                       # the first call; the context then keeps the chain separate
    assert "in-launch wait for the ScaleDown chain expired" in err
    assert res["frames"] == baseline["frames"]
    record("chain_wait_fallback", fallbacks=res["fallbacks"])


def test_fused_orientation_and_descriptor_launch_equals_the_two_launches(baseline):
    """r06: orient_descr_fused_kernel — a single call's orientations and descriptors as ONE launch whose descriptor pass
    waits in the kernel for the coarser octaves' orientations.  Built, bit-identical, deadlock-free under CU masks, and
    SLOWER than the two launches, hence off by default: MISIFT_FUSE_ORIENT=1 selects it."""
    assert baseline["fuse_fallbacks"] == 0
    res, _ = _child({"MISIFT_FUSE_ORIENT": "1"})
    assert res["fuse_fallbacks"] == 0 and res["frames"] == baseline["frames"]
    record("fused_orient_descr", keypoints=[f["n"] for f in res["frames"]], fuse_fallbacks=res["fuse_fallbacks"])


def test_expired_fuse_wait_falls_back_to_two_launches(baseline):
    """A bound of zero expires before the first poll: every workgroup that has to wait skips its descriptors and raises
    CNT_FUSETMO, the host re-runs the call with separate launches and keeps them on that context."""
    res, err = _child({"MISIFT_FUSE_ORIENT": "1", "MISIFT_FUSE_WAIT_US": "0"})
    assert res["fuse_fallbacks"] == 1, res
    assert "fused orientation + descriptor kernel expired" in err
    assert res["frames"] == baseline["frames"]
    record("fuse_wait_timeout", fuse_fallbacks=res["fuse_fallbacks"])


@pytest.mark.parametrize("mask", ["0:0-7", "0:0-31"])
def test_fused_launch_under_a_reduced_cu_mask(baseline, mask):
    """Forward progress without co-residency: a workgroup waits only for workgroups with lower indices (8 CUs hold 32 of the
    launch's 1 024 workgroups).  r06's first version waited for all coarser octaves in every workgroup and sat here until
    the bound expired."""
    res, _ = _child({"MISIFT_FUSE_ORIENT": "1", "HSA_CU_MASK": mask})
    assert res["fallbacks"] == 0 and res["fuse_fallbacks"] == 0, res
    assert res["frames"] == baseline["frames"]
    record("fuse_wait_cu_mask/" + mask, fuse_fallbacks=res["fuse_fallbacks"])


@pytest.mark.parametrize("mask", ["0:0-7", "0:0-31"])
def test_embedded_chain_under_a_reduced_cu_mask(baseline, mask):
    res, _ = _child({"HSA_CU_MASK": mask})
    assert res["fallbacks"] == 0, res
    assert res["frames"] == baseline["frames"]
    record("chain_wait_cu_mask/" + mask, fallbacks=res["fallbacks"])


def test_embedded_chain_equals_stand_alone_under_load(ctx, baseline):
    """200 single-frame calls while a second context streams 64-frame batches (HBM-bound co-runner) on the same GPU."""
    import hashlib
    import threading
    from cudasift_amd import capi
    img = synth_frame(3, 1920, 1080)
    want = baseline["frames"][0]
    stop = threading.Event()

    def corunner():
        c2 = capi.Context(0)
        frames = np.stack([synth_frame(20 + i, 1920, 1080) for i in range(8)])
        while not stop.is_set():
            c2.extract_batch(frames, num_octaves=5, thresh=3.0, max_pts=8192)
        c2.close()
    t = threading.Thread(target=corunner)
    t.start()
    try:
        bad = 0
        for it in range(200):
            pts, n, cnt = ctx.extract(img, num_octaves=5, thresh=3.0)
            k = [pts[:n][f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")]
            sha = hashlib.sha256(pts[:n][np.lexsort(k)].tobytes()).hexdigest()
            bad += (n != want["n"]) or (cnt.tolist() != want["cnt"]) or (sha != want["sha"])
    finally:
        stop.set()
        t.join()
    record("chain_embed_under_load", iterations=200, mismatches=int(bad), fallbacks=ctx.chain_fallbacks())
    assert bad == 0 and ctx.chain_fallbacks() == 0


def test_default_completion_is_a_full_synchronisation():
    """A fresh context (early return off): right after misift_extract returns, a blocking hipMemcpy — the null stream, not
    the context's — must see every record (the advisor's r04 scenario).  torch is only the other stream's copy engine."""
    import ctypes as C
    from cudasift_amd import capi
    from oracle import pyoracle as orc
    c = capi.Context(0)
    img = synth_frame(9, 1920, 1080)
    ref, nref, _ = orc.extract(img, 5, 1.0, 3.0)
    src, p = c.upload_image(img)
    sc = capi.DevBuf(4 * capi.scratch_floats(1920, 1080, 5, False))
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    bad = 0
    for it in range(50):
        pts = c.zeros(576 * 32768)
        n = C.c_int(0)
        capi.check(capi.lib().misift_extract(c.h, src.ptr, 1920, 1080, p, 5, 1.0, 3.0, 0.0, 0, sc.ptr, pts.ptr, 32768, C.byref(n)))
        host = np.zeros(n.value, capi.POINT_DTYPE)
        assert hip.hipMemcpy(host.ctypes.data, pts.ptr, 576 * n.value, 2) == 0          # hipMemcpyDeviceToHost, null stream
        bad += int((host["subsampling"] == 0).sum() + (np.abs(host["data"]).sum(axis=1) == 0).sum())
        assert n.value == nref
    c.close()
    record("default_completion", iterations=50, unwritten_records=bad)
    assert bad == 0

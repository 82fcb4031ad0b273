"""The kernel / launch-structure variants of the extraction path, each on a FRESH context created under its switch, in
the suite the driver runs (VERDICT r2 "Missing" #5: the variant sweep used to be a separate script whose last kept log
showed a failure).  Every variant must return what the oracle returns — same records, same 17 counters — on a real image,
a 1080p synthetic frame, a batch, the timed entry point (packed, async) and in deterministic mode.

  MISIFT_FUSED=0        separate LaplaceMulti / extremum kernels (DoG planes in HBM); also the overflow re-run path
  MISIFT_TILE_DESCR=0   descriptor samples fetched from global memory instead of the LDS window
  MISIFT_TILE_ORIENT=1  orientation gradients from an LDS window
  MISIFT_GRAPH=1        hipGraph replay wherever a call repeats
  MISIFT_SPLIT_TAIL=0   all pyramid levels in one dog_scan launch on one stream
  MISIFT_BIN=0          no spatial binning of the detections before orientation / descriptor
"""
import os

import numpy as np
import pytest

from conftest import record
from synth import synth_frame
from util import compare_points

pytestmark = pytest.mark.gpu

VARIANTS = [{"MISIFT_FUSED": "0"}, {"MISIFT_TILE_DESCR": "0"}, {"MISIFT_TILE_ORIENT": "1"}, {"MISIFT_GRAPH": "1"},
            {"MISIFT_SPLIT_TAIL": "0"}, {"MISIFT_BIN": "0"}]


def _canon(recs):
    k = [recs[f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")]
    return recs[np.lexsort(k)].tobytes()


@pytest.fixture(params=VARIANTS, ids=lambda v: ",".join("%s=%s" % kv for kv in v.items()))
def vctx(request):
    from cudasift_amd import capi
    # (MISIFT_TUNABLES=1: the launch-shape / path variables are read only under this master switch — a production process
    #  ignores them; MISIFT_FUSED is a behaviour option and is always honoured)
    env = dict(request.param, MISIFT_TUNABLES="1")
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = capi.Context(0)               # the switches are read when the context is created
    finally:
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    c.variant = ",".join("%s=%s" % kv for kv in request.param.items())
    yield c
    c.close()


def test_variant_equals_oracle(vctx, stereo):
    from cudasift_amd import capi
    from oracle import pyoracle as orc
    c = vctx
    # 1. the reference's own sample image, mainSift.cpp parameters
    ref, nref, cref = orc.extract(stereo[0], 5, 1.0, 4.5)
    for rep in range(2):                  # twice: MISIFT_GRAPH replays the second call
        got, n, cnt = c.extract(stereo[0], num_octaves=5, init_blur=1.0, thresh=4.5)
        assert n == nref and np.array_equal(cnt, cref), (c.variant, rep)
        compare_points(ref[:nref], got[:n], "variant[%s]/left/%d" % (c.variant, rep), record)
    # 2. one synthetic 1080p frame
    img = synth_frame(11)
    ref, nref, cref = orc.extract(img, 5, 1.0, 3.0)
    got, n, cnt = c.extract(img, num_octaves=5, init_blur=1.0, thresh=3.0)
    assert n == nref and np.array_equal(cnt, cref), c.variant
    compare_points(ref[:nref], got[:n], "variant[%s]/synth1080" % c.variant, record)
    # 3. a batch (odd sizes: the generic, width % 4 != 0 kernels)
    imgs = np.stack([synth_frame(300 + f, 483, 270) for f in range(3)])
    bp, bn = c.extract_batch(imgs, num_octaves=4, thresh=2.5, max_pts=8192)
    for f in range(3):
        ref, nref, _ = orc.extract(imgs[f], 4, 1.0, 2.5, max_pts=8192)
        assert bn[f] == nref, (c.variant, f)
        compare_points(ref[:nref], bp[f, :nref], "variant[%s]/batch_odd_f%d" % (c.variant, f), record)
    # 4. the timed entry point: packed, async, d_pts = NULL, 8 x 1080p (>= 8 frames: the split-tail branch unless switched off)
    B, mp = 8, 8192
    frames = np.stack([synth_frame(7300 + f) for f in range(B)])
    d = c.upload(frames)
    scratch = capi.DevBuf(4 * capi.scratch_floats(1920, 1080, 5, False) * B)
    cntb = c.zeros(4 * (2 * B + 1))
    packed = c.upload(np.full(576 * mp * B, 0xA5, np.uint8))
    fused_now = c.get_options().fused
    dpts = None if fused_now else c.zeros(576 * mp * B)        # the dense path shares one record array
    for rep in range(2):
        capi.check(capi.lib().misift_extract_batch_packed_async(c.h, d.ptr, B, 1920 * 1080, 1920, 1080, 1920, 5, 1.0, 3.0, 0.0,
                                                                scratch.ptr, dpts.ptr if dpts is not None else None,
                                                                mp, cntb.ptr, cntb.ptr + 4 * B, packed.ptr),
                   "misift_extract_batch_packed_async")
        c.sync()
    cb = c.download(cntb, (2 * B + 1,), np.int32)
    counts, offs = cb[:B], cb[B:]
    recs = c.download(packed, (int(offs[B]),), capi.POINT_DTYPE)
    ref, nref, _ = orc.extract_batch(frames, 5, 1.0, 3.0, max_pts=mp)
    assert np.array_equal(counts, nref), (c.variant, counts, nref)
    for f in range(B):
        compare_points(ref[f, :nref[f]], recs[offs[f]:offs[f + 1]], "variant[%s]/timed_f%d" % (c.variant, f), record)
    # 5. deterministic mode: byte-identical arrays run to run, same set
    c.set_options(deterministic=1)
    try:
        a, an = c.extract_batch(imgs, num_octaves=4, thresh=2.5, max_pts=8192)
        b, bn2 = c.extract_batch(imgs, num_octaves=4, thresh=2.5, max_pts=8192)
    finally:
        c.set_options(deterministic=0)
    assert np.array_equal(an, bn) and np.array_equal(bn2, bn)
    for f in range(3):
        assert a[f, :an[f]].tobytes() == b[f, :an[f]].tobytes(), (c.variant, f)
        assert _canon(a[f, :an[f]]) == _canon(bp[f, :bn[f]]), (c.variant, f)
    record("variants", **{c.variant: "ok"})

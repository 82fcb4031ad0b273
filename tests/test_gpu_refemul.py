"""The HIP path against the reference's OWN kernels and host code on FRESH inputs, in the driver's GPU run: the emulated
reference (oracle/_ref/libcudasift_refemul_fast.so — cudaSiftH.cu + cudaSiftD.cu compiled by oracle/build_ref.sh against the
CPU SIMT emulator where /root/reference exists) is a prebuilt .so that travels to the GPU box with the snapshot.  No oracle
in between, no committed vectors: both sides compute here.  (tools/hip_vs_refemul.py is the 544 385-record version.)"""
import os

import numpy as np
import pytest

from conftest import record
from synth import synth_frame
from util import compare_with_reference

pytestmark = pytest.mark.gpu
os.environ.setdefault("SIMT_THREADS", "16")     # the GPU boxes show 256 logical CPUs behind a 16-CPU quota


def _ref():
    from oracle import pyrefemul
    if not pyrefemul.available("fast"):
        pytest.skip("oracle/_ref/libcudasift_refemul_fast.so not built (needs /root/reference at build time)")
    return pyrefemul


@pytest.mark.parametrize("seed,w,h,noct,th", [(100, 1920, 1080, 5, 3.0), (101, 1920, 1080, 5, 3.0), (102, 1280, 960, 5, 2.5),
                                              (103, 1000, 750, 4, 2.0), (104, 641, 479, 6, 2.0)])
def test_hip_equals_emulated_reference_on_fresh_frames(ctx, seed, w, h, noct, th):
    ref = _ref()
    img = synth_frame(seed, w, h)
    r_pts, r_n, r_cnt = ref.extract(img, noct, 1.0, th, flavour="fast")
    pts, n, cnt = ctx.extract(img, num_octaves=noct, init_blur=1.0, thresh=th)
    assert n == r_n
    compare_with_reference(pts, cnt, r_pts, r_cnt, noct, "hip_vs_emulated_reference/seed%d_%dx%d" % (seed, w, h), "ulp", record,
                           flip_budget=2)


def test_hip_matcher_equals_emulated_reference(ctx):
    """MatchSiftData on real descriptors of two fresh frames: the reference's FindMaxCorr10 on the emulator vs match_kernel."""
    ref = _ref()
    a, na, _ = ctx.extract(synth_frame(110, 960, 540), thresh=2.0)
    b, nb, _ = ctx.extract(synth_frame(111, 960, 540), thresh=2.0)
    assert na > 300 and nb > 300
    want = a[:na].copy()
    ref.match(want, na, b[:nb].copy(), nb, "fast")
    got = ctx.match(a[:na].copy(), na, b[:nb].copy(), nb)
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert np.array_equal(got[f], want[f]), f

"""The HIP path against GOLDEN VECTORS PRODUCED BY THE REFERENCE ITSELF (tests/golden/refemul_golden.npz).

The file holds outputs of the reference's own kernels and host code (cudaSiftH.cu + cudaSiftD.cu + matching.cu compiled
by oracle/build_ref.sh against the CPU SIMT emulator, -ffp-contract=fast flavour; generator:
tests/golden/make_fixtures.py).  No oracle in between: every assertion here compares what libmisift.so computes on
the MI355X with what the reference's code computed.
  dense stages (LowPassBlock, ScaleDown, LaplaceMultiMem)        the same bits (sha256 of the arrays)
  ExtractSift (crop of left.pgm, left.pgm)                        17 counters and keypoint set identical; x, y to
      1.5e-7, scale/sharpness/edgeness to 5e-7 (1-2 ulp: libm vs written-out exp2, contraction of the refinement),
      orientation <= 0.036 deg, descriptors see util.compare_with_reference
  MatchSiftData (1000 x 1500, n2 % 32 != 0), FindHomography       the same bits
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import record
from synth import descriptors_to_points, synth_descriptors, synth_matches
from util import compare_with_reference

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "refemul_golden.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_dense_stages_equal_reference_kernels(ctx, stereo, golden):
    crop = stereo[0][300:540, 400:720].copy()
    low = ctx.lowpass(crop, 1.0)
    assert sha(low) == str(golden["sha_lowpass"])
    assert sha(ctx.scaledown(low)) == str(golden["sha_scaledown"])
    assert sha(ctx.laplace(low, 5, 5)) == str(golden["sha_laplace"])
    lo2, down = ctx.lowpass_scaledown(crop, 1.0)                   # the fused kernel of the timed path
    assert sha(lo2) == str(golden["sha_lowpass"]) and sha(down) == str(golden["sha_scaledown"])
    odd = crop[:37, :131].copy()                                   # width % 4 != 0: the generic path
    assert sha(ctx.lowpass(odd, 1.3)) == str(golden["sha_lowpass_odd"])
    assert sha(ctx.scaledown(odd)) == str(golden["sha_scaledown_odd"])
    assert sha(ctx.laplace(odd, 5, 3)) == str(golden["sha_laplace_odd"])


def _golden_image(stereo, name):
    crop = stereo[0][300:540, 400:720].copy()
    if name in ("crop", "crop_up"):
        return crop
    if name == "wide":         # 1920x1080: left.pgm mirrored outwards (tests/golden/make_fixtures.py::wide_1080p)
        return np.pad(stereo[0], ((60, 60), (320, 320)), mode="reflect").astype(np.float32)
    return stereo[0] if name == "left" else stereo[1]


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("name,noct,th,up,flips", [("crop", 4, 3.5, False, 0), ("left", 5, 4.5, False, 0),
                                                   ("wide", 5, 3.0, False, 2), ("righ", 5, 4.5, False, 2),
                                                   ("crop_up", 4, 3.5, True, 1)])
def test_extract_equals_reference_kernels(ctx, stereo, golden, name, noct, th, up, flips, fused):
    """`wide` is the bench workload's shape and parameters (1920x1080, 5 octaves, initBlur 1.0, thresh 3.0)."""
    img = _golden_image(stereo, name)
    saved = ctx.get_options()
    ctx.set_options(fused=fused)
    try:
        pts, n, cnt = ctx.extract(img, num_octaves=noct, init_blur=1.0, thresh=th, scale_up=up)
    finally:
        ctx.set_options(fused=saved.fused)
    assert n == int(golden[name + "_n"])
    compare_with_reference(pts, cnt, golden[name + "_records"], golden[name + "_counters"], noct,
                           "hip_vs_reference_golden/%s_fused%d" % (name, fused), "ulp", record, flip_budget=flips, desc_stride=4 if name in ("wide", "righ", "crop_up") else 1,
                           img=img, scale_up=up)


def test_match_equals_reference_kernel(ctx, golden):
    from cudasift_amd.capi import POINT_DTYPE
    a = descriptors_to_points(synth_descriptors(1000, 7), POINT_DTYPE)
    b = descriptors_to_points(synth_descriptors(1500, 8), POINT_DTYPE)
    got = ctx.match(a, 1000, b, 1500)
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert np.array_equal(got[f], golden["match_" + f]), f


def test_find_homography_equals_reference_kernels(ctx, golden):
    from cudasift_amd.capi import POINT_DTYPE
    from oracle import pyoracle as orc                              # only for srand(): the libc state both sides draw from
    m, _, _ = synth_matches(3000, seed=5, dtype=POINT_DTYPE)
    d = ctx.upload(m)
    orc.srand(1)
    H, nm = ctx.find_homography(d.ptr, 3000, num_loops=2000, min_score=0.85, max_ambiguity=0.95, thresh=5.0)
    assert nm == int(golden["homography_inliers"]) and np.array_equal(H, golden["homography_H"])

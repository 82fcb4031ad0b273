"""The oracle pinned to the reference's OWN code.

oracle/build_ref.sh compiles the reference's cudaImage.cu + cudaSiftH.cu (+ cudaSiftD.cu: all 25 kernels) +
matching.cu where they lie against a CPU SIMT emulation (oracle/simt_emul.h: fibers for threads, barriers, 32-lane
warp shuffles/votes, atomics, __shared__, textures with 1.8 fixed-point weights), twice: `-ffp-contract=fast` (g++
fuses multiply-adds, like a -fmad=true CUDA compile may) and `-ffp-contract=off`.  What runs is the reference's own
control flow, tilings, constants, shared-memory protocols and counter protocol; what stays hardware-defined (the
fast-math intrinsics, which FMAs nvcc forms) is mapped to libm / bracketed by the two flavours.

Three layers:
  1. the emulator itself against closed forms (kernels of our own, oracle/simt_selftest.cu);
  2. oracle/sift_oracle.c against the emulated reference, live (needs oracle/_ref: built where /root/reference exists);
  3. oracle/sift_oracle.c against tests/golden/refemul_golden.npz, vectors the emulated reference produced
     (tests/golden/make_fixtures.py) — runs anywhere, the GPU suite checks the HIP path against the same file.
"""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import pyrefemul as ref
from synth import descriptors_to_points, synth_descriptors, synth_frame, synth_matches
import util
from util import compare_with_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
needs_ref = pytest.mark.skipif(not ref.available("fast"), reason="oracle/_ref not built (needs /root/reference at build time)")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------ 1. the emulator itself
@pytest.fixture(scope="module")
def st():
    path = os.path.join(ROOT, "oracle", "libsimt_selftest.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libsimt_selftest.so"])
    L = C.CDLL(path)
    L.st_fmul_rz.restype = C.c_float
    L.st_fmul_rz.argtypes = [C.c_float, C.c_float]
    L.st_f2i.argtypes = [C.c_float]
    return L


def test_emulator_warp_collectives(st):
    nb, nt = 3, 96                                  # 3 warps per block; tid 77 sits in block 0, warp 2
    n = nb * nt
    out = np.zeros((6, n), np.int32)
    st.st_warp(out.ctypes.data_as(C.c_void_p), nb, nt)
    tid = np.arange(n)
    lane = (tid % nt) % 32
    v = 1000 + tid
    assert np.array_equal(out[0], np.where(lane + 3 < 32, v + 3, v))            # shfl_down: out of range -> own value
    assert np.array_equal(out[1], np.where(lane - 2 >= 0, v - 2, v))            # shfl_up
    assert np.array_equal(out[2], v - lane + 5)                                 # shfl idx 5
    assert np.array_equal(out[3], np.where(lane % 8 + 1 < 8, v + 1, v))         # width-8 segments
    warp_of_77 = (tid // 32) == (77 // 32)
    assert np.array_equal(out[4], warp_of_77.astype(np.int32))                  # any: per warp
    assert (out[5].view(np.uint32) == 0xAAAAAAAA).all()                         # ballot of odd lanes


def test_emulator_barriers_shared_memory_and_thread_indices(st):
    nb, bx, by = 5, 16, 8
    n = bx * by
    rng = np.random.default_rng(0)
    a = rng.integers(0, 100, nb * n).astype(np.int32)
    out = np.zeros_like(a)
    tids = np.zeros_like(a)
    st.st_scan(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), tids.ctypes.data_as(C.c_void_p), nb, bx, by)
    assert np.array_equal(out.reshape(nb, n), np.cumsum(a.reshape(nb, n), axis=1))
    t = np.arange(n)
    want = (t % bx + 100 * (t // bx))[None, :] + 10000 * np.arange(nb)[:, None]
    assert np.array_equal(tids.reshape(nb, n), want)


def test_emulator_atomics_across_blocks(st):
    cnt = np.zeros(4, np.uint32)
    fsum = C.c_float(0)
    nb, nt = 37, 100
    st.st_atomics(nb, nt, cnt.ctypes.data_as(C.c_void_p), C.byref(fsum))
    assert cnt[0] == nb * nt and cnt[1] == nb * nt - 1 and cnt[2] == (nb * nt) % 10
    assert fsum.value == 0.5 * nb * nt


def test_emulator_exited_threads_do_not_block(st):
    out = np.full(64, -1, np.int32)
    st.st_exit(out.ctypes.data_as(C.c_void_p))
    t = np.arange(64)
    v = t ^ 1
    want = np.where(t >= 48, v + np.where(t + 4 < 64, v + 4, v), v)
    want[32:48] = -1
    assert np.array_equal(out, want)


def test_emulator_texture_unit_and_conversions(st):
    rng = np.random.default_rng(1)
    h, w, pitch = 9, 13, 16
    img = np.zeros((h, pitch), np.float32)
    img[:, :w] = rng.uniform(0, 255, (h, w)).astype(np.float32)
    xy = np.concatenate([rng.uniform(-3, w + 3, (200, 1)), rng.uniform(-3, h + 3, (200, 1))], axis=1).astype(np.float32)
    xy[:4] = [[0.5, 0.5], [1.0, 0.5], [w - 0.5, h - 0.5], [3.25, 4.75]]
    out = np.zeros(200, np.float32)
    st.st_tex(img.ctypes.data_as(C.c_void_p), w, h, pitch, xy.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 200)
    # CUDA programming guide, linear filtering: xB = x - 0.5, weights in 1.8 fixed point, clamp addressing
    xb, yb = xy[:, 0] - np.float32(0.5), xy[:, 1] - np.float32(0.5)
    i, j = np.floor(xb), np.floor(yb)
    a = np.rint((xb - i).astype(np.float32) * 256) / 256
    b = np.rint((yb - j).astype(np.float32) * 256) / 256
    i0, i1 = np.clip(i, 0, w - 1).astype(int), np.clip(i + 1, 0, w - 1).astype(int)
    j0, j1 = np.clip(j, 0, h - 1).astype(int), np.clip(j + 1, 0, h - 1).astype(int)
    I = img.astype(np.float64)
    want = (1 - a) * (1 - b) * I[j0, i0] + a * (1 - b) * I[j0, i1] + (1 - a) * b * I[j1, i0] + a * b * I[j1, i1]
    assert np.array_equal(out, want.astype(np.float32))
    assert out[0] == img[0, 0] and out[2] == img[h - 1, w - 1]                     # texel centres
    assert out[1] == np.float32(0.5 * (float(img[0, 0]) + float(img[0, 1])))
    # the oracle's texture fetch is the same function up to the rounding of the blend (fp32 fmaf chain vs exact)
    got = np.array([orc.tex2d(img[:, :w], float(x), float(y)) for x, y in xy], np.float32)
    assert np.abs(got - out).max() <= 4e-5
    # __fmul_rz and the GPU's float -> int conversion
    for x, y in ((1.1, 3.3), (-1.1, 3.3), (1e-3, 7.7), (123456.7, 0.3333)):
        p = float(np.float32(x)) * float(np.float32(y))
        r = st.st_fmul_rz(x, y)
        away = np.nextafter(np.float32(r), np.float32(np.copysign(np.inf, r)), dtype=np.float32)
        assert abs(r) <= abs(p) < abs(float(away))
    assert st.st_f2i(float("nan")) == 0 and st.st_f2i(3e9) == 2**31 - 1 and st.st_f2i(-3e9) == -2**31 and st.st_f2i(-7.9) == -7


# ------------------------------------------------------------------ 2. oracle vs the emulated reference, live
@needs_ref
@pytest.mark.parametrize("shape", [(240, 320), (37, 131), (64, 256), (33, 17)])
def test_dense_stages_bit_identical_to_reference_kernels(stereo, shape):
    """LowPassBlock, ScaleDown, LaplaceMultiMem (cudaSiftD.cu:1986-2037, 84-168, 1753-1793) compiled with
    contraction: the oracle's explicit fmaf chains are the SAME bits.  Without contraction: within 1e-4."""
    h, w = shape
    img = stereo[0][300:300 + h, 400:400 + w].copy()
    for sigma in (1.0, 1.3):
        assert np.array_equal(ref.lowpass(img, sigma, "fast"), orc.lowpass(img, sigma))
    low = orc.lowpass(img, 1.0)
    assert np.array_equal(ref.scaledown(low, "fast"), orc.scaledown(low))
    for octave in (5, 3, 1):
        assert np.array_equal(ref.laplace(low, 5, octave, "fast"), orc.laplace(low, 5, octave))
    assert np.array_equal(ref.laplace_taps(5, "fast"), orc.laplace_taps(5))
    assert np.abs(ref.lowpass(img, 1.0, "off") - orc.lowpass(img, 1.0)).max() <= 1e-4
    assert np.abs(ref.scaledown(low, "off") - orc.scaledown(low)).max() <= 1e-4
    assert np.abs(ref.laplace(low, 5, 5, "off") - orc.laplace(low, 5, 5)).max() <= 1e-4


@needs_ref
@pytest.mark.parametrize("noct", [1, 2, 3, 4, 5, 6, 7])
def test_laplace_taps_equal_reference_host_code(noct):
    """PrepareLaplaceKernels (cudaSiftH.cu:439-458) is HOST code: never contracted, in either flavour of the emulated
    build (r03: the contraction flavour used to fuse it and differed in the taps of a 6th / 7th octave)."""
    for fl in ("fast", "off"):
        assert np.array_equal(ref.laplace_taps(noct, fl), orc.laplace_taps(noct)), (noct, fl)


@needs_ref
@pytest.mark.parametrize("case", ["left", "right_crop", "synth1080", "synth_odd"])
def test_extract_pinned_to_reference_kernels(stereo, case):
    """ExtractSift end to end: the reference's host code + six live kernels on the emulator vs the oracle.
    Counters, keypoint set: identical.  Fields: same bits (positions, edgeness) or 1-2 ulp (libm vs det_*)."""
    from conftest import record
    if case == "left":
        img, noct, th = stereo[0], 5, 4.5
    elif case == "right_crop":
        img, noct, th = stereo[1][100:580, 200:840].copy(), 4, 2.0
    elif case == "synth1080":
        img, noct, th = synth_frame(0), 5, 3.0
    else:
        img, noct, th = synth_frame(5, 333, 251), 3, 2.5
    r_pts, r_n, r_cnt = ref.extract(img, noct, 1.0, th, flavour="fast")
    orc.stats_reset()
    with orc.contract(1):          # (the comparison too: the explanation model of util.descriptor_tail_bound follows the mode)
        o_pts, o_n, o_cnt = orc.extract(img, noct, 1.0, th)
        assert o_n == r_n
        compare_with_reference(o_pts, o_cnt, r_pts, r_cnt, noct, "refemul_fast_vs_oracle_nvcc/" + case, "bits", record, img=img)
    # the mode every HIP parity test uses (no contraction outside the filters): same set, values within 3e-7
    orc.stats_reset()
    o_pts, o_n, o_cnt = orc.extract(img, noct, 1.0, th)
    compare_with_reference(o_pts, o_cnt, r_pts, r_cnt, noct, "refemul_fast_vs_oracle_plain/" + case, "ulp", record, img=img)
    # the reference built with every product rounded: same counters and set, values within the rounding of the filters
    r_pts, r_n, r_cnt = ref.extract(img, noct, 1.0, th, flavour="off")
    compare_with_reference(o_pts, o_cnt, r_pts, r_cnt, noct, "refemul_off_vs_oracle_plain/" + case, "", record)


@needs_ref
def test_extract_scaleup_and_lowest_scale_pinned(stereo):
    img = stereo[0][200:440, 300:620].copy()
    r_pts, r_n, r_cnt = ref.extract(img, 4, 1.0, 3.0, lowest_scale=1.5, scale_up=True, flavour="fast")
    orc.stats_reset()
    with orc.contract(1):
        o_pts, o_n, o_cnt = orc.extract(img, 4, 1.0, 3.0, lowest_scale=1.5, scale_up=True)
        assert o_n == r_n and o_n > 50
        compare_with_reference(o_pts, o_cnt, r_pts, r_cnt, 4, "refemul_scaleup", "bits", img=img, scale_up=True)


@needs_ref
@pytest.mark.parametrize("noct,blur,th,ls", [(5, 0.0, 3.0, 0.0), (5, 0.5, 2.0, 0.0), (4, 2.0, 1.0, 0.0), (1, 1.0, 2.0, 0.0),
                                             (6, 1.0, 1.5, 0.0), (3, 1.0, 2.0, 3.0), (2, 1.0, 0.2, 0.0)])
def test_extract_parameter_sweep_pinned(stereo, noct, blur, th, ls):
    """The ExtractSift argument surface on a 320x240 crop: initBlur 0 (the reference clamps sigma to 0.001: a delta
    kernel), 0.5, 2.0; 1 and 6 octaves (coarsest level 10x7 px); lowestScale > 0; a threshold of 0.2 (1 042 keypoints in
    320x240; the reference's 32-candidates-per-tile cap is still not reached — orc.stats()['tile_overflows'] == 0 even on
    white noise).  Counters and keypoint set identical, positions the same bits."""
    from conftest import record
    img = stereo[0][200:440, 300:620].copy()
    r_pts, r_n, r_cnt = ref.extract(img, noct, blur, th, lowest_scale=ls, flavour="fast")
    orc.stats_reset()
    with orc.contract(1):
        o_pts, o_n, o_cnt = orc.extract(img, noct, blur, th, lowest_scale=ls)
    assert o_n == r_n and o_n > 40
    compare_with_reference(o_pts, o_cnt, r_pts, r_cnt, noct, "refemul_sweep/o%d_b%g_t%g_l%g" % (noct, blur, th, ls), "bits",
                           record, img=img, init_blur=blur)


@needs_ref
@pytest.mark.parametrize("w,h,noct,th", [(64, 48, 3, 1.0), (48, 36, 4, 0.5), (40, 30, 2, 0.5), (31, 17, 1, 0.3), (16, 16, 1, 0.1),
                                         (9, 9, 1, 0.1), (33, 65, 3, 0.5), (257, 19, 4, 0.5), (12, 200, 2, 0.5)])
def test_extract_tiny_images_pinned(w, h, noct, th):
    """White-noise images down to 9x9 (pyramid levels down to 6x4): every clamp path of the reference's kernels at once.
    numPts and detection counters identical, duplicate counters within one keypoint, positions to 3e-7."""
    from util import compare_tiny
    img = np.random.default_rng(9 + w).uniform(0, 255, (h, w)).astype(np.float32)
    r_pts, r_n, r_cnt = ref.extract(img, noct, 1.0, th, flavour="fast")
    with orc.contract(1):
        o_pts, o_n, o_cnt = orc.extract(img, noct, 1.0, th)
    compare_tiny(o_pts, o_n, o_cnt, r_pts, r_n, r_cnt, noct)


@needs_ref
def test_capacity_overflow_against_reference(stereo):
    """maxPts too small (a documented deviation): the reference clamps every overflowing detection to
    slot maxPts-1 (cudaSiftD.cu:1421 — whichever thread writes last stays; order-dependent on a GPU) and drops overflowing
    duplicates (:1044); we drop both.  numPts is min(count, maxPts) on both sides, the records that fit are mostly the
    same keypoints, and with maxPts == the number of points everything is identical again."""
    img = stereo[0]
    for mp, common in ((500, 0.9), (100, 0.7)):
        r_pts, r_n, _ = ref.extract(img, 5, 1.0, 4.5, max_pts=mp, flavour="fast")
        with orc.contract(1):
            o_pts, o_n, _ = orc.extract(img, 5, 1.0, 4.5, max_pts=mp)
        assert r_n == mp and o_n == mp
        from util import associate
        ia, _, _, _ = associate(o_pts[:mp], r_pts[:mp])
        assert len(ia) >= common * mp, (mp, len(ia))
    r_pts, r_n, r_cnt = ref.extract(img, 5, 1.0, 4.5, max_pts=1453, flavour="fast")
    with orc.contract(1):
        o_pts, o_n, o_cnt = orc.extract(img, 5, 1.0, 4.5, max_pts=1453)
    assert r_n == o_n == 1453 and np.array_equal(r_cnt, o_cnt)


@needs_ref
def test_32_candidates_per_block_cap_against_reference():
    """FindPointsMultiNew keeps at most MEMWID = 32 extrema per 30 x 8 block and scale (`pos<MEMWID`, cudaSiftD.cu:1369-1375)
    and silently drops the rest — a documented deviation: ours keeps every
    extremum.  Natural images never come near 32 per 240 pixels (white noise averages ~18), so the case is synthetic: a DoG
    stack with a peak on every (2, 3)-pixel lattice point.  The reference's count is exactly sum(min(n_block, 32)), ours is
    the lattice; what the reference keeps is a subset of ours with identical fields."""
    from synth import dense_extrema_dog
    from util import associate, rel_err
    dog, ys, xs = dense_extrema_dog()
    per_block = {}
    for y, x in zip(ys, xs):
        per_block[(x // 30, y // 8)] = per_block.get((x // 30, y // 8), 0) + 1
    assert max(per_block.values()) > 32
    capped = sum(min(v, 32) for v in per_block.values())
    r_pts, r_n = ref.findpoints(dog, 1.0, flavour="fast")
    o_pts, o_n = orc.findpoints(dog, 1.0)
    assert o_n == len(ys) == 1240 and r_n == capped == 1044, (o_n, r_n, capped)
    ia, ib, only_o, only_r = associate(o_pts[:o_n], r_pts[:r_n])
    assert len(only_r) == 0 and len(only_o) == o_n - r_n                # the reference's records are a subset of ours
    for f in ("xpos", "ypos", "scale", "sharpness", "edgeness"):
        assert rel_err(o_pts[:o_n][ia][f], r_pts[:r_n][ib][f]).max() <= 5e-7, f
    # ... and with the cap switched ON (r05: options.reference_cap / orc.reference_cap — "identical results" has a mode
    # on dense extrema too): the SAME 1 044 records, i.e. the same choice of which 32 a block keeps (by column, then row)
    with orc.reference_cap(1):
        c_pts, c_n = orc.findpoints(dog, 1.0)
    assert c_n == r_n == 1044
    ia, ib, only_c, only_r = associate(c_pts[:c_n], r_pts[:r_n])
    assert len(only_c) == 0 and len(only_r) == 0
    for f in ("xpos", "ypos", "scale", "sharpness", "edgeness"):
        assert rel_err(c_pts[:c_n][ia][f], r_pts[:r_n][ib][f]).max() <= 5e-7, f


@needs_ref
def test_matcher_pinned_to_reference_kernel(stereo):
    """MatchSiftData = CleanMatches + FindMaxCorr10 (matching.cu:289-397) on the emulator: score, ambiguity (the lossy
    8-class runner-up merge), match, match_xpos/ypos are the oracle's bits, incl. the n2 % 32 truncation."""
    p1, n1, _ = orc.extract(stereo[0][:480, :640], 5, 1.0, 3.0)
    p2, n2, _ = orc.extract(stereo[1][:480, :640], 5, 1.0, 3.0)
    for m1, m2 in ((n1, n2), (100, 63), (33, 64), (1, 32), (77, 1000)):
        a, b = p1.copy(), p1.copy()
        ref.match(a, m1, p2, m2, "fast")
        orc.match(b, m1, p2, m2)
        for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
            assert np.array_equal(a[f][:m1], b[f][:m1]), (m1, m2, f)
        a = p1.copy()
        ref.match(a, m1, p2, m2, "off")                      # products rounded separately: same winners, scores to 1e-6
        assert np.array_equal(a["match"][:m1], b["match"][:m1])
        assert np.allclose(a["score"][:m1], b["score"][:m1], rtol=2e-6, atol=0)


@needs_ref
def test_matcher_with_nan_inf_zero_and_negative_descriptors():
    """The reference's own descriptors can be NaN (Appendix B #7): a NaN row matches nothing (score 0, match -1), a NaN or
    inf column never wins, all-negative rows never win.  Oracle == the emulated FindMaxCorr10, NaNs included."""
    DT = orc.POINT_DTYPE
    n1, n2 = 300, 416
    a = descriptors_to_points(synth_descriptors(n1, 5, l2=True), DT)
    b = descriptors_to_points(synth_descriptors(n2, 6, l2=True), DT)
    a["data"][7] = np.nan; a["data"][8, 5] = np.nan; b["data"][100] = np.nan; b["data"][200, 3] = np.inf
    a["data"][9] = 0; b["data"][300] = 0; a["data"][10] *= -1
    r = a.copy(); ref.match(r, n1, b.copy(), n2, "fast")
    o = a.copy(); orc.match(o, n1, b.copy(), n2)
    for f in ("score", "ambiguity", "match_xpos", "match_ypos"):
        assert np.array_equal(r[f], o[f], equal_nan=True), f
    assert np.array_equal(r["match"], o["match"])
    assert (o["match"][7:11] == -1).all() and (o["score"][7:11] == 0).all()


@needs_ref
def test_find_homography_pinned_to_reference_kernels():
    """FindHomography (matching.cu:1000-1087: host rand() sampling, ComputeHomographies, TestHomographies with
    __fmul_rz) on the emulator vs the oracle: same 8 coefficients, same inlier count."""
    m, _, _ = synth_matches(3000, seed=5, dtype=orc.POINT_DTYPE)
    for loops, seed in ((2000, 1), (500, 7)):
        H, nm = ref.find_homography(m, 3000, loops, 0.85, 0.95, 5.0, seed=seed, flavour="fast")
        orc.srand(seed)
        Ho, no, _ = orc.find_homography(m, 3000, loops, 0.85, 0.95, 5.0)
        assert nm == no and np.array_equal(H, Ho)


@needs_ref
@pytest.mark.parametrize("n", [0, 5, 7, 8, 9, 40])
def test_find_homography_few_points_pinned(n):
    """matching.cu:1016-1017: fewer than 8 valid points -> identity and 0 matches; from 8 on the hypotheses draw repeated
    points.  The oracle follows the emulated reference bit for bit through the boundary."""
    m, _, _ = synth_matches(max(n, 1), seed=3, dtype=orc.POINT_DTYPE)
    m = m[:n]
    m["score"], m["ambiguity"] = 0.99, 0.1
    Hr, nr = ref.find_homography(m.copy(), n, 100, 0.85, 0.95, 5.0, seed=1, flavour="fast")
    orc.srand(1)
    Ho, no, _ = orc.find_homography(m.copy(), n, 100, 0.85, 0.95, 5.0)
    assert nr == no and np.array_equal(Hr, Ho)
    if n < 8:
        assert no == 0 and np.array_equal(Ho, np.eye(3, dtype=np.float32))


@needs_ref
def test_reference_demo_program_runs_on_the_emulator(tmp_path, stereo):
    """mainSift.cpp + geomFuncs.cpp + the emulated library: the reference's whole program without a GPU.  Its feature
    counts are the oracle's; its match counts are what the golden file recorded."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cudasift_refemul_main")
    z = np.load(os.path.join(GOLDEN, "stereo_pair_u8.npz"))
    os.makedirs(tmp_path / "data")
    for nm, k in (("left", "left"), ("righ", "right")):
        with open(tmp_path / "data" / (nm + ".pgm"), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (z[k].shape[1], z[k].shape[0]))
            f.write(z[k].tobytes())
    r = subprocess.run([exe, "0", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SIMT_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    m1 = re.search(r"Number of original features: (\d+) (\d+)", r.stdout)
    m2 = re.search(r"Number of matching features: (\d+) (\d+)", r.stdout)
    _, n1, _ = orc.extract(stereo[0], 5, 1.0, 4.5)
    _, n2, _ = orc.extract(stereo[1], 5, 1.0, 4.5)
    assert (int(m1.group(1)), int(m1.group(2))) == (n1, n2)
    g = np.load(os.path.join(GOLDEN, "refemul_golden.npz"))
    assert [int(m2.group(1)), int(m2.group(2))] == g["main_matching"].tolist()


# ------------------------------------------------- 3. oracle vs the committed vectors of the emulated reference
@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "refemul_golden.npz"))


def test_golden_dense_stages(stereo, golden):
    crop = stereo[0][300:540, 400:720].copy()
    low = orc.lowpass(crop, 1.0)
    assert sha(low) == str(golden["sha_lowpass"])
    assert sha(orc.scaledown(low)) == str(golden["sha_scaledown"])
    assert sha(orc.laplace(low, 5, 5)) == str(golden["sha_laplace"])
    odd = crop[:37, :131].copy()
    assert sha(orc.lowpass(odd, 1.3)) == str(golden["sha_lowpass_odd"])
    assert sha(orc.scaledown(odd)) == str(golden["sha_scaledown_odd"])
    assert sha(orc.laplace(odd, 5, 3)) == str(golden["sha_laplace_odd"])


@pytest.mark.parametrize("name,noct,th,up,flips", [("crop", 4, 3.5, False, 0), ("left", 5, 4.5, False, 0),
                                                   ("wide", 5, 3.0, False, 2), ("righ", 5, 4.5, False, 2),
                                                   ("crop_up", 4, 3.5, True, 1)])
def test_golden_extract(stereo, golden, name, noct, th, up, flips):
    """`wide`: 1920x1080 with the bench's parameters (left.pgm mirrored outwards); `crop_up`: scaleUp."""
    crop = stereo[0][300:540, 400:720].copy()
    img = {"crop": crop, "crop_up": crop, "left": stereo[0], "righ": stereo[1],
           "wide": np.pad(stereo[0], ((60, 60), (320, 320)), mode="reflect").astype(np.float32)}[name]
    orc.stats_reset()
    with orc.contract(1):
        pts, n, cnt = orc.extract(img, noct, 1.0, th, scale_up=up)
    assert n == int(golden[name + "_n"])
    compare_with_reference(pts, cnt, golden[name + "_records"], golden[name + "_counters"], noct, "golden/" + name, "bits",
                           flip_budget=flips, desc_stride=4 if name in ("wide", "righ", "crop_up") else 1, img=img, scale_up=up)


def test_golden_match_and_homography(golden):
    a = descriptors_to_points(synth_descriptors(1000, 7), orc.POINT_DTYPE)
    b = descriptors_to_points(synth_descriptors(1500, 8), orc.POINT_DTYPE)
    orc.match(a, 1000, b, 1500)
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert np.array_equal(a[f], golden["match_" + f]), f
    m, _, _ = synth_matches(3000, seed=5, dtype=orc.POINT_DTYPE)
    orc.srand(1)
    H, nm, _ = orc.find_homography(m, 3000, 2000, 0.85, 0.95, 5.0)
    assert nm == int(golden["homography_inliers"]) and np.array_equal(H, golden["homography_H"])


# ------------------------------------------------------------------------------------------ texture-weight rounding
_TEX_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from oracle import pyrefemul as ref
from synth import synth_frame
img = synth_frame(7, 960, 540)
pts, n, cnt = ref.extract(img, 4, 1.0, 2.5, flavour="fast")
tot = int(cnt[2 * 4 + 1])
np.savez(%(out)r, pts=pts[:tot], n=n, cnt=np.array(cnt))
"""


@needs_ref
def test_texture_weight_rounding_is_bracketed(tmp_path):
    """How the CUDA texture unit rounds the filter fraction into its 1.8 fixed-point format is not documented; oracle,
    kernels and emulator assume round-to-nearest-even (simt_emul.cpp tex_fetch).  The emulated reference is run here
    under the alternative — SIMT_TEX_ROUND=trunc, in a child process: the mode is latched at the first fetch — and both
    are compared with the oracle (= the HIP default to 1e-6): the keypoint SET does not depend on the rounding (detection
    reads no texture), and the oracle must be far closer to the nearest-rounding reference than to the truncating one;
    the distance to the truncating one — what the assumption could cost against real CUDA hardware if it truncates — is
    recorded."""
    from conftest import record
    res = {}
    for mode in ("nearest", "trunc"):
        out = str(tmp_path / ("ref_%s.npz" % mode))
        env = dict(os.environ, MISIFT_QUIET="1", SIMT_TEX_ROUND=mode)
        subprocess.run([os.sys.executable, "-c", _TEX_CHILD % {"root": ROOT, "out": out}], env=env, check=True, capture_output=True,
                       timeout=900)
        res[mode] = np.load(out)
    img = synth_frame(7, 960, 540)
    o_pts, o_n, o_cnt = orc.extract(img, 4, 1.0, 2.5)
    tot = int(o_cnt[9])
    st = {}
    for mode, z in res.items():
        # detections per octave (cnt[2o] - cnt[2o-1]) are independent of the texture unit; the second-orientation
        # duplicates — hence the running totals and numPts — may differ under the other rounding
        zc, oc = np.asarray(z["cnt"], np.int64), np.asarray(o_cnt, np.int64)
        assert np.array_equal(zc[2:9:2] - zc[1:8:2], oc[2:9:2] - oc[1:8:2]), mode
        if mode == "nearest":
            assert int(z["n"]) == o_n and np.array_equal(zc, oc)
        ia, ib, only_o, only_r = util.associate(o_pts[:tot], z["pts"])
        A, B = o_pts[:tot][ia], z["pts"][ib]
        od = util.circ_diff_deg(A["orientation"], B["orientation"])
        same = (od <= 0.036) & ~np.isnan(B["data"]).any(axis=1)          # (FastAtan2(0, 0) = NaN descriptors of the reference, B#7)
        dd = np.abs(A["data"][same].astype(np.float64) - B["data"][same]).max(axis=1)
        st[mode] = {"records": int(len(A)), "unassociated": int(len(only_o) + len(only_r)), "orientation_over_0.036deg": int((od > 0.036).sum()),
                    "orientation_max_deg": float(od.max()),
                    "desc_over_1e-4": float((dd > 1e-4).mean()), "desc_over_1e-3": float((dd > 1e-3).mean()),
                    "desc_median": float(np.median(dd)), "desc_max": float(dd.max())}
    record("texture_rounding_bracket", **{"%s_%s" % (m, k): v for m, d in st.items() for k, v in d.items()})
    assert st["nearest"]["desc_over_1e-4"] <= 0.02 and st["nearest"]["unassociated"] == 0
    assert st["trunc"]["unassociated"] <= 0.01 * st["trunc"]["records"]
    assert st["trunc"]["desc_median"] > 20 * max(st["nearest"]["desc_median"], 1e-8)      # the default IS the nearer one, by far
    assert st["trunc"]["desc_max"] < 0.2                                                  # ... and the other is no catastrophe

"""GPU parity tests: every HIP kernel (through the C-ABI) against the CPU oracle.

Bar (north_star): keypoints (x, y, scale, orientation) and descriptors within 1e-4 relative;
match indices/scores identical modulo ties.  The separable filters, DoG pyramid, extremum
decisions, refinement and the matcher's scores are held to BIT equality (explicit-FMA
arithmetic contract); orientation/descriptor go through libm/ocml and use SURVEY §7.5 tolerances.
"""
import numpy as np
import pytest

from conftest import record
from synth import descriptors_to_points, synth_descriptors, synth_frame
from util import ATOL, RTOL, associate, compare_points

pytestmark = pytest.mark.gpu


def orc():
    from oracle import pyoracle
    return pyoracle


def rand_img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.random((h, w), dtype=np.float32) * 255.0
    # add structure so blurred values are not ~constant
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return (0.5 * base + 60.0 * np.sin(xx * 0.11) * np.cos(yy * 0.07) + 64.0).astype(np.float32)


SIZES = [(37, 1281), (70, 130), (64, 1920), (135, 240), (67, 120), (200, 250), (33, 17)]


# ------------------------------------------------------------------ pyramid
@pytest.mark.parametrize("h,w", SIZES)
def test_lowpass_bitexact(ctx, h, w):
    img = rand_img(h, w, 1)
    for sigma in (1.0, 0.5, 1.7):
        ref = orc().lowpass(img, sigma)
        got = ctx.lowpass(img, sigma)
        diff = np.abs(ref - got).max()
        record("lowpass_%dx%d_s%.1f" % (w, h, sigma), maxabs=diff, bitexact=bool(np.array_equal(ref, got)))
        assert diff <= ATOL
        assert np.array_equal(ref, got), "lowpass not bit-exact, max diff %g" % diff


@pytest.mark.parametrize("h,w", SIZES)
def test_scaledown_bitexact(ctx, h, w):
    img = rand_img(h, w, 2)
    ref = orc().scaledown(img)
    got = ctx.scaledown(img)
    assert ref.shape == got.shape == (h // 2, w // 2)
    diff = np.abs(ref - got).max() if ref.size else 0.0
    record("scaledown_%dx%d" % (w, h), maxabs=diff, bitexact=bool(np.array_equal(ref, got)))
    assert diff <= ATOL
    assert np.array_equal(ref, got)


def test_scaleup_bitexact(ctx):
    img = rand_img(45, 77, 3)
    assert np.array_equal(orc().scaleup(img), ctx.scaleup(img))


def test_laplace_taps_equal():
    from cudasift_amd import capi
    for no in (1, 3, 5, 6):
        assert np.array_equal(orc().laplace_taps(no), capi.laplace_taps(no))


@pytest.mark.parametrize("h,w", SIZES[:6])
def test_laplace_bitexact(ctx, h, w):
    img = orc().lowpass(rand_img(h, w, 4), 1.0)
    for octave in (5, 3):
        ref = orc().laplace(img, 5, octave)
        got = ctx.laplace(img, 5, octave)
        diff = np.abs(ref - got).max()
        record("laplace_%dx%d_o%d" % (w, h, octave), maxabs=diff, bitexact=bool(np.array_equal(ref, got)))
        assert diff <= ATOL
        assert np.array_equal(ref, got)


# -------------------------------------------------------------- find points
def _dog_of(img, octave=5):
    base = orc().lowpass(img, 1.0)
    return base, orc().laplace(base, 5, octave)


def _cmp_detections(a, na, b, nb, name):
    """Detections carry xpos, ypos, scale, sharpness, edgeness, subsampling only."""
    A, B = a[:na].copy(), b[:nb].copy()
    A["orientation"] = 0
    B["orientation"] = 0
    ia, ib, oa, ob = associate(A, B)
    st = {"n_oracle": na, "n_hip": nb, "paired": len(ia), "paired_exact": associate.last_exact,
          "only_oracle": len(oa), "only_hip": len(ob)}
    if len(ia):
        for f in ("xpos", "ypos", "scale", "sharpness", "edgeness"):
            x, y = A[f][ia].astype(np.float64), B[f][ib].astype(np.float64)
            st[f + "_relerr"] = float((np.abs(x - y) / np.maximum(np.abs(x), 1.0)).max())
    record(name, **st)
    assert na == nb and not oa and not ob, st
    assert st["paired_exact"] == na, st
    for f in ("xpos", "ypos", "scale", "sharpness", "edgeness"):
        assert st.get(f + "_relerr", 0.0) <= RTOL, st
    return st


def test_findpoints_unfused_vs_oracle(ctx, stereo):
    img = stereo[0][200:680, 300:940]          # 640x480 real-image crop
    base, dog = _dog_of(img)
    ref, nref = orc().findpoints(dog, 2.0)
    got, ngot = ctx.findpoints(dog, 2.0, octave=5)
    assert nref > 200
    _cmp_detections(ref, nref, got, ngot, "findpoints_unfused_640x480")


@pytest.mark.parametrize("h,w", [(480, 640), (135, 240), (67, 121), (270, 483)])
def test_findpoints_fused_vs_oracle(ctx, stereo, h, w):
    img = stereo[1][100:100 + h, 200:200 + w]
    base, dog = _dog_of(img)
    ref, nref = orc().findpoints(dog, 1.5)
    got, ngot = ctx.dog_findpoints(base, 5, 5, 1.5)
    _cmp_detections(ref, nref, got, ngot, "findpoints_fused_%dx%d" % (w, h))


def test_findpoints_fused_equals_unfused_synthetic(ctx):
    img = synth_frame(3, 1920, 270)            # full-width strip decomposition
    base, dog = _dog_of(img, 4)
    a, na = ctx.findpoints(dog, 3.0, octave=4, subsampling=2.0)
    b, nb = ctx.dog_findpoints(base, 5, 4, 3.0, subsampling=2.0)
    ref, nref = orc().findpoints(dog, 3.0, subsampling=2.0)
    _cmp_detections(ref, nref, a, na, "findpoints_unfused_1920x270")
    _cmp_detections(ref, nref, b, nb, "findpoints_fused_1920x270")


# ------------------------------------------------- orientation + descriptor
@pytest.mark.parametrize("fracbits", [8, 23])
def test_orient_descr_vs_oracle(ctx, stereo, fracbits):
    img = stereo[0][100:580, 500:1140]
    base, dog = _dog_of(img)
    pts, n = orc().findpoints(dog, 2.5, max_pts=4096)
    assert n > 100
    ref = pts.copy()
    ndup = orc().orientations(base, ref, 0, n, 4096, fracbits)
    orc().descriptors(base, ref, 0, ndup, 1.0, fracbits)
    ctx.set_options(texfrac_bits=fracbits)
    try:
        got, cnt = ctx.orient_and_describe(base, pts, 0, n, octave=5, subsampling=1.0, max_pts=4096)
    finally:
        ctx.set_options(texfrac_bits=8)
    assert int(cnt[11]) == ndup, (cnt, ndup)
    compare_points(ref[:ndup], got[:ndup], "orient_descr_tex%d" % fracbits, record)


# ----------------------------------------------------------- whole pipeline
def _extract_both(ctx, img, **kw):
    ref, nref, cref = orc().extract(img, **kw)
    got, ngot, cgot = ctx.extract(img, **kw)
    return ref, nref, cref, got, ngot, cgot


@pytest.mark.parametrize("fused", [1, 0])
def test_extract_stereo_left(ctx, stereo, fused):
    ctx.set_options(fused=fused)
    try:
        ref, nref, cref, got, ngot, cgot = _extract_both(ctx, stereo[0], num_octaves=5, init_blur=1.0, thresh=4.5)
    finally:
        ctx.set_options(fused=1)
    record("extract_left_fused%d" % fused, counters_oracle=cref[:12], counters_hip=cgot[:12])
    assert np.array_equal(cref, cgot), (cref, cgot)
    assert nref == ngot
    tot = int(cref[11])                         # includes finest-octave duplicates written past numPts
    compare_points(ref[:tot], got[:tot], "extract_left_fused%d" % fused, record)


def test_extract_stereo_right_1280x960(ctx, stereo):
    ref, nref, cref, got, ngot, cgot = _extract_both(ctx, stereo[1], num_octaves=5, init_blur=1.0, thresh=4.5)
    assert np.array_equal(cref, cgot), (cref, cgot)
    compare_points(ref[:int(cref[11])], got[:int(cref[11])], "extract_right", record)


def test_extract_synthetic_1920x1080(ctx):
    img = synth_frame(0)
    ref, nref, cref, got, ngot, cgot = _extract_both(ctx, img, num_octaves=5, init_blur=1.0, thresh=3.0)
    record("extract_synth1080", counters_oracle=cref[:12], counters_hip=cgot[:12])
    assert np.array_equal(cref, cgot), (cref, cgot)
    assert 1000 < nref < 4000
    compare_points(ref[:int(cref[11])], got[:int(cref[11])], "extract_synth1080", record)


def test_extract_options_and_edge_cases(ctx, stereo):
    img = stereo[0][:333, :517]                 # odd sizes, 3 octaves, lowestScale, no scratch, tiny capacity
    for kw in (dict(num_octaves=3, thresh=3.0, lowest_scale=1.5),
               dict(num_octaves=4, thresh=2.0, max_pts=256),
               dict(num_octaves=2, thresh=1000.0)):
        ref, nref, cref = orc().extract(img, **kw)
        got, ngot, cgot = ctx.extract(img, scratch=False, **kw)
        cap = kw.get("max_pts", 32768)
        assert nref == ngot, (kw, cref, cgot)
        # counters are deterministic until the capacity is hit (afterwards WHICH points were kept, and
        # hence the duplicate count, depends on the append order — also in the reference)
        below = cref < cap
        assert np.array_equal(cref[below], cgot[below]) and (cgot[~below] >= cap).all(), (kw, cref, cgot)
        if cap > 1000 and nref:
            compare_points(ref[:nref], got[:ngot], "extract_opts_%d" % kw["num_octaves"], record)
    ctx.set_options(fix_numpts=1)
    try:
        ref, nref, cref = orc().extract(img, num_octaves=3, thresh=3.0, fix_numpts=True)
        got, ngot, cgot = ctx.extract(img, num_octaves=3, thresh=3.0)
    finally:
        ctx.set_options(fix_numpts=0)
    assert nref == ngot == int(cref[7])


def test_extract_scaleup(ctx, stereo):
    img = stereo[0][300:540, 400:720]
    ref, nref, cref = orc().extract(img, num_octaves=4, thresh=3.0, scale_up=True)
    got, ngot, cgot = ctx.extract(img, num_octaves=4, thresh=3.0, scale_up=True)
    assert np.array_equal(cref, cgot), (cref, cgot)
    compare_points(ref[:nref], got[:ngot], "extract_scaleup", record)


@pytest.mark.parametrize("fused", [1, 0])
def test_extract_scaleup_with_fix_numpts(ctx, stereo, fused):
    """options.fix_numpts puts the finest octave's second orientations INSIDE numPts: with scale_up they must be
    rescaled like every other returned record (r03 advisor finding: the fused kernels skipped them)."""
    img = stereo[0][300:540, 400:720]
    ctx.set_options(fix_numpts=1, fused=fused)
    try:
        ref, nref, cref = orc().extract(img, num_octaves=4, thresh=3.0, scale_up=True, fix_numpts=True)
        got, ngot, cgot = ctx.extract(img, num_octaves=4, thresh=3.0, scale_up=True)
    finally:
        ctx.set_options(fix_numpts=0, fused=1)
    assert np.array_equal(cref, cgot), (cref, cgot)
    assert nref == ngot == int(cref[2 * 4 + 1]) and int(cref[9]) > int(cref[8]), cref     # the case needs duplicates
    compare_points(ref[:nref], got[:ngot], "extract_scaleup_fixnum_%d" % fused, record)


def test_extract_batch_equals_single(ctx):
    imgs = np.stack([synth_frame(10 + f, 640, 360) for f in range(5)])
    pts, n = ctx.extract_batch(imgs, num_octaves=4, thresh=3.0, max_pts=8192)
    for f in range(5):
        ref, nref, cref = orc().extract(imgs[f], num_octaves=4, thresh=3.0, max_pts=8192)
        assert nref == n[f], (f, nref, n)
        compare_points(ref[:nref], pts[f][:n[f]], "extract_batch_f%d" % f, record)


@pytest.mark.parametrize("nframes,noct,w,h,u8", [(2, 5, 640, 360, False), (4, 3, 322, 250, False), (3, 6, 770, 516, True),
                                                  (2, 7, 1283, 1030, False), (1, 2, 193, 131, False)])
def test_small_batches_take_the_single_call_kernels(ctx, nframes, noct, w, h, u8):
    """Batches of 1-4 frames run the latency-shaped kernels of the single-call path (r04: tiled prefilter, chained
    ScaleDowns inside the scan launch or — more than three coarse levels — as launches of their own, no binning, the last
    kernel's counter export): several frames per launch, ragged widths, 8-bit sources, 2 ... 7 octaves vs the oracle."""
    imgs = np.stack([synth_frame(300 + 7 * f + noct, w, h) for f in range(nframes)])
    if u8:
        imgs = np.clip(np.rint(imgs), 0, 255).astype(np.uint8)
    pts, n = ctx.extract_batch_ex(imgs, num_octaves=noct, thresh=2.5, max_pts=8192)
    for f in range(nframes):
        ref, nref, cref = orc().extract(imgs[f].astype(np.float32), num_octaves=noct, thresh=2.5, max_pts=8192)
        assert nref == n[f] and nref > 50, (f, nref, n)
        compare_points(ref[:nref], pts[f][:n[f]], "small_batch_%d_%d_f%d" % (nframes, noct, f), record)


def test_extract_deterministic_set(ctx, stereo):
    a, na, ca = ctx.extract(stereo[1][:480, :640], thresh=3.0)
    b, nb, cb = ctx.extract(stereo[1][:480, :640], thresh=3.0)
    assert na == nb and np.array_equal(ca, cb)
    ia, ib, oa, ob = associate(a[:na], b[:nb])
    assert not oa and not ob and associate.last_exact == na


# ------------------------------------------------------------------ matcher
def _match_case(ctx, n1, n2, seed, full, exact, l2=False):
    o = orc()
    p1 = descriptors_to_points(synth_descriptors(n1, seed, l2), o.POINT_DTYPE)
    p2 = descriptors_to_points(synth_descriptors(n2, seed + 1, l2), o.POINT_DTYPE)
    ref = p1.copy()
    o.match(ref, n1, p2, n2, full=full, exact=exact)
    ctx.set_options(match_full=int(full), match_exact_top2=int(exact))
    try:
        got = ctx.match(p1, n1, p2, n2)
    finally:
        ctx.set_options(match_full=0, match_exact_top2=0)
    name = "match_%dx%d_f%d_e%d" % (n1, n2, full, exact)
    st = {"idx_equal": float((ref["match"] == got["match"]).mean()),
          "score_bitexact": bool(np.array_equal(ref["score"], got["score"])),
          "amb_bitexact": bool(np.array_equal(ref["ambiguity"], got["ambiguity"])),
          "score_maxabs": float(np.abs(ref["score"] - got["score"]).max()),
          "amb_maxabs": float(np.abs(ref["ambiguity"] - got["ambiguity"]).max())}
    record(name, **st)
    assert np.array_equal(ref["match"], got["match"]), st
    assert np.array_equal(ref["score"], got["score"]), st
    assert np.array_equal(ref["ambiguity"], got["ambiguity"]), st
    assert np.array_equal(ref["match_xpos"], got["match_xpos"]) and np.array_equal(ref["match_ypos"], got["match_ypos"])
    return ref, got


@pytest.mark.parametrize("n1,n2", [(1000, 1000), (2000, 2085), (37, 31), (129, 64), (1, 33), (4100, 3000)])
def test_match_reference_mode(ctx, n1, n2):
    _match_case(ctx, n1, n2, 7, full=False, exact=False)


@pytest.mark.parametrize("full,exact", [(True, False), (False, True), (True, True)])
def test_match_modes(ctx, full, exact):
    _match_case(ctx, 777, 1003, 11, full, exact, l2=True)


def test_match_empty_and_ties(ctx):
    o = orc()
    p1 = descriptors_to_points(synth_descriptors(64, 3), o.POINT_DTYPE)
    # n2 < 32 in reference mode: no column takes part -> match -1, score 0, xpos/ypos 0
    got = ctx.match(p1, 64, p1[:20].copy(), 20)
    assert (got["match"] == -1).all() and (got["score"] == 0).all() and (got["match_xpos"] == 0).all()
    # exact duplicates in set 2: the earliest index must win, ambiguity ~ 1
    p2 = np.concatenate([p1, p1])
    ref = p1.copy()
    o.match(ref, 64, p2, 128)
    got = ctx.match(p1, 64, p2, 128)
    assert np.array_equal(ref["match"], got["match"]) and (got["match"] == np.arange(64)).all()
    assert np.array_equal(ref["ambiguity"], got["ambiguity"])
    # all-negative correlations never win (score initialised to 0, matching.cu:317)
    neg = p1.copy()
    neg["data"] *= -1.0
    got = ctx.match(neg, 64, p1, 64)
    assert (got["match"] == -1).all()


def test_match_rows_split(ctx):
    o = orc()
    n1, n2 = 1500, 1536
    p1 = descriptors_to_points(synth_descriptors(n1, 21), o.POINT_DTYPE)
    p2 = descriptors_to_points(synth_descriptors(n2, 22), o.POINT_DTYPE)
    full = ctx.match(p1, n1, p2, n2)
    part = p1.copy()
    for r0, rc in ((0, 700), (700, 800)):
        tmp = ctx.match(part, n1, p2, n2, row_begin=r0, row_count=rc)
        part[r0:r0 + rc] = tmp[r0:r0 + rc]
    for f in ("score", "ambiguity", "match"):
        assert np.array_equal(full[f], part[f])


@pytest.mark.parametrize("n2,t0,t1", [(1536, 0, 24), (1536, 5, 9), (2085, 0, 3), (2085, 31, 33), (2085, 12, 12), (100, 0, 1)])
def test_match_column_split_is_bit_identical(ctx, n2, t0, t1):
    """The two-launch column cut of the sharded matcher (own super-tiles first, the rest around them, one merge over the
    chunks of both) gives the single sweep's bits, ties included (duplicated columns on both sides of the cut)."""
    o = orc()
    n1 = 333
    p1 = descriptors_to_points(synth_descriptors(n1, 51), o.POINT_DTYPE)
    p2 = descriptors_to_points(synth_descriptors(n2, 52), o.POINT_DTYPE)
    if n2 > 700:
        p2["data"][64 * 6 + 3] = p2["data"][17]             # ties straddling the cut: the smaller column must win
        p2["data"][64 * 2 + 1] = p2["data"][64 * 8 + 7]
    ref = p1.copy()
    o.match(ref, n1, p2, n2)
    one = ctx.match(p1, n1, p2, n2)
    two = ctx.match_split(p1, n1, p2, n2, t0, t1)
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert np.array_equal(one[f], two[f]), f
    for f in ("score", "ambiguity", "match"):
        assert np.array_equal(ref[f], two[f]), f


def test_match_real_descriptors(ctx, stereo):
    a, na, _ = ctx.extract(stereo[0], thresh=4.5)
    b, nb, _ = ctx.extract(stereo[1], thresh=4.5)
    ref = a.copy()
    orc().match(ref, na, b, nb)
    got = ctx.match(a, na, b, nb)
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert np.array_equal(ref[f][:na], got[f][:na]), f
    m = got[:na]
    good = m["ambiguity"] < 0.95
    dy = m["match_ypos"][good] - m["ypos"][good]
    dx = m["match_xpos"][good] - m["xpos"][good]
    # the pair is a ~(-533, +18) px shift: most confident matches must agree with it
    frac = float(((np.abs(dx + 533) < 40) & (np.abs(dy - 18) < 40)).mean())
    record("match_real", n1=na, n2=nb, confident=int(good.sum()), geometric_consistency=frac)
    assert good.sum() > 200 and frac > 0.7


def test_match_large_property(ctx):
    """Full-size style property (no oracle): matching a set against a shuffled copy of itself
    must return the permutation with score == self-correlation, at 20000 x 20000."""
    o = orc()
    n = 20000
    d = synth_descriptors(n, 99, l2=True)
    perm = np.random.default_rng(5).permutation(n)
    p1 = descriptors_to_points(d, o.POINT_DTYPE)
    p2 = descriptors_to_points(d[perm], o.POINT_DTYPE)
    ctx.set_options(match_full=1, match_exact_top2=1)
    try:
        got = ctx.match(p1, n, p2, n)
    finally:
        ctx.set_options(match_full=0, match_exact_top2=0)
    inv = np.empty(n, int)
    inv[perm] = np.arange(n)
    assert np.array_equal(got["match"], inv)
    self_corr = np.array([o.lib().orc_dot128(d[i].ctypes.data, d[i].ctypes.data) for i in range(0, n, 997)], np.float32)
    # orc_dot128 returns c_float; compare a sample of rows bit-exactly
    assert np.array_equal(got["score"][::997], self_corr)
    assert (got["ambiguity"] < 1.0).all()


def test_homography_from_matches(ctx, stereo):
    from cudasift_amd import capi
    a, na, _ = ctx.extract(stereo[0], thresh=4.5)
    b, nb, _ = ctx.extract(stereo[1], thresh=4.5)
    m = ctx.match(a, na, b, nb)
    d = ctx.upload(m)
    H, nmatch = ctx.find_homography(d.ptr, na, num_loops=2000, min_score=0.0, max_ambiguity=0.95, thresh=5.0)
    record("homography", H=H.reshape(-1), inliers=nmatch)
    assert nmatch > 100 and H[2, 2] == 1.0
    # the winning hypothesis must reproject the confident matches (ambiguity < 0.95) within the 5 px threshold
    g = m[:na][(m[:na]["ambiguity"] < 0.95) & (m[:na]["score"] > 0)]
    den = H[2, 0] * g["xpos"] + H[2, 1] * g["ypos"] + 1.0
    ex = (H[0, 0] * g["xpos"] + H[0, 1] * g["ypos"] + H[0, 2]) / den - g["match_xpos"]
    ey = (H[1, 0] * g["xpos"] + H[1, 1] * g["ypos"] + H[1, 2]) / den - g["match_ypos"]
    frac = float((ex * ex + ey * ey < 25.0).mean())
    record("homography", reprojected_within_5px=frac)
    assert frac > 0.5


@pytest.mark.parametrize("n,loops,seed", [(1500, 1000, 1), (777, 10000, 2), (40, 50, 3), (5000, 2000, 4)])
def test_find_homography_bit_exact_vs_oracle(ctx, n, loops, seed):
    """GPU FindHomography (gather / solve / count / pick kernels) against orc_find_homography with the
    same libc rand() state: H (8 floats) and the inlier count must be bit-identical."""
    from synth import synth_matches
    pts, _, inl = synth_matches(n, inlier_frac=0.55, seed=seed)
    orc().srand(seed)
    Ho, co, _ = orc().find_homography(pts, n, num_loops=loops, min_score=0.85, max_ambiguity=0.95, thresh=5.0)
    d = ctx.upload(pts)
    orc().srand(seed)
    Hg, cg = ctx.find_homography(d.ptr, n, num_loops=loops, min_score=0.85, max_ambiguity=0.95, thresh=5.0)
    record("homography_parity", n=n, loops=loops, inliers_oracle=co, inliers_hip=cg,
           H_equal=bool(np.array_equal(Ho, Hg)))
    assert cg == co
    assert np.array_equal(Ho.view(np.uint32), Hg.view(np.uint32))
    assert co >= 0.8 * inl.sum()


def test_find_homography_degenerate(ctx):
    from synth import synth_matches
    pts, _, _ = synth_matches(64, seed=5)
    d = ctx.upload(pts)
    H, c = ctx.find_homography(d.ptr, 7)
    assert np.array_equal(H, np.eye(3, dtype=np.float32)) and c == 0
    pts["score"] = 0.1
    d = ctx.upload(pts)
    H, c = ctx.find_homography(d.ptr, 64)
    assert np.array_equal(H, np.eye(3, dtype=np.float32)) and c == 0


def test_find_homography_stereo_vs_oracle(ctx, stereo):
    """End of the reference demo (mainSift.cpp:72-77): extract, match, FindHomography — vs the oracle chain."""
    a, na, _ = ctx.extract(stereo[0], thresh=4.5)
    b, nb, _ = ctx.extract(stereo[1], thresh=4.5)
    m = ctx.match(a, na, b, nb)
    d = ctx.upload(m)
    orc().srand(7)
    Hg, cg = ctx.find_homography(d.ptr, na, num_loops=10000, min_score=0.0, max_ambiguity=0.80, thresh=5.0)
    orc().srand(7)
    Ho, co, _ = orc().find_homography(m, na, num_loops=10000, min_score=0.0, max_ambiguity=0.80, thresh=5.0)
    assert cg == co and np.array_equal(Ho.view(np.uint32), Hg.view(np.uint32))


# ------------------------------------------------------------------ 8-bit frames + host-fed pipeline
def _canon(recs):
    """Records of one frame in a canonical order (appends within an octave are atomic, so the order varies)."""
    k = [recs[f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")]
    return recs[np.lexsort(k)].tobytes()


def _u8_frames(n, h, w, seed0=100):
    return np.stack([np.clip(np.rint(synth_frame(seed0 + i, width=w, height=h)), 0, 255).astype(np.uint8) for i in range(n)])


@pytest.mark.parametrize("h,w,noct", [(272, 480, 5), (131, 203, 3), (160, 330, 4)])
def test_extract_u8_bit_identical_to_f32_upload(ctx, h, w, noct):
    """misift_extract_batch_u8 == misift_extract_batch on the same pixel values (the in-register conversion is
    exact), for the aligned fast path and for widths that are not a multiple of 4."""
    f8 = _u8_frames(2, h, w)
    pf, nf = ctx.extract_batch(f8.astype(np.float32), num_octaves=noct, thresh=2.0, max_pts=4096)
    pu, nu = ctx.extract_batch_u8(f8, num_octaves=noct, thresh=2.0, max_pts=4096)
    assert np.array_equal(nf, nu) and nf.min() > 20
    for b in range(2):
        assert _canon(pf[b, :nf[b]]) == _canon(pu[b, :nu[b]])
    # ... and equal to the oracle on that image (same bar as the fp32 path)
    o, no_, _ = orc().extract(f8[0].astype(np.float32), num_octaves=noct, thresh=2.0, max_pts=4096)
    rep = compare_points(o[:no_], pu[0, :nu[0]], "extract_u8_%dx%d" % (w, h), record)
    assert rep["paired_exact"] == rep["paired"]


@pytest.mark.parametrize("src_u8", [True, False])
def test_pipe_matches_batch_extraction(ctx, src_u8):
    """The 3-stream pipeline returns, batch after batch and frame after frame, exactly the records the
    synchronous batch call returns (ragged last batch, more batches than slots)."""
    from cudasift_amd import capi
    h, w, B, nb = 272, 480, 3, 5
    frames = _u8_frames(B * nb - 1, h, w, seed0=300)           # last batch has B-1 frames
    src = frames if src_u8 else frames.astype(np.float32)
    ref_fn = ctx.extract_batch_u8 if src_u8 else ctx.extract_batch
    pin = capi.PinnedArray(src.shape, src.dtype)
    pin.array[...] = src
    out = capi.PinnedArray((B * 4096,), capi.POINT_DTYPE)
    pipe = capi.Pipe(ctx, w, h, B, src_u8=src_u8, thresh=2.0, max_pts=4096, depth=2)
    esz = src.dtype.itemsize * h * w
    got = []

    def collect():
        counts, nrec = pipe.collect(out.ptr, B * 4096)
        assert nrec == counts.sum()
        got.append((counts, out.array[:nrec].copy()))

    for k in range(nb):
        if pipe.pending() == 2:
            collect()
        n = min(B, len(src) - k * B)
        pipe.submit(pin.ptr + k * B * esz, n)
    while pipe.pending():
        collect()
    pipe.close()
    assert len(got) == nb
    for k, (counts, recs) in enumerate(got):
        n = min(B, len(src) - k * B)
        rp, rn = ref_fn(src[k * B:k * B + n], thresh=2.0, max_pts=4096)
        assert np.array_equal(counts, rn) and counts.min() > 20
        off = 0
        for f in range(n):
            assert _canon(recs[off:off + rn[f]]) == _canon(rp[f, :rn[f]])
            off += rn[f]
    record("pipe_parity_u8" if src_u8 else "pipe_parity_f32", batches=nb, frames=len(src), identical=True)


def test_pipe_error_paths(ctx):
    from cudasift_amd import capi
    h, w = 128, 128
    pin = capi.PinnedArray((1, h, w), np.uint8)
    pin.array[...] = _u8_frames(1, h, w)
    pipe = capi.Pipe(ctx, w, h, 1, src_u8=True, num_octaves=3, thresh=1.0, max_pts=2048, depth=1)
    with pytest.raises(capi.MisiftError):
        pipe.collect()                                           # nothing in flight
    pipe.submit(pin.ptr, 1)
    with pytest.raises(capi.MisiftError):
        pipe.submit(pin.ptr, 1)                                  # all slots busy
    out = capi.PinnedArray((4,), capi.POINT_DTYPE)
    with pytest.raises(capi.MisiftError):
        pipe.collect(out.ptr, 4)                                 # room for 4 records only
    assert pipe.pending() == 0
    pipe.submit(pin.ptr, 1)
    counts, nrec = pipe.collect()                                # counts only
    assert nrec == counts.sum() and counts[0] > 4
    pipe.close()


@pytest.mark.parametrize("h,w,noct", [(9, 12, 2), (40, 100, 5), (15, 15, 1)])
def test_pipe_takes_tiny_frames(ctx, h, w, noct):
    """misift_pipe accepts every size the other entry points accept (r06; r05 refused frames under 16 x 16 and pyramids
    whose coarsest level is under 8 px): same records as the batch call."""
    from cudasift_amd import capi
    B = 3
    frames = np.random.default_rng(w).integers(0, 256, (B, h, w)).astype(np.uint8)
    pin = capi.PinnedArray(frames.shape, np.uint8)
    pin.array[...] = frames
    out = capi.PinnedArray((B * 2048,), capi.POINT_DTYPE)
    pipe = capi.Pipe(ctx, w, h, B, src_u8=True, num_octaves=noct, thresh=0.5, max_pts=2048, depth=2)
    pipe.submit(pin.ptr, B)
    counts, nrec = pipe.collect(out.ptr, B * 2048)
    pipe.close()
    rp, rn = ctx.extract_batch_ex(frames, num_octaves=noct, thresh=0.5, max_pts=2048)
    assert np.array_equal(counts, rn)
    off = 0
    for f in range(B):
        assert _canon(out.array[off:off + rn[f]]) == _canon(rp[f, :rn[f]])
        off += rn[f]


@pytest.mark.parametrize("h,w", [(1080, 1920), (960, 1280), (135, 240), (37, 260), (8, 4), (270, 480), (539, 484),
                                 (67, 1024), (135, 241), (64, 1281), (270, 483), (33, 17), (100, 250), (77, 999)])
def test_fused_lowpass_scaledown_bit_exact(ctx, h, w):
    """lowpass_down_kernel == LowPass then ScaleDown (oracle), bit for bit: strip seams (w > 240), segment
    seams, odd and even heights (bottom clamp), tiny images; since r03 also widths that are not a multiple of 4 (the
    ragged-quad instantiation: remainders 1, 2 and 3, one strip and several)."""
    rng = np.random.default_rng(h * 10007 + w)
    img = (rng.random((h, w), dtype=np.float32) * 255.0).astype(np.float32)
    lp, dn = ctx.lowpass_scaledown(img, 1.0)
    ref_lp = orc().lowpass(img, 1.0)
    ref_dn = orc().scaledown(ref_lp)
    record("fused_lowpass_scaledown_%dx%d" % (w, h), lowpass_equal=bool(np.array_equal(lp, ref_lp)),
           down_equal=bool(np.array_equal(dn, ref_dn)))
    assert np.array_equal(lp, ref_lp)
    assert dn.shape == ref_dn.shape and np.array_equal(dn, ref_dn)


def test_extract_batch_packed_async_equals_batch(ctx):
    """misift_extract_batch_packed_async (what the multi-GPU gather ships): counts, offsets and the packed
    records equal the synchronous batch call's, frame after frame."""
    import ctypes as C
    from cudasift_amd import capi
    h, w, B, mp = 272, 480, 3, 4096
    frames = np.stack([synth_frame(500 + i, width=w, height=h) for i in range(B)]).astype(np.float32)
    rp, rn = ctx.extract_batch(frames, thresh=2.0, max_pts=mp)
    d = ctx.upload(frames)
    sc = capi.DevBuf(4 * capi.scratch_floats(w, h, 5, False) * B)
    pts = ctx.zeros(576 * mp * B)
    packed = ctx.zeros(576 * mp * B)
    cnt = ctx.zeros(4 * (2 * B + 1))
    capi.check(capi.lib().misift_extract_batch_packed_async(ctx.h, d.ptr, B, h * w, w, h, w, 5, 1.0, 2.0, 0.0, sc.ptr,
                                                            pts.ptr, mp, cnt.ptr, cnt.ptr + 4 * B, packed.ptr),
               "misift_extract_batch_packed_async")
    ctx.sync()
    ci = ctx.download(cnt, (2 * B + 1,), np.int32)
    counts, offs = ci[:B], ci[B:]
    assert np.array_equal(counts, rn) and counts.min() > 20
    assert offs[0] == 0 and np.array_equal(np.diff(offs), counts)
    recs = ctx.download(packed, (int(offs[B]),), capi.POINT_DTYPE)
    for f in range(B):
        assert _canon(recs[offs[f]:offs[f + 1]]) == _canon(rp[f, :rn[f]])
    unp = ctx.download(pts, (B, mp), capi.POINT_DTYPE)                      # d_pts was given: filled as well
    for f in range(B):
        assert _canon(unp[f, :rn[f]]) == _canon(rp[f, :rn[f]])
    # packed-only (d_pts = NULL), into a dirty buffer: every byte of every record must be written
    packed2 = ctx.upload(np.full(576 * mp * B, 0xA5, np.uint8))
    capi.check(capi.lib().misift_extract_batch_packed_async(ctx.h, d.ptr, B, h * w, w, h, w, 5, 1.0, 2.0, 0.0, sc.ptr,
                                                            None, mp, cnt.ptr, cnt.ptr + 4 * B, packed2.ptr),
               "misift_extract_batch_packed_async")
    ctx.sync()
    recs2 = ctx.download(packed2, (int(offs[B]),), capi.POINT_DTYPE)
    for f in range(B):
        assert _canon(recs2[offs[f]:offs[f + 1]]) == _canon(rp[f, :rn[f]])


def _noise_u8(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w), dtype=np.uint8)


def test_candidate_overflow_falls_back_to_dense_kernels(ctx):
    """White noise at a tiny threshold floods the fused scan's pre-candidate list (cap = max(16384, w*h/4) per
    octave); extract must notice and redo the frame with the dense kernels — same keypoints as the oracle,
    nothing dropped silently (the reference caps silently at 32 candidates per tile, cudaSiftD.cu:1371)."""
    img = _noise_u8(256, 256, 11).astype(np.float32)
    ref, nref, cref = orc().extract(img, num_octaves=3, thresh=0.05, max_pts=32768)
    got, ngot, cgot = ctx.extract(img, num_octaves=3, thresh=0.05, max_pts=32768)
    record("overflow_fallback", n_oracle=int(nref), n_hip=int(ngot))
    assert nref > 2000 and ngot == nref and np.array_equal(cref, cgot)
    compare_points(ref[:nref], got[:ngot], "overflow_fallback_points", record)
    # the premise: the fused path alone does overflow on this frame (the async entry reports it as count -1)
    from cudasift_amd import capi
    d = ctx.upload(img)
    sc = capi.DevBuf(4 * capi.scratch_floats(256, 256, 3, False))
    pts = ctx.zeros(576 * 32768)
    cnt = ctx.zeros(4)
    was_fused = ctx.get_options().fused
    ctx.set_options(fused=1)                  # (a MISIFT_FUSED=0 run of the suite starts on the dense kernels)
    try:
        capi.check(capi.lib().misift_extract_batch_async(ctx.h, d.ptr, 1, 256 * 256, 256, 256, 256, 3, 1.0, 0.05, 0.0,
                                                         sc.ptr, pts.ptr, 32768, cnt.ptr), "misift_extract_batch_async")
        ctx.sync()
    finally:
        ctx.set_options(fused=was_fused)
    assert ctx.download(cnt, (1,), np.int32)[0] == -1


def test_overflow_rerun_touches_only_the_overflowed_frames(ctx):
    """A batch in which frames 1, 2 and 5 of 6 flood their candidate lists: the exact dense re-run covers those frames
    only (two runs of consecutive frames -> 2 detect launches per octave, not one over the whole batch), every frame
    equals the oracle, and the 17 counters of every frame — re-run or not — are the oracle's."""
    from cudasift_amd import capi
    h, w, noct, th = 256, 256, 3, 0.05
    normal = [np.clip(np.rint(synth_frame(8800 + i, w, h)), 0, 255).astype(np.float32) for i in range(3)]
    noise = [_noise_u8(h, w, 21 + i).astype(np.float32) for i in range(3)]
    frames = np.stack([normal[0], noise[0], noise[1], normal[1], normal[2], noise[2]])
    c = capi.Context(0)
    try:
        c.set_options(fused=1)
        c.profile_enable(True)
        pts, n = c.extract_batch(frames, num_octaves=noct, thresh=th, max_pts=32768)
        prof = c.profile_read()
        counters = np.stack([c.get_counters(f) for f in range(len(frames))])
    finally:
        c.close()
    assert prof["detect"]["calls"] == 2 * noct, prof["detect"]             # frames {1,2} and {5}: two sub-calls
    assert prof["dog_scan"]["calls"] >= 1
    for f in range(len(frames)):
        ref, nref, cref = orc().extract(frames[f], num_octaves=noct, thresh=th, max_pts=32768)
        assert n[f] == nref and np.array_equal(counters[f], cref), f
        compare_points(ref[:nref], pts[f, :nref], "overflow_selective_f%d" % f, record)


def test_pipe_overflow_falls_back(ctx):
    """The same inside the pipeline: the overflowed batch is redone when it is collected; the batches around
    it are untouched."""
    from cudasift_amd import capi
    h, w, B = 256, 256, 2
    frames = np.stack([_u8_frames(1, h, w, seed0=700)[0], _noise_u8(h, w, 11),      # batch 0: normal, noise
                       _u8_frames(1, h, w, seed0=701)[0], _u8_frames(1, h, w, seed0=702)[0]])   # batch 1: normal
    pin = capi.PinnedArray(frames.shape, np.uint8)
    pin.array[...] = frames
    out = capi.PinnedArray((B * 32768,), capi.POINT_DTYPE)
    pipe = capi.Pipe(ctx, w, h, B, src_u8=True, num_octaves=3, thresh=0.05, max_pts=32768, depth=2)
    pipe.submit(pin.ptr, B)
    pipe.submit(pin.ptr + B * h * w, B)
    got = []
    for _ in range(2):
        counts, nrec = pipe.collect(out.ptr, B * 32768)
        got.append((counts, out.array[:nrec].copy()))
    pipe.close()
    for k, (counts, recs) in enumerate(got):
        rp, rn = ctx.extract_batch_u8(frames[k * B:(k + 1) * B], num_octaves=3, thresh=0.05, max_pts=32768)
        assert np.array_equal(counts, rn) and counts.min() >= 0
        off = 0
        for f in range(B):
            assert _canon(recs[off:off + rn[f]]) == _canon(rp[f, :rn[f]])
            off += rn[f]
    assert got[0][0][1] > 2000


def test_repeated_call_replays_graph_with_identical_results(ctx, stereo):
    """A synchronous call repeated with the same buffers (what mainSift.cpp:64-69 does 1000 times) is captured
    into a hipGraph on its 2nd occurrence and replayed from the 3rd: every repetition must return the same
    keypoint set as the first, ordinary run — also after an interleaved different call and a re-allocation."""
    import ctypes as C
    from cudasift_amd import capi
    img = stereo[0]
    h, w = img.shape
    src, p = ctx.upload_image(img)
    sc = capi.DevBuf(4 * capi.scratch_floats(w, h, 5, False))
    pts = ctx.zeros(576 * 8192)
    n = C.c_int(0)

    def run():
        capi.check(capi.lib().misift_extract(ctx.h, src.ptr, w, h, p, 5, 1.0, 3.5, 0.0, 0, sc.ptr, pts.ptr, 8192,
                                             C.byref(n)), "misift_extract")
        return _canon(ctx.download(pts, (n.value,), capi.POINT_DTYPE))
    first = run()                                           # ordinary launches (replay is off by default)
    assert n.value > 500
    capi.check(capi.lib().misift_ctx_set_graph_replay(ctx.h, 1), "misift_ctx_set_graph_replay")
    for _ in range(5):
        assert run() == first
    ctx.extract(stereo[1], thresh=3.5)                      # a different call in between (new buffers)
    assert run() == first
    big = np.stack([synth_frame(900 + i, width=640, height=480) for i in range(3)]).astype(np.float32)
    ctx.extract_batch(big, thresh=3.0, max_pts=2048)         # forces the per-frame buffers to grow (re-allocation)
    for _ in range(3):
        assert run() == first
    capi.check(capi.lib().misift_ctx_set_graph_replay(ctx.h, 0), "misift_ctx_set_graph_replay")
    assert run() == first


def test_extract_large_frame_4096x3072(ctx):
    """A 12.6 Mpx frame (17 strips, many segments per level, candidate words with 12-bit coordinates): same
    keypoint set and counters as the oracle."""
    img = synth_frame(4242, width=4096, height=3072)
    ref, nref, cref = orc().extract(img, num_octaves=5, thresh=3.0, max_pts=32768)
    got, ngot, cgot = ctx.extract(img, num_octaves=5, thresh=3.0, max_pts=32768)
    record("extract_4096x3072", n_oracle=int(nref), n_hip=int(ngot))
    assert nref > 5000 and ngot == nref and np.array_equal(cref, cgot)
    tot = int(cref[11])
    compare_points(ref[:tot], got[:tot], "extract_4096x3072_points", record)


# ------------------------------------------------------------------ the path bench.py times
def _packed_async_1080p(c, d_frames, B, scratch, cnt, packed, mp):
    """One call of the exact entry point bench.py times: packed-only output (d_pts = NULL), 1920x1080, B frames."""
    from cudasift_amd import capi
    capi.check(capi.lib().misift_extract_batch_packed_async(c.h, d_frames.ptr, B, 1080 * 1920, 1920, 1080, 1920, 5, 1.0, 3.0,
                                                            0.0, scratch.ptr, None, mp, cnt.ptr, cnt.ptr + 4 * B,
                                                            packed.ptr), "misift_extract_batch_packed_async")
    c.sync()
    ci = c.download(cnt, (2 * B + 1,), np.int32)
    counts, offs = ci[:B].copy(), ci[B:].copy()
    recs = c.download(packed, (int(offs[B]),), capi.POINT_DTYPE)
    counters = np.stack([c.get_counters(f) for f in range(B)])
    return counts, offs, recs, counters


def test_timed_path_16x1080p_packed_async_vs_oracle(ctx):
    """The path bench.py times — misift_extract_batch_packed_async with d_pts = NULL on a batch of 1920x1080 frames
    large enough (>= 8) to take the split-tail branch (second high-priority stream, fork/join events, two
    dog_scan launches) — against the oracle, frame by frame: 17 counters + every record.  Then the same batch
    with the split disabled (MISIFT_SPLIT_TAIL=0): byte-identical records after canonical sort."""
    import os
    from cudasift_amd import capi
    B, mp = 16, 8192
    frames = np.stack([synth_frame(7000 + f) for f in range(B)])
    d = ctx.upload(frames)
    scratch = capi.DevBuf(4 * capi.scratch_floats(1920, 1080, 5, False) * B)
    cnt = ctx.zeros(4 * (2 * B + 1))
    packed = ctx.upload(np.full(576 * mp * B, 0xA5, np.uint8))          # dirty: every byte of every record must be written
    was_fused = ctx.get_options().fused
    ctx.set_options(fused=1)              # the timed path IS the merged-octave path (a MISIFT_FUSED=0 run of the suite included)
    try:
        counts, offs, recs, counters = _packed_async_1080p(ctx, d, B, scratch, cnt, packed, mp)
    finally:
        ctx.set_options(fused=was_fused)
    prof_ctx = capi.Context(0)
    try:                                                                  # the premise: this batch does take two scan launches
        prof_ctx.set_options(fused=1)
        prof_ctx.profile_enable(True)
        _packed_async_1080p(prof_ctx, d, B, scratch, cnt, ctx.zeros(576 * mp * B), mp)
        assert prof_ctx.profile_read()["dog_scan"]["calls"] == 2
    finally:
        prof_ctx.close()
    ref, nref, cref = orc().extract_batch(frames, num_octaves=5, init_blur=1.0, thresh=3.0, max_pts=mp)
    assert offs[0] == 0 and np.array_equal(np.diff(offs), counts)
    assert np.array_equal(counts, nref), (counts, nref)
    assert np.array_equal(counters, cref), "17-counter protocol differs from the oracle"
    assert 1500 < counts.min() and counts.max() < 3000
    for f in range(B):
        compare_points(ref[f, :nref[f]], recs[offs[f]:offs[f + 1]], "timed_path_f%d" % f, record)
    record("timed_path", frames=B, keypoints=int(counts.sum()), split_tail=True)
    c2 = capi.Context(0)
    c2.set_knob("split_tail", 0)
    try:
        c2.set_options(fused=1)
        packed2 = c2.upload(np.full(576 * mp * B, 0x5A, np.uint8))
        c2.profile_enable(True)
        counts2, offs2, recs2, counters2 = _packed_async_1080p(c2, d, B, scratch, cnt, packed2, mp)
        assert c2.profile_read()["dog_scan"]["calls"] == 1               # single-stream structure
    finally:
        c2.close()
    assert np.array_equal(counts, counts2) and np.array_equal(offs, offs2) and np.array_equal(counters, counters2)
    for f in range(B):
        assert _canon(recs[offs[f]:offs[f + 1]]) == _canon(recs2[offs[f]:offs[f + 1]]), f


def test_balanced_keypoint_launches_on_a_skewed_batch(ctx):
    """MISIFT_BALANCE (on by default since r05): the workgroups of orient_all / descr_all are dealt out in proportion to the
    frames' keypoint counts (a block -> (frame, sub-block, sub-blocks) table, built behind refine_all by one extra workgroup
    of the bin_detections launch) instead of the same number per frame.  A batch of eight frames from a busy one down to an
    empty one must come out record for record as from the plain launch (a context created under MISIFT_BALANCE=0), and equal
    to the oracle."""
    import os
    from cudasift_amd import capi
    from synth import SYNTH_AMP
    w, h, mp = 640, 360, 8192
    amps = [3.0, 1.0, 1.5, 0.8, 0.6, 0.0, 2.0, 0.5]             # oracle: 7224, 207, 1654, 28, 3, 0, 4405, 0 keypoints
    frames = np.stack([synth_frame(9100 + i, w, h, amp=SYNTH_AMP * a) for i, a in enumerate(amps)])

    def fresh(balance):
        c = capi.Context(0)
        c.set_knob("balance", balance)
        return c
    c1, c2 = fresh(0), fresh(1)
    try:
        c1.set_options(fused=1)
        c2.set_options(fused=1)
        pts_u, n_u = c1.extract_batch(frames, num_octaves=5, thresh=3.0, max_pts=mp)
        assert c1.last_call_balanced() == 0
        c2.profile_enable(True)
        pts_b, n_b = c2.extract_batch(frames, num_octaves=5, thresh=3.0, max_pts=mp)
        assert c2.last_call_balanced() == 1                              # the premise: the table was built and used ...
        assert "frame_shares" not in c2.profile_read()                   # ... inside the bin_detections launch, no kernel of its own
        # ... and batches of a frame or two keep the plain launch (single-call path)
        c2.extract_batch(frames[:2], num_octaves=5, thresh=3.0, max_pts=mp)
        assert c2.last_call_balanced() == 0
    finally:
        c1.close()
        c2.close()
    assert np.array_equal(n_u, n_b), (n_u, n_b)
    assert n_u[5] == 0 and n_u[0] > 4 * n_u[1] > 0, n_u                   # skewed indeed, one frame empty
    for f in range(len(amps)):
        assert _canon(pts_u[f, :n_u[f]]) == _canon(pts_b[f, :n_b[f]]), f
    ref, nref, _ = orc().extract_batch(frames, num_octaves=5, init_blur=1.0, thresh=3.0, max_pts=mp)
    assert np.array_equal(n_b, nref), (n_b, nref)
    for f in range(len(amps)):
        compare_points(ref[f, :nref[f]], pts_b[f, :n_b[f]], "balanced_f%d" % f, record)
    record("balanced_skewed_batch", frames=len(amps), keypoints=[int(x) for x in n_b])


def test_balanced_batch_without_binning_builds_its_tables_in_a_kernel_of_their_own():
    """MISIFT_BIN=0: no bin_detections launch to ride in, so frame_shares_kernel runs as its own launch; same records."""
    import os
    from cudasift_amd import capi
    frames = np.stack([synth_frame(9200 + i, 640, 360) for i in range(6)])
    c = capi.Context(0)
    c.set_knob("bin", 0)
    try:
        c.profile_enable(True)
        pts, n = c.extract_batch(frames, num_octaves=4, thresh=3.0, max_pts=8192)
        prof = c.profile_read()
        assert c.last_call_balanced() == 1 and prof["frame_shares"]["calls"] == 1 and "bin_detections" not in prof
    finally:
        c.close()
    ref, nref, _ = orc().extract_batch(frames, num_octaves=4, init_blur=1.0, thresh=3.0, max_pts=8192)
    assert np.array_equal(n, nref)
    for f in range(len(frames)):
        compare_points(ref[f, :nref[f]], pts[f, :n[f]], "balanced_nobin_f%d" % f, record)


def test_three_contexts_in_flight_equal_one_context(ctx):
    """`bench.py --contexts K` / INTEGRATION.md section 5: one context per batch in flight.  Three contexts (own stream,
    staging and scratch arena each) get six batches queued round-robin with nothing synchronising in between — their
    kernels share the GPU — and every batch must come out exactly as from one context working alone."""
    from cudasift_amd import capi
    B, mp, K, NBATCH = 4, 8192, 3, 6
    frames = np.stack([synth_frame(7100 + f) for f in range(2 * B)])
    d = [ctx.upload(frames[:B]), ctx.upload(frames[B:])]
    S = 4 * capi.scratch_floats(1920, 1080, 5, False) * B
    ref = []
    scratch0 = capi.DevBuf(S)
    was_fused = ctx.get_options().fused
    ctx.set_options(fused=1)
    try:
        for k in range(2):
            ref.append(_packed_async_1080p(ctx, d[k], B, scratch0, ctx.zeros(4 * (2 * B + 1)), ctx.zeros(576 * mp * B), mp))
    finally:
        ctx.set_options(fused=was_fused)
    cs = [capi.Context(0) for _ in range(K)]
    try:
        for c in cs:
            c.set_options(fused=1)
        scr = [capi.DevBuf(S) for _ in range(K)]
        cnts = [ctx.zeros(4 * (2 * B + 1)) for _ in range(NBATCH)]
        packs = [ctx.upload(np.full(576 * mp * B, 0xA5, np.uint8)) for _ in range(NBATCH)]
        ctx.sync()
        for k in range(NBATCH):                       # queue everything, then wait once
            c = cs[k % K]
            capi.check(capi.lib().misift_extract_batch_packed_async(
                c.h, d[k % 2].ptr, B, 1080 * 1920, 1920, 1080, 1920, 5, 1.0, 3.0, 0.0, scr[k % K].ptr, None, mp,
                cnts[k].ptr, cnts[k].ptr + 4 * B, packs[k].ptr), "misift_extract_batch_packed_async")
        for c in cs:
            c.sync()
        for k in range(NBATCH):
            ci = ctx.download(cnts[k], (2 * B + 1,), np.int32)
            counts, offs = ci[:B], ci[B:]
            rcounts, roffs, rrecs, _ = ref[k % 2]
            assert np.array_equal(counts, rcounts) and np.array_equal(offs, roffs), (k, counts, rcounts)
            recs = ctx.download(packs[k], (int(offs[B]),), capi.POINT_DTYPE)
            for f in range(B):
                assert _canon(recs[offs[f]:offs[f + 1]]) == _canon(rrecs[roffs[f]:roffs[f + 1]]), (k, f)
        record("contexts_in_flight", contexts=K, batches=NBATCH, keypoints=int(sum(int(r[0].sum()) for r in ref)))
    finally:
        for c in cs:
            c.close()


def test_batches_in_flight_inside_one_context(ctx):
    """misift_ctx_set_batches_in_flight(K): ONE context, K pipelines behind it.  Seven batches queued back to back with
    nothing synchronising in between (their kernels share the GPU) come out exactly as from the plain in-order context;
    misift_ctx_wait_batch orders a side stream behind the most recent batch; counters and the per-kernel profile follow the
    pipeline that took the batch; K back to 1 restores the in-order context."""
    from cudasift_amd import capi
    B, mp, K, NBATCH = 4, 8192, 3, 7
    frames = np.stack([synth_frame(7100 + f) for f in range(2 * B)])
    d = [ctx.upload(frames[:B]), ctx.upload(frames[B:])]
    S = 4 * capi.scratch_floats(1920, 1080, 5, False) * B
    ref = []
    scratch0 = capi.DevBuf(S)
    was_fused = ctx.get_options().fused
    ctx.set_options(fused=1)
    try:
        for k in range(2):
            ref.append(_packed_async_1080p(ctx, d[k], B, scratch0, ctx.zeros(4 * (2 * B + 1)), ctx.zeros(576 * mp * B), mp))
    finally:
        ctx.set_options(fused=was_fused)
    c = capi.Context(0)
    try:
        c.set_options(fused=1)
        c.set_batches_in_flight(K)
        assert capi.lib().misift_ctx_get_batches_in_flight(c.h) == K
        scr = [capi.DevBuf(S) for _ in range(K)]
        cnts = [ctx.zeros(4 * (2 * B + 1)) for _ in range(NBATCH)]
        packs = [ctx.upload(np.full(576 * mp * B, 0xA5, np.uint8)) for _ in range(NBATCH)]
        ctx.sync()
        c.profile_enable(True)
        for k in range(NBATCH):                       # queue everything, then wait once
            capi.check(capi.lib().misift_extract_batch_packed_async(
                c.h, d[k % 2].ptr, B, 1080 * 1920, 1920, 1080, 1920, 5, 1.0, 3.0, 0.0, scr[k % K].ptr, None, mp,
                cnts[k].ptr, cnts[k].ptr + 4 * B, packs[k].ptr), "misift_extract_batch_packed_async")
        c.sync()
        prof = c.profile_read()
        assert prof["descr_all"]["calls"] == NBATCH and prof["lowpass_down"]["calls"] == NBATCH     # merged over the pipelines
        c.profile_enable(False)
        last = (NBATCH - 1) % 2
        assert np.array_equal(np.stack([c.get_counters(f) for f in range(B)]), ref[last][3])        # the last batch's pipeline
        for k in range(NBATCH):
            ci = ctx.download(cnts[k], (2 * B + 1,), np.int32)
            counts, offs = ci[:B], ci[B:]
            rcounts, roffs, rrecs, _ = ref[k % 2]
            assert np.array_equal(counts, rcounts) and np.array_equal(offs, roffs), (k, counts, rcounts)
            recs = ctx.download(packs[k], (int(offs[B]),), capi.POINT_DTYPE)
            for f in range(B):
                assert _canon(recs[offs[f]:offs[f + 1]]) == _canon(rrecs[roffs[f]:roffs[f + 1]]), (k, f)
        # misift_ctx_wait_batch: a copy on the caller's own stream ordered behind the most recent batch
        import ctypes as C
        side = C.c_void_p()
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipStreamCreateWithFlags(C.byref(side), 1) == 0
        probe = ctx.zeros(4 * (2 * B + 1))
        capi.check(capi.lib().misift_extract_batch_packed_async(
            c.h, d[0].ptr, B, 1080 * 1920, 1920, 1080, 1920, 5, 1.0, 3.0, 0.0, scr[0].ptr, None, mp,
            cnts[0].ptr, cnts[0].ptr + 4 * B, packs[0].ptr), "misift_extract_batch_packed_async")
        c.wait_batch(side)
        assert hip.hipMemcpyAsync(C.c_void_p(probe.ptr), C.c_void_p(cnts[0].ptr), 4 * (2 * B + 1), 3, side) == 0     # device to device
        assert hip.hipStreamSynchronize(side) == 0
        assert np.array_equal(ctx.download(probe, (2 * B + 1,), np.int32)[:B], ref[0][0])
        hip.hipStreamDestroy(side)
        # back to the plain in-order context
        c.set_batches_in_flight(1)
        got = _packed_async_1080p(c, d[1], B, scratch0, ctx.zeros(4 * (2 * B + 1)), ctx.zeros(576 * mp * B), mp)
        assert np.array_equal(got[0], ref[1][0]) and np.array_equal(got[3], ref[1][3])
        record("batches_in_flight", pipelines=K, batches=NBATCH)
    finally:
        c.close()


def test_extract_batch_scaleup_and_u8(ctx, stereo):
    """misift_extract_batch_ex: the scaleUp path over a BATCH (the reference applies it per call, cudaSiftH.cu:118-132),
    fp32 and 8-bit frames — every frame against the oracle's scaleUp extraction."""
    crops = np.stack([stereo[0][300:540, 400:720], stereo[1][200:440, 100:420], stereo[0][600:840, 800:1120]])
    f8 = np.clip(np.rint(crops), 0, 255).astype(np.uint8)
    for imgs in (crops.astype(np.float32), f8):
        pts, n = ctx.extract_batch_ex(imgs, num_octaves=4, thresh=3.0, scale_up=True, max_pts=8192)
        for f in range(len(imgs)):
            ref, nref, cref = orc().extract(imgs[f].astype(np.float32), num_octaves=4, thresh=3.0, scale_up=True, max_pts=8192)
            assert nref == n[f] and nref > 200, (f, nref, n)
            compare_points(ref[:nref], pts[f][:n[f]], "extract_batch_scaleup_%s_f%d" % (imgs.dtype.name, f), record)
    ctx.set_options(fused=0)                         # dense kernels + the batched RescalePositions kernel
    try:
        pts2, n2 = ctx.extract_batch_ex(crops.astype(np.float32), num_octaves=4, thresh=3.0, scale_up=True, max_pts=8192)
    finally:
        ctx.set_options(fused=1)
    pts, n = ctx.extract_batch_ex(crops.astype(np.float32), num_octaves=4, thresh=3.0, scale_up=True, max_pts=8192)
    assert np.array_equal(n, n2)
    for f in range(len(crops)):
        assert _canon(pts[f][:n[f]]) == _canon(pts2[f][:n2[f]])


def test_improve_homography_device_vs_oracle(ctx, stereo):
    """misift_improve_homography (device version of the reference's host-side ImproveHomography, geomFuncs.cpp:6-72)
    against orc_improve_homography — itself pinned bit for bit to the reference's own geomFuncs.cpp: refined H,
    inlier count and every match_error identical.  Synthetic matches + the stereo chain of mainSift.cpp:72-78."""
    from synth import synth_matches
    for n, seed, loops in ((1500, 1, 5), (333, 2, 8), (20, 3, 2)):
        pts, Htrue, _ = synth_matches(n, inlier_frac=0.6, seed=seed)
        H0 = Htrue.copy()
        H0[0, 2] += 3.0
        H0[1, 2] -= 2.0
        ref = pts.copy()
        Ho, no = orc().improve_homography(ref, n, H0, loops, 0.0, 0.95, 3.0)
        d = ctx.upload(pts)
        Hg, ng = ctx.improve_homography(d.ptr, n, H0, loops, 0.0, 0.95, 3.0)
        got = ctx.download(d, (n,), pts.dtype)
        record("improve_homography_n%d" % n, numfit_oracle=no, numfit_hip=ng, H_equal=bool(np.array_equal(Ho, Hg)))
        assert ng == no
        assert np.array_equal(Ho.view(np.uint32), Hg.view(np.uint32))
        assert np.array_equal(ref["match_error"].view(np.uint32), got["match_error"].view(np.uint32))
    a, na, _ = ctx.extract(stereo[0], thresh=4.5)
    b, nb, _ = ctx.extract(stereo[1], thresh=4.5)
    m = ctx.match(a, na, b, nb)
    d = ctx.upload(m)
    orc().srand(3)
    H, _ = ctx.find_homography(d.ptr, na, num_loops=10000, min_score=0.0, max_ambiguity=0.80, thresh=5.0)
    ref = m.copy()
    Ho, no = orc().improve_homography(ref, na, H, 5, 0.0, 0.80, 3.0)                 # mainSift.cpp:78 arguments
    Hg, ng = ctx.improve_homography(d.ptr, na, H, 5, 0.0, 0.80, 3.0)
    assert ng == no and no > 100 and np.array_equal(Ho.view(np.uint32), Hg.view(np.uint32))


@pytest.mark.parametrize("fused", [1, 0])
def test_deterministic_mode_is_byte_identical_across_runs(ctx, fused):
    """options.deterministic: the reference appends records with atomics, so only the SET is
    reproducible run to run; with the switch on the order is fixed by the keypoints themselves — repeated runs and a
    run on another context return byte-identical arrays, and the set still equals the default mode's and the oracle's.
    fused = 1: the merged-octave path (total order where it bins its detections); fused = 0: the dense kernels (also the
    exact re-run after a candidate-list overflow), every segment sorted after the fact."""
    from cudasift_amd import capi
    imgs = np.stack([synth_frame(2100 + f, 960, 540) for f in range(3)])
    base_pts, base_n = ctx.extract_batch(imgs, num_octaves=5, thresh=3.0, max_pts=8192)
    was_fused = ctx.get_options().fused
    ctx.set_options(deterministic=1, fused=fused)
    try:
        runs = [ctx.extract_batch(imgs, num_octaves=5, thresh=3.0, max_pts=8192) for _ in range(3)]
    finally:
        ctx.set_options(deterministic=0, fused=was_fused)
    c2 = capi.Context(0)
    try:
        c2.set_options(deterministic=1, fused=fused)
        runs.append(c2.extract_batch(imgs, num_octaves=5, thresh=3.0, max_pts=8192))
    finally:
        c2.close()
    for pts, n in runs:
        assert np.array_equal(n, base_n)
    for f in range(len(imgs)):
        n = int(base_n[f])
        assert n > 300
        ref_bytes = runs[0][0][f][:n].tobytes()
        for pts, _ in runs[1:]:
            assert pts[f][:n].tobytes() == ref_bytes, "deterministic mode must give identical ORDER, frame %d" % f
        assert _canon(runs[0][0][f][:n]) == _canon(base_pts[f][:n])          # same set as the default mode
        o, no, _ = orc().extract(imgs[f], num_octaves=5, thresh=3.0, max_pts=8192)
        compare_points(o[:no], runs[0][0][f][:n], "deterministic_fused%d_f%d" % (fused, f), record)


def test_misift_devices_env():
    """MISIFT_DEVICES: the library's own device list.  "0" leaves one device that works; an index the box
    does not have leaves none and misift_ctx_create says so (read once per process: subprocesses)."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np\n"
            "from cudasift_amd import capi\n"
            "n = capi.device_count()\n"
            "print('count', n)\n"
            "try:\n"
            "    c = capi.Context(0)\n"
            "    pts, k, _ = c.extract(np.random.default_rng(1).random((64, 64), np.float32) * 255, num_octaves=2, thresh=1.0, max_pts=512)\n"
            "    print('ctx ok', k >= 0)\n"
            "    c.close()\n"
            "except capi.MisiftError as e:\n"
            "    print('ctx failed', e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for val in ("0", "63", ""):
        env = dict(os.environ, PYTHONPATH=root)
        if val:
            env["MISIFT_DEVICES"] = val
        else:
            env.pop("MISIFT_DEVICES", None)
        outs[val] = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300).stdout
    assert "count 1" in outs["0"] and "ctx ok True" in outs["0"], outs
    assert "count 0" in outs["63"] and "ctx failed" in outs["63"] and "no HIP device visible" in outs["63"], outs
    assert "ctx ok True" in outs[""], outs


def test_device_elementary_functions_have_the_oracles_bits(ctx):
    """The device copies of det_exp2 / det_atan2 / det_exp / det_sincos (test-only entry misift_test_elementary) return
    the oracle's bits on 10^6 inputs each — the premise of the bit-identical scale / orientation assertions above; the
    oracle's copies are checked against float64 libm in tests/test_oracle_cpu.py::test_det_functions_accuracy."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    x = np.concatenate([rng.uniform(-0.12, 0.12, n // 2), rng.uniform(-150, 150, n // 2 - 3),
                        [np.nan, -200.0, 200.0]]).astype(np.float32)
    a, b = ctx.test_elementary(0, x), orc().det_eval(0, x)
    assert np.array_equal(a.view(np.uint32)[:-3], b.view(np.uint32)[:-3]) and np.isnan(a[-3]) and a[-2] == 0.0 and a[-1] == b[-1]
    gx = np.concatenate([rng.uniform(-255, 255, n // 2), rng.normal(0, 1e-3, n // 2 - 4), [0.0, -0.0, 1.0, -1.0]]).astype(np.float32)
    gy = np.concatenate([rng.uniform(-255, 255, n // 2), rng.normal(0, 1e-3, n // 2 - 4), [0.0, 0.0, 0.0, -0.0]]).astype(np.float32)
    assert np.array_equal(ctx.test_elementary(1, gx, gy).view(np.uint32), orc().det_eval(1, gx, gy).view(np.uint32))
    x = np.concatenate([-rng.uniform(0, 1, n // 2), -rng.uniform(0, 100, n // 2)]).astype(np.float32)
    assert np.array_equal(ctx.test_elementary(2, x).view(np.uint32), orc().det_eval(2, x).view(np.uint32))
    x = rng.uniform(0.0, 2.0 * 3.1415, n).astype(np.float32)
    s, c = ctx.test_elementary(3, x)
    so, co = orc().det_eval(3, x)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32)) and np.array_equal(c.view(np.uint32), co.view(np.uint32))


@pytest.mark.parametrize("w,h,noct", [(1000, 750, 5), (1283, 721, 5), (642, 483, 4), (250, 187, 3)])
def test_extract_ragged_widths_take_the_fast_kernels(ctx, w, h, noct):
    """Widths that are not a multiple of 4 at SOME pyramid level (1000 -> 500 -> 250 -> 125 -> 62; 1283 at every level):
    r02 sent the whole image through the generic kernels (separate LowPass + ScaleDown, scalar edge loads: 0.73x).  r03:
    the fused prefilter, the ScaleDowns and the merged-octave scan keep their dwordx4 row loads with a ragged last quad —
    the profile shows lowpass_down (not lowpass) — and the records equal the oracle's."""
    from cudasift_amd import capi
    imgs = np.stack([synth_frame(8200 + f, w, h) for f in range(2)])
    c = capi.Context(0)
    try:
        c.profile_enable(True)
        pts, n = c.extract_batch(imgs, num_octaves=noct, thresh=2.5, max_pts=16384)
        prof = c.profile_read()
        assert "lowpass_down" in prof and "lowpass" not in prof, prof.keys()
    finally:
        c.close()
    for f in range(2):
        ref, nref, _ = orc().extract(imgs[f], noct, 1.0, 2.5, max_pts=16384)
        assert n[f] == nref and nref > 100
        compare_points(ref[:nref], pts[f, :nref], "ragged_%dx%d_f%d" % (w, h, f), record)


@pytest.mark.parametrize("reach", ["9.0", "13.0"])
def test_descr_big_path_on_ordinary_keypoints(reach):
    """descr_big_kernel takes the keypoints whose descriptor window does not fit descr_all's 40 x 40 LDS tile (scale > 2.12
    at its level: only reachable through the refinement's unclamped fallback — none among 40 synthetic frames + the stereo
    pair, so the path was all but untested).  MISIFT_TEST_PATCH_REACH lowers the limit: with 13 texels every keypoint of
    scale > 1.5, with 9 every keypoint of scale > 1.0 goes down the global-memory path — single call, batch and the packed
    entry point must still return the oracle's records."""
    import os
    from cudasift_amd import capi
    c = capi.Context(0)
    c.set_knob("patch_reach", float(reach))
    try:
        img = synth_frame(31, 960, 540)
        ref, nref, cref = orc().extract(img, 5, 1.0, 2.5)
        got, n, cnt = c.extract(img, num_octaves=5, init_blur=1.0, thresh=2.5)
        big = int(c.get_counter_block(0)[48])                                   # CNT_BIG: keypoints deferred to descr_big
        assert n == nref and np.array_equal(cnt, cref)
        assert big > (0.05 if reach == "13.0" else 0.3) * n, (big, n)           # the rare path is the busy one here
        # a synchronous single call ends with descr_all (its last workgroup exports the counters, r05); descr_big is
        # launched only because the host found deferred keypoints in them — this IS that case
        assert c.descr_big_fallbacks() == 1
        compare_points(ref[:nref], got[:n], "descr_big_reach%s" % reach, record)
        frames = np.stack([synth_frame(32 + i, 640, 360) for i in range(6)])
        rb, nb, _ = orc().extract_batch(frames, num_octaves=4, init_blur=1.0, thresh=3.0, max_pts=8192)
        gb, gn = c.extract_batch(frames, num_octaves=4, thresh=3.0, max_pts=8192)[:2]
        assert np.array_equal(gn, nb)
        for f in range(len(frames)):
            compare_points(rb[f, :nb[f]], gb[f, :gn[f]], "descr_big_batch_reach%s_f%d" % (reach, f), record)
        assert c.descr_big_fallbacks() == 1                                     # (batches launch descr_big unconditionally)
        record("descr_big_path/reach" + reach, deferred=big, keypoints=int(n))
    finally:
        c.close()


def test_single_call_ends_with_the_descriptor_launch(ctx):
    """The single-call path is five dependent dispatches (r05): descr_all's last workgroup hands the counters to the host, no
    descr_big launch unless a keypoint was deferred to it; MISIFT_FOLD_TAIL=0 is the r04 sequence.  Same records either way."""
    import os
    from cudasift_amd import capi
    img = synth_frame(44, 1280, 960)
    ref, nref, cref = orc().extract(img, 5, 1.0, 3.0)
    out = {}
    for fold in ("1", "0"):
        c = capi.Context(0)
        c.set_knob("fold_tail", int(fold))
        try:
            for rep in range(3):
                got, n, cnt = c.extract(img, num_octaves=5, init_blur=1.0, thresh=3.0)
                assert n == nref and np.array_equal(cnt, cref), (fold, rep)
            out[fold] = int(n)
            assert c.descr_big_fallbacks() == 0
            compare_points(ref[:nref], got[:n], "fold_tail%s" % fold, record)
        finally:
            c.close()
    record("single_call_fold_tail", keypoints=out["1"])
    assert out["1"] == out["0"]            # (the number of GPU activities per call is in profiles/r05_single_call_budget_*.txt)

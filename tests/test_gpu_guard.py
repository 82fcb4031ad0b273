"""Out-of-bounds and uninitialised-read policing of every device buffer (SURVEY section 5 "race detection / sanitizers":
guard-paged scratch arenas in the test build; GPU AddressSanitizer is not available on this pool).

In guard mode (capi.set_guard / misift_test_set_guard) EVERY device allocation — the caller's images, scratch arenas,
record arrays, packed outputs, count arrays, matcher inputs and results (misift_malloc), and the library's own counters,
candidate lists, detection staging, block tables, matcher scratch, 32-per-block masks and pipeline buffers — carries 64 KiB
of a byte pattern on either side of the payload, and the payload starts out as 0xFF bytes (NaN as a float, -1 as an int, a
huge count as an unsigned).  Each scenario below runs on a fresh guarded context, then
  * every band of every live allocation is intact (nothing wrote outside its buffer — the reference itself does:
    ScaleDown writes rows beyond h/2, cudaSiftD.cu:116-166, FindMaxCorr10 rows past n1, matching.cu:391-395; App. B #11, #13),
  * the results equal those of the ordinary (unguarded, zero-initialised) session context on the same input — nothing read
    memory it had not written."""
import numpy as np
import pytest

from conftest import record
from synth import descriptors_to_points, synth_descriptors, synth_frame

pytestmark = pytest.mark.gpu


WRITTEN = ("xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")


def _canon(recs):
    """The fields ExtractSift writes (the match fields of a record array stay whatever the caller's memory held — here the
    poison — exactly as with the reference), in a canonical record order."""
    k = [recs[f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")]
    r = recs[np.lexsort(k)]
    return b"".join(np.ascontiguousarray(r[f]).tobytes() for f in WRITTEN)


@pytest.fixture
def guarded():
    from cudasift_amd import capi
    old = capi.set_guard(True)
    c = capi.Context(0)
    c.poison_outputs = True
    try:
        yield c
    finally:
        c.close()
        capi.set_guard(old)


def _same(a, na, b, nb, what):
    assert na == nb, (what, na, nb)
    assert _canon(a[:na]) == _canon(b[:nb]), what


def _intact(name, **kw):
    from cudasift_amd import capi
    n = capi.check_guards()
    assert n >= 3, n
    record("guard/" + name, allocations_checked=n, **kw)


@pytest.mark.parametrize("w,h,noct,th", [(1917, 1079, 5, 3.0), (333, 251, 3, 2.5), (97, 61, 4, 1.0), (13, 9, 2, 0.2), (3, 2, 1, 0.1),
                                         (1920, 1080, 5, 3.0)])
def test_single_call_ragged_and_tiny(ctx, guarded, w, h, noct, th):
    img = synth_frame(700 + w, w, h) if w * h >= 64 else np.random.default_rng(w).uniform(0, 255, (h, w)).astype(np.float32)
    a, na, ca = guarded.extract(img, num_octaves=noct, thresh=th)
    _intact("single_%dx%d" % (w, h), numPts=na)
    b, nb, cb = ctx.extract(img, num_octaves=noct, thresh=th)
    assert np.array_equal(ca, cb)
    _same(a, na, b, nb, (w, h))


def test_single_call_4096x3072(ctx, guarded):
    img = synth_frame(4242, width=4096, height=3072)
    a, na, ca = guarded.extract(img, num_octaves=5, thresh=3.0)
    _intact("single_4096x3072", numPts=na)
    b, nb, cb = ctx.extract(img, num_octaves=5, thresh=3.0)
    assert na > 5000 and np.array_equal(ca, cb)
    _same(a, na, b, nb, "4096x3072")


@pytest.mark.parametrize("up,u8", [(False, False), (True, False), (False, True), (True, True)])
def test_batch_scaleup_and_u8(ctx, guarded, up, u8):
    imgs = np.stack([synth_frame(800 + f, 322, 250) for f in range(5)])
    if u8:
        imgs = np.clip(np.rint(imgs), 0, 255).astype(np.uint8)
    a, na = guarded.extract_batch_ex(imgs, num_octaves=4, thresh=3.0, scale_up=up, max_pts=8192)
    _intact("batch_up%d_u8%d" % (up, u8), records=int(na.sum()))
    b, nb = ctx.extract_batch_ex(imgs, num_octaves=4, thresh=3.0, scale_up=up, max_pts=8192)
    for f in range(len(imgs)):
        _same(a[f], na[f], b[f], nb[f], (up, u8, f))


def _packed_async(c, frames, mp):
    from cudasift_amd import capi
    B, h, w = frames.shape
    d = c.upload(frames)
    sc = capi.DevBuf(4 * capi.scratch_floats(w, h, 5, False) * B)
    cnt = c.zeros(4 * (2 * B + 1))
    packed = c.zeros(576 * mp * B)
    capi.check(capi.lib().misift_extract_batch_packed_async(c.h, d.ptr, B, h * w, w, h, w, 5, 1.0, 3.0, 0.0, sc.ptr, None, mp,
                                                            cnt.ptr, cnt.ptr + 4 * B, packed.ptr), "misift_extract_batch_packed_async")
    c.sync()
    ci = c.download(cnt, (2 * B + 1,), np.int32)
    return ci[:B].copy(), ci[B:].copy(), c.download(packed, (int(ci[2 * B]),), capi.POINT_DTYPE)


def test_the_timed_entry_point_64_frames_of_1080p(ctx, guarded):
    """bench.py's call: 64 x 1920x1080 through misift_extract_batch_packed_async (packed-only output), twice on the same
    context (the second call finds the first one's leftovers in every internal buffer, not the poison)."""
    frames = np.stack([synth_frame(40 + (f % 8)) for f in range(64)])
    for rep in range(2):
        ca, oa, ra = _packed_async(guarded, frames, 4096)
        _intact("packed_async_64x1080p_call%d" % rep, records=int(oa[-1]))
    cb, ob, rb = _packed_async(ctx, frames, 4096)
    assert np.array_equal(ca, cb) and np.array_equal(oa, ob) and ca.min() > 1000
    for f in range(64):
        assert _canon(ra[oa[f]:oa[f + 1]]) == _canon(rb[ob[f]:ob[f + 1]]), f


def test_candidate_overflow_rerun(ctx, guarded):
    """Frames 1 and 3 of 4 flood their candidate lists (white noise, tiny threshold): the exact re-run on the dense kernels."""
    rng = np.random.default_rng(5)
    imgs = np.stack([synth_frame(900 + f, 256, 256) if f % 2 == 0 else rng.integers(0, 256, (256, 256)).astype(np.float32)
                     for f in range(4)])
    a, na = guarded.extract_batch(imgs, num_octaves=3, thresh=0.05, max_pts=32768)
    _intact("overflow_rerun", records=int(na.sum()))
    b, nb = ctx.extract_batch(imgs, num_octaves=3, thresh=0.05, max_pts=32768)
    assert na[1] > 2000
    for f in range(4):
        _same(a[f], na[f], b[f], nb[f], f)


def test_reference_cap_and_dense_kernels(ctx, guarded):
    img = synth_frame(950, 640, 360)
    for kw in (dict(reference_cap=1), dict(fused=0)):
        saved_g, saved_c = guarded.get_options(), ctx.get_options()
        guarded.set_options(**kw); ctx.set_options(**kw)
        try:
            a, na, ca = guarded.extract(img, num_octaves=4, thresh=1.0)
            _intact("options_%s" % list(kw)[0], numPts=na)
            b, nb, cb = ctx.extract(img, num_octaves=4, thresh=1.0)
        finally:
            guarded.set_options(reference_cap=saved_g.reference_cap, fused=saved_g.fused)
            ctx.set_options(reference_cap=saved_c.reference_cap, fused=saved_c.fused)
        assert np.array_equal(ca, cb)
        _same(a, na, b, nb, kw)


def test_descr_big_path(ctx):
    """Keypoints sent down descr_big (global-memory descriptor path) by the patch-reach test hook."""
    from cudasift_amd import capi
    old = capi.set_guard(True)
    g = capi.Context(0)
    try:
        g.set_knob("patch_reach", 9.0)
        g.poison_outputs = True
        img = synth_frame(31, 960, 540)
        a, na, ca = g.extract(img, num_octaves=5, thresh=2.5)
        assert int(g.get_counter_block(0)[48]) > 0.3 * na
        frames = np.stack([synth_frame(32 + i, 640, 360) for i in range(6)])
        ab, nab = g.extract_batch(frames, num_octaves=4, thresh=3.0, max_pts=8192)
        _intact("descr_big", numPts=na)
        b, nb, cb = ctx.extract(img, num_octaves=5, thresh=2.5)
        assert np.array_equal(ca, cb)
        assert na == nb
        # (descr_big sums a descriptor's votes in another order than descr_all: <= 1e-6 instead of the same bytes)
        ka = np.lexsort([a[:na][f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")])
        kb = np.lexsort([b[:nb][f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")])
        for f in ("xpos", "ypos", "scale", "orientation", "sharpness", "edgeness"):
            assert np.array_equal(a[:na][ka][f], b[:nb][kb][f]), f
        assert np.abs(a[:na][ka]["data"] - b[:nb][kb]["data"]).max() <= 1e-6
        bb, nbb = ctx.extract_batch(frames, num_octaves=4, thresh=3.0, max_pts=8192)
        assert np.array_equal(nab, nbb)
    finally:
        g.close()
        capi.set_guard(old)


@pytest.mark.parametrize("src_u8", [True, False])
def test_pipe(ctx, guarded, src_u8):
    from cudasift_amd import capi
    h, w, B, nb = 272, 480, 3, 4
    frames = np.stack([np.clip(np.rint(synth_frame(300 + i, width=w, height=h)), 0, 255).astype(np.uint8) for i in range(B * nb - 1)])
    src = frames if src_u8 else frames.astype(np.float32)
    pin = capi.PinnedArray(src.shape, src.dtype)
    pin.array[...] = src
    out = capi.PinnedArray((B * 4096,), capi.POINT_DTYPE)
    pipe = capi.Pipe(guarded, w, h, B, src_u8=src_u8, thresh=2.0, max_pts=4096, depth=2)
    esz = src.dtype.itemsize * h * w
    got = []
    for k in range(nb):
        if pipe.pending() == 2:
            counts, nrec = pipe.collect(out.ptr, B * 4096)
            got.append((counts, out.array[:nrec].copy()))
        pipe.submit(pin.ptr + k * B * esz, min(B, len(src) - k * B))
    while pipe.pending():
        counts, nrec = pipe.collect(out.ptr, B * 4096)
        got.append((counts, out.array[:nrec].copy()))
    _intact("pipe_u8%d" % src_u8, batches=nb)
    pipe.close()
    ref_fn = ctx.extract_batch_u8 if src_u8 else ctx.extract_batch
    for k, (counts, recs) in enumerate(got):
        n = min(B, len(src) - k * B)
        rp, rn = ref_fn(src[k * B:k * B + n], thresh=2.0, max_pts=4096)
        assert np.array_equal(counts, rn)
        off = 0
        for f in range(n):
            assert _canon(recs[off:off + rn[f]]) == _canon(rp[f, :rn[f]])
            off += rn[f]


@pytest.mark.parametrize("n1,n2", [(1000, 1500), (37, 2049), (2051, 33), (64, 64), (1, 1), (4099, 4101)])
def test_matcher_ragged_sizes(ctx, guarded, n1, n2):
    """n1 % 32 != 0 and n2 % 32 != 0 (the reference's FindMaxCorr10 writes rows past n1 and ignores the last n2 % 32
    columns, matching.cu:391-395, App. B #13): exactly n1 records change, nothing outside them."""
    from cudasift_amd import capi
    a = descriptors_to_points(synth_descriptors(n1, 7 + n1), capi.POINT_DTYPE)
    b = descriptors_to_points(synth_descriptors(n2, 8 + n2), capi.POINT_DTYPE)
    got = guarded.match(a, n1, b, n2)
    _intact("match_%dx%d" % (n1, n2))
    want = ctx.match(a, n1, b, n2)
    assert got.tobytes() == want.tobytes()
    # rows only: a row block in the middle of a larger array leaves the other records untouched
    if n1 >= 64:
        part = guarded.match(a, n1, b, n2, row_begin=17, row_count=n1 - 40)
        _intact("match_rows_%dx%d" % (n1, n2))
        keep = np.r_[0:17, n1 - 23:n1]
        assert part[keep].tobytes() == a[keep].tobytes()
        assert part[17:n1 - 23].tobytes() == want[17:n1 - 23].tobytes()


def test_find_homography_and_improve(ctx, guarded):
    from cudasift_amd import capi
    from oracle import pyoracle as orc
    from synth import synth_matches
    m, _, _ = synth_matches(3001, seed=5, dtype=capi.POINT_DTYPE)
    res = []
    for c in (guarded, ctx):
        d = c.upload(m)
        orc.srand(1)
        H, nm = c.find_homography(d.ptr, 3001, num_loops=1999, min_score=0.85, max_ambiguity=0.95, thresh=5.0)
        H2, nf = c.improve_homography(d.ptr, 3001, H, num_loops=5, min_score=0.0, max_ambiguity=0.80, thresh=3.0)
        res.append((H.copy(), nm, H2.copy(), nf))
        if c is guarded:
            _intact("homography")
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]


def test_match_sharded_and_gather_single_rank(ctx, guarded):
    from cudasift_amd import capi
    comm = capi.Comm(guarded, 1, 0, capi.comm_unique_id())
    try:
        n1, n2 = 515, 389
        p1 = descriptors_to_points(synth_descriptors(n1, 31), capi.POINT_DTYPE)
        p2 = descriptors_to_points(synth_descriptors(n2, 32), capi.POINT_DTYPE)
        d1, d2 = guarded.upload(p1), guarded.upload(p2)
        all2 = guarded.zeros(capi.COLUMN_DTYPE.itemsize * n2)
        res = guarded.zeros(12 * n1)
        comm.match_sharded(d1.ptr, n1, d2.ptr, n2, all2.ptr, res.ptr)
        rows = guarded.download(d1, (n1,), capi.POINT_DTYPE)
        _intact("match_sharded")
        # gather of a packed batch
        frames = np.stack([synth_frame(9000 + i, width=480, height=272) for i in range(3)]).astype(np.float32)
        B, h, w = frames.shape
        d = guarded.upload(frames)
        sc = capi.DevBuf(4 * capi.scratch_floats(w, h, 5, False) * B)
        cnt = guarded.zeros(4 * (2 * B + 1))
        packed = guarded.zeros(576 * 4096 * B)
        capi.check(capi.lib().misift_extract_batch_packed_async(guarded.h, d.ptr, B, h * w, w, h, w, 5, 1.0, 2.0, 0.0, sc.ptr, None,
                                                                4096, cnt.ptr, cnt.ptr + 4 * B, packed.ptr), "packed_async")
        comm.gather_post(0, cnt.ptr, B, packed.ptr)
        recv = guarded.zeros(576 * 4096 * B)
        counts, offs = comm.gather_complete(0, B, 0, recv.ptr, 4096 * B)
        _intact("gather_single_rank", records=int(offs[-1]))
    finally:
        comm.close()
    want = ctx.match(p1, n1, p2, n2)
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert np.array_equal(rows[f], want[f]), f
    rp, rn = ctx.extract_batch(frames, thresh=2.0, max_pts=4096)
    assert np.array_equal(counts[0], rn)


def test_the_guard_catches_a_stray_write(ctx):
    """The policing itself: a store one float behind a guarded buffer (and one in front of another) is reported."""
    from cudasift_amd import capi
    old = capi.set_guard(True)
    try:
        buf = capi.DevBuf(4 * 1000)
        capi.check_guards()
        one = np.array([1.0], np.float32)
        capi.check(capi.lib().misift_copy_h2d(ctx.h, buf.ptr + 4 * 1000, one.ctypes.data, 4), "copy")      # first float behind
        ctx.sync()
        buf.free()                                              # a buffer is verified when it is freed, too ...
        with pytest.raises(capi.MisiftError, match="BEHIND"):
            capi.check_guards()                                 # ... and the next check reports it
        capi.check_guards()
        buf = capi.DevBuf(4096)
        capi.check(capi.lib().misift_copy_h2d(ctx.h, buf.ptr - 4, one.ctypes.data, 4), "copy")             # last float in front
        ctx.sync()
        with pytest.raises(capi.MisiftError, match="IN FRONT"):
            capi.check_guards()                                 # a live buffer
        buf.free()
        with pytest.raises(capi.MisiftError, match="IN FRONT"):
            capi.check_guards()
        capi.check_guards()
        # and the poison: a fresh guarded payload reads as NaN
        fresh = capi.DevBuf(64)
        assert np.isnan(ctx.download(fresh, (16,), np.float32)).all()
    finally:
        capi.set_guard(old)

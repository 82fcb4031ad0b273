"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py (the checker, never the thing measured or shipped).  The product
package `cudasift_amd` never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# numpy view of the 576-byte SiftPoint record (reference cudaSift.h:6-22)
POINT_DTYPE = np.dtype([
    ("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("sharpness", "<f4"),
    ("edgeness", "<f4"), ("orientation", "<f4"), ("score", "<f4"), ("ambiguity", "<f4"),
    ("match", "<i4"), ("match_xpos", "<f4"), ("match_ypos", "<f4"), ("match_error", "<f4"),
    ("subsampling", "<f4"), ("empty", "<f4", (3,)), ("data", "<f4", (128,)),
])
assert POINT_DTYPE.itemsize == 576


class OrcStats(C.Structure):
    _fields_ = [("tile_overflows", C.c_long), ("nan_guards", C.c_long), ("empty_hists", C.c_long),
                ("oob_votes", C.c_long), ("capacity_drops", C.c_long)]


_lib = None


def build():
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def cpu_budget():
    """CPUs this process may actually use (affinity mask and cgroup CPU quota): the GPU boxes of this pool show 256 logical
    CPUs behind a quota of 16, and an OpenMP team of 256 spinning threads on 16 CPUs' worth of time makes every call of the
    oracle take seconds whatever the image size (r06: ~4 s per orc.extract of a 120 x 120 image; half of the GPU suite's time)."""
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


def lib():
    global _lib
    if _lib is None:
        # the OpenMP runtime reads these when it is loaded: size the team for the CPUs we may use, and let idle threads sleep
        os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        path = os.environ.get("ORACLE_LIB") or os.path.join(HERE, "liboracle.so")     # ORACLE_LIB: the sanitizer flavour (make -C oracle asan)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        vp = C.c_void_p
        L.orc_lowpass_taps.argtypes = [C.c_float, fp]
        L.orc_scaledown_taps.argtypes = [C.c_float, fp]
        L.orc_laplace_taps.argtypes = [C.c_int, fp]
        L.orc_lowpass.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_float]
        L.orc_scaledown.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orc_scaleup.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orc_laplace.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
        L.orc_findpoints.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, vp, C.c_int, C.c_int]
        L.orc_findpoints.restype = C.c_int
        L.orc_tex2d.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc_tex2d.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_orientations.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                       C.POINTER(C.c_uint), C.c_int, C.c_int]
        L.orc_descriptors.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int]
        L.orc_scratch_floats.argtypes = [C.c_int] * 4
        L.orc_scratch_floats.restype = C.c_size_t
        L.orc_extract.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                  C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint)]
        L.orc_extract.restype = C.c_int
        L.orc_extract_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp,
                                        C.c_int, C.c_int, vp, vp, C.c_int, C.c_int]
        L.orc_set_contract.argtypes = [C.c_int]
        L.orc_get_contract.restype = C.c_int
        L.orc_dot128.argtypes = [vp, vp]
        L.orc_dot128.restype = C.c_float
        L.orc_match.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
        L.orc_match_rows.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
        L.orc_match_argmax.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        L.orc_find_homography.argtypes = [vp, C.c_int, fp, C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_find_homography.restype = C.c_int
        L.orc_improve_homography.argtypes = [vp, C.c_int, fp, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_improve_homography.restype = C.c_int
        L.orc_det_eval.argtypes = [C.c_int, vp, vp, vp, vp, C.c_long]
        L.orc_stats_get.argtypes = [C.POINTER(OrcStats)]
        L.orc_sizeof_point.restype = C.c_int
        assert L.orc_sizeof_point() == 576
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- taps
def lowpass_taps(sigma):
    k = np.zeros(9, np.float32)
    lib().orc_lowpass_taps(sigma, k.ctypes.data_as(C.POINTER(C.c_float)))
    return k


def scaledown_taps(variance=0.5):
    k = np.zeros(5, np.float32)
    lib().orc_scaledown_taps(variance, k.ctypes.data_as(C.POINTER(C.c_float)))
    return k


def laplace_taps(num_octaves):
    k = np.zeros(8 * 12 * 16, np.float32)
    lib().orc_laplace_taps(num_octaves, k.ctypes.data_as(C.POINTER(C.c_float)))
    return k


# ---------------------------------------------------------------- stages (2-D arrays, pitch == width)
def lowpass(img, sigma):
    img = _f32(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().orc_lowpass(_p(img), w, h, w, _p(out), w, sigma)
    return out


def scaledown(img):
    img = _f32(img)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().orc_scaledown(_p(img), w, h, w, _p(out), w // 2)
    return out


def scaleup(img):
    img = _f32(img)
    h, w = img.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    lib().orc_scaleup(_p(img), w, h, w, _p(out), 2 * w)
    return out


def laplace(base, num_octaves, octave):
    """7 DoG planes [7,h,w] of one octave base image."""
    base = _f32(base)
    h, w = base.shape
    taps = laplace_taps(num_octaves)
    dog = np.empty((7, h, w), np.float32)
    lib().orc_laplace(_p(base), w, h, w, _p(taps), octave, _p(dog))
    return dog


def findpoints(dog, thresh, subsampling=1.0, lowest_scale=0.0, max_pts=32768, edge_limit=10.0):
    dog = _f32(dog)
    _, h, w = dog.shape
    pts = np.zeros(max_pts, POINT_DTYPE)
    n = lib().orc_findpoints(_p(dog), w, h, w, thresh, edge_limit, 1.0 / 5, lowest_scale, subsampling,
                             _p(pts), 0, max_pts)
    return pts, n


def orientations(base, pts, first, last, max_pts, fracbits=8):
    base = _f32(base)
    h, w = base.shape
    dup = C.c_uint(last)
    lib().orc_orientations(_p(base), w, h, w, _p(pts), first, last, C.byref(dup), max_pts, fracbits)
    return int(dup.value)


def descriptors(base, pts, first, last, subsampling=1.0, fracbits=8):
    base = _f32(base)
    h, w = base.shape
    lib().orc_descriptors(_p(base), w, h, w, _p(pts), first, last, subsampling, fracbits)


def det_eval(fn, x, y=None):
    """The oracle's written-out elementary functions on arrays: fn 0 = exp2(x), 1 = atan2(y, x), 2 = exp(x),
    3 = sincos(x) -> (sin, cos)."""
    x = _f32(x)
    y = _f32(y) if y is not None else x
    out, out2 = np.empty_like(x), np.empty_like(x)
    lib().orc_det_eval(fn, _p(x), _p(y), _p(out), _p(out2), x.size)
    return (out, out2) if fn == 3 else out


def tex2d(img, x, y, fracbits=8):
    img = _f32(img)
    h, w = img.shape
    return float(lib().orc_tex2d(_p(img), w, h, w, x, y, fracbits))


# ---------------------------------------------------------------- whole path
def extract(img, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, scale_up=False,
            max_pts=32768, fracbits=8, fix_numpts=False):
    """orc_extract on a host image.  Returns (points[max_pts] structured array, numPts, counters[17])."""
    img = _f32(img)
    h, w = img.shape
    pts = np.zeros(max_pts, POINT_DTYPE)
    cnt = (C.c_uint * 17)()
    n = lib().orc_extract(_p(img), w, h, w, num_octaves, init_blur, thresh, lowest_scale, int(scale_up),
                          _p(pts), max_pts, fracbits, int(fix_numpts), cnt)
    return pts, n, np.array(list(cnt), dtype=np.uint32)


def descriptor_bounds(img, pts, n, num_octaves=5, init_blur=1.0, ulps=3.0, dtheta_deg=None, scale_up=False, coord_scale=None):
    """orc_descriptor_bounds2: for the first n records of one ExtractSift call on `img`, the largest change of every
    descriptor element that a last-bit difference of the sample coordinates can cause through the 8-bit texture weights
    (see sift_oracle.c).  dtheta_deg[n]: difference of the two sides' orientations (it turns the whole sample grid).
    scale_up: the call up-sampled the image first; coord_scale[n]: 2 for the records RescalePositions has halved.
    Returns (bound[n,128], flips[n], wraps[n])."""
    img = _f32(img)
    h, w = img.shape
    pts = np.ascontiguousarray(pts[:n])
    bound = np.zeros((n, 128), np.float32)
    flips, wraps = np.zeros(n, np.int32), np.zeros(n, np.int32)
    L = lib()
    L.orc_descriptor_bounds2.restype = None
    dth = None if dtheta_deg is None else np.ascontiguousarray(dtheta_deg, np.float32)
    cs = None if coord_scale is None else np.ascontiguousarray(coord_scale, np.float32)
    L.orc_descriptor_bounds2(_p(img), w, h, w, num_octaves, C.c_float(init_blur), int(bool(scale_up)), _p(pts), n,
                             None if cs is None else _p(cs), C.c_float(ulps), None if dth is None else _p(dth),
                             _p(bound), _p(flips), _p(wraps))
    return bound, flips, wraps


def descriptor_explain(img, pts, targets, target_recs=None, num_octaves=5, init_blur=1.0, ulps=3.0, scale_up=False,
                       coord_scale=None, tol=1e-5, max_flips=48):
    """orc_descriptor_explains: for each record, search for the set of tie-weight roundings / seam decisions that turns
    this side's descriptor — sampled on the OTHER side's grid: position, scale and orientation of target_recs (structured
    records, or an array of orientations only) — into `targets` (the other side's descriptor).
    Returns (residual[n], overrides used[n], candidates[n])."""
    img = _f32(img)
    h, w = img.shape
    n = len(pts)
    pts = np.ascontiguousarray(pts)
    targets = np.ascontiguousarray(targets, np.float32)
    res = np.zeros(n, np.float32)
    nset, ncand = np.zeros(n, np.int32), np.zeros(n, np.int32)
    L = lib()
    L.orc_descriptor_explains.restype = None
    geom = None
    if target_recs is not None:
        geom = np.zeros((n, 4), np.float32)
        if getattr(target_recs, "dtype", None) is not None and target_recs.dtype.names:
            for k, f in enumerate(("xpos", "ypos", "scale", "orientation")):
                geom[:, k] = target_recs[f]
        else:                                             # orientations only: our own position and scale
            geom[:, 0], geom[:, 1], geom[:, 2] = pts["xpos"], pts["ypos"], pts["scale"]
            geom[:, 3] = np.asarray(target_recs, np.float32)
    cs = None if coord_scale is None else np.ascontiguousarray(coord_scale, np.float32)
    L.orc_descriptor_explains(_p(img), w, h, w, num_octaves, C.c_float(init_blur), int(bool(scale_up)), _p(pts), n,
                              None if cs is None else _p(cs), _p(targets), None if geom is None else _p(geom),
                              C.c_float(ulps), C.c_float(tol), int(max_flips), _p(res), _p(nset), _p(ncand))
    return res, nset, ncand


def extract_batch(imgs, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, max_pts=32768, fracbits=8,
                  outer_threads=None, inner_threads=1):
    """orc_extract_batch: frames [B,h,w] processed one per OpenMP thread.
    Returns (points[B,max_pts], numPts[B], counters[B,17])."""
    imgs = _f32(imgs)
    B, h, w = imgs.shape
    pts = np.zeros((B, max_pts), POINT_DTYPE)
    n = np.zeros(B, np.int32)
    cnt = np.zeros((B, 17), np.uint32)
    if outer_threads is None:
        outer_threads = min(B, os.cpu_count() or 1)
    lib().orc_extract_batch(_p(imgs), B, w, h, num_octaves, init_blur, thresh, lowest_scale, _p(pts), max_pts,
                            fracbits, _p(n), _p(cnt), outer_threads, inner_threads)
    return pts, n, cnt


class contract:
    """Context manager: run the oracle in another contraction mode (0 = plain = what the HIP kernels implement,
    1 = nvcc-style fused multiply-adds outside the separable filters; see sift_oracle.c header)."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.saved = lib().orc_get_contract()
        lib().orc_set_contract(self.mode)
        return self

    def __exit__(self, *exc):
        lib().orc_set_contract(self.saved)
        return False


def match(pts1, n1, pts2, n2, full=False, exact=False):
    """In-place MatchSiftData on structured arrays."""
    flags = (1 if full else 0) | (2 if exact else 0)
    lib().orc_match(_p(pts1), n1, _p(pts2), n2, flags)


def match_rows(pts1, row0, nrows, pts2, n2, full=False, exact=False):
    flags = (1 if full else 0) | (2 if exact else 0)
    lib().orc_match_rows(_p(pts1), row0, nrows, _p(pts2), n2, flags)


def match_argmax(a, b):
    a = _f32(a); b = _f32(b)
    score = np.zeros(a.shape[0], np.float32)
    index = np.zeros(a.shape[0], np.int32)
    lib().orc_match_argmax(_p(a), a.shape[0], _p(b), b.shape[0], _p(score), _p(index))
    return score, index


def find_homography(pts, npts, num_loops=1000, min_score=0.85, max_ambiguity=0.95, thresh=5.0):
    """FindHomography on a structured SiftPoint array; consumes libc rand() like the reference.
    Returns (H 3x3, inlier count, winning loop index)."""
    H = (C.c_float * 9)()
    nm = C.c_int(0)
    best = lib().orc_find_homography(_p(pts), npts, H, C.byref(nm), num_loops, min_score, max_ambiguity, thresh)
    return np.array(list(H), np.float32).reshape(3, 3), nm.value, best


def improve_homography(pts, npts, H, num_loops=5, min_score=0.0, max_ambiguity=0.80, thresh=3.0):
    """ImproveHomography (geomFuncs.cpp:6-72) on a structured SiftPoint array, in place (match_error is written).
    Returns (H 3x3 float32, numfit)."""
    h = np.ascontiguousarray(H, np.float32).reshape(9).copy()
    n = lib().orc_improve_homography(_p(pts), npts, h.ctypes.data_as(C.POINTER(C.c_float)), num_loops, min_score,
                                     max_ambiguity, thresh)
    return h.reshape(3, 3), n


def srand(seed):
    """Seed the process-wide libc rand() that FindHomography (oracle and HIP host side) draws from."""
    C.CDLL(None).srand(C.c_uint(seed))


class reference_cap:
    """Context manager: the reference's 32-extrema-per-block cap of FindPointsMultiNew (cudaSiftD.cu:1369-1377) on / off
    (off = the default: every extremum is kept, a documented deviation)."""

    def __init__(self, on):
        self.on = int(on)

    def __enter__(self):
        lib().orc_set_reference_cap(self.on)
        return self

    def __exit__(self, *exc):
        lib().orc_set_reference_cap(0)
        return False


def stats():
    s = OrcStats()
    lib().orc_stats_get(C.byref(s))
    return {k: getattr(s, k) for k, _ in OrcStats._fields_}


def stats_reset():
    lib().orc_stats_reset()


# ---------------------------------------------------------------- oracle/_ref (reference's own CPU matcher)
def ref_lib(npts):
    path = os.path.join(HERE, "_ref", "libmatchref_%d.so" % npts)
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ref_generate.argtypes = [vp, vp, C.c_uint]
    L.ref_match_c1.argtypes = [vp, vp, vp, vp]
    L.ref_match_c3.argtypes = [vp, vp, vp, vp]
    L.ref_npts.restype = C.c_int
    assert L.ref_npts() == npts
    return L


def ref_improve_homography(pts, npts, H, num_loops=5, min_score=0.0, max_ambiguity=0.80, thresh=3.0):
    """The reference's own ImproveHomography (geomFuncs.cpp, built by build_ref.sh into oracle/_ref/libgeomref.so).
    Returns (H, numfit) or None where the reference tree was absent at build time."""
    path = os.path.join(HERE, "_ref", "libgeomref.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_improve_homography.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, C.c_float]
    L.ref_improve_homography.restype = C.c_int
    h = np.ascontiguousarray(H, np.float32).reshape(9).copy()
    n = L.ref_improve_homography(_p(pts), npts, h.ctypes.data_as(C.POINTER(C.c_float)), num_loops, min_score,
                                 max_ambiguity, thresh)
    return h.reshape(3, 3), n


def aligned_f32(n, align=32):
    """float32 array whose data pointer is `align`-byte aligned (MatchC3 uses _mm256_load_ps)."""
    raw = np.zeros(n + align, np.float32)
    off = (-raw.ctypes.data % align) // 4
    return raw[off:off + n]

"""ctypes binding of oracle/_ref/libcudasift_refemul_{off,fast}.so — TEST INFRASTRUCTURE ONLY.

That library is the reference's OWN cudaImage.cu + cudaSiftH.cu (+ cudaSiftD.cu) + matching.cu, compiled from
/root/reference by oracle/build_ref.sh against the CPU SIMT emulation of oracle/simt_emul.{h,cpp}: every kernel
of the reference executes thread by thread on the CPU with its own tilings, shared memory, shuffles, atomics
and counter protocol.  It is what pins oracle/sift_oracle.c (tests/test_refemul_cpu.py).  It exists only where
/root/reference was present at build time (this container); the GPU box receives the prebuilt file.
"""
import ctypes as C
import os

import numpy as np

from .pyoracle import POINT_DTYPE, _f32, _p

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def lib(flavour="off"):
    """flavour: "off" = -ffp-contract=off (every product rounded), "fast" = g++ may fuse multiply-adds."""
    if flavour not in _libs:
        path = os.path.join(HERE, "_ref", "libcudasift_refemul_%s.so" % flavour)
        if not os.path.exists(path):
            _libs[flavour] = None
            return None
        from oracle.pyoracle import cpu_budget        # team sizes for the CPUs this process may use (cgroup quota)
        os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        os.environ.setdefault("SIMT_THREADS", str(min(cpu_budget(), 32)))
        L = C.CDLL(path)
        vp = C.c_void_p
        L.refemul_sizeof_siftpoint.restype = C.c_int
        assert L.refemul_sizeof_siftpoint() == 576
        L.refemul_extract.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.c_int,
                                      vp, C.POINTER(C.c_uint)]
        L.refemul_extract.restype = C.c_int
        L.refemul_lowpass.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp]
        L.refemul_scaledown.argtypes = [vp, C.c_int, C.c_int, vp]
        L.refemul_laplace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.refemul_laplace_taps.argtypes = [C.c_int, vp]
        L.refemul_match.argtypes = [vp, C.c_int, vp, C.c_int]
        L.refemul_match.restype = C.c_double
        L.refemul_find_homography.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.c_float,
                                              C.c_float, C.c_float, C.c_uint]
        L.refemul_find_homography.restype = C.c_double
        _libs[flavour] = L
    return _libs[flavour]


def available(flavour="off"):
    return lib(flavour) is not None


def extract(img, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, scale_up=False, max_pts=32768,
            flavour="off"):
    """The reference's ExtractSift.  Returns (points[max_pts], numPts, counters[17]); records in [numPts, counters[2*noct+1])
    are the finest octave's second orientations (they exist on the device, past numPts — cudaSiftH.cu:115)."""
    img = _f32(img)
    h, w = img.shape
    pts = np.zeros(max_pts, POINT_DTYPE)
    cnt = (C.c_uint * 17)()
    n = lib(flavour).refemul_extract(_p(img), w, h, num_octaves, init_blur, thresh, lowest_scale, int(scale_up), max_pts,
                                     _p(pts), cnt)
    return pts, n, np.array(list(cnt), dtype=np.uint32)


def lowpass(img, sigma, flavour="off"):
    img = _f32(img)
    out = np.empty_like(img)
    lib(flavour).refemul_lowpass(_p(img), img.shape[1], img.shape[0], sigma, _p(out))
    return out


def scaledown(img, flavour="off"):
    img = _f32(img)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib(flavour).refemul_scaledown(_p(img), w, h, _p(out))
    return out


def laplace(base, num_octaves, octave, flavour="off"):
    base = _f32(base)
    h, w = base.shape
    out = np.empty((7, h, w), np.float32)
    lib(flavour).refemul_laplace(_p(base), w, h, octave, num_octaves, _p(out))
    return out


def findpoints(dog, thresh, subsampling=1.0, lowest_scale=0.0, max_pts=32768, edge_limit=10.0, octave=1, flavour="off"):
    """The reference's FindPointsMulti on a stack of 7 DoG planes [7, h, w]: (records, detection counter — not clamped)."""
    dog = _f32(dog)
    assert dog.ndim == 3 and dog.shape[0] == 7
    _, h, w = dog.shape
    pts = np.zeros(max_pts, POINT_DTYPE)
    L = lib(flavour)
    L.refemul_findpoints.restype = C.c_int
    n = L.refemul_findpoints(_p(dog), w, h, C.c_float(thresh), C.c_float(edge_limit), C.c_float(1.0 / 5),
                             C.c_float(lowest_scale), C.c_float(subsampling), octave, max_pts, _p(pts))
    return pts, int(n)


def laplace_taps(num_octaves, flavour="off"):
    k = np.zeros(8 * 12 * 16, np.float32)
    lib(flavour).refemul_laplace_taps(num_octaves, _p(k))
    return k


def match(pts1, n1, pts2, n2, flavour="off"):
    """The reference's MatchSiftData, in place on pts1 (score, ambiguity, match, match_xpos, match_ypos)."""
    lib(flavour).refemul_match(_p(pts1), n1, _p(pts2), n2)


def find_homography(pts, n, num_loops=1000, min_score=0.85, max_ambiguity=0.95, thresh=5.0, seed=1, flavour="off"):
    H = (C.c_float * 9)()
    nm = C.c_int(0)
    lib(flavour).refemul_find_homography(_p(pts), n, H, C.byref(nm), num_loops, min_score, max_ambiguity, thresh, seed)
    return np.array(list(H), np.float32).reshape(3, 3), nm.value

"""ctypes binding of the C-ABI in include/misift.h (libmisift.so).

Thin by design: every method is one C call plus argument marshalling.  There is NO
CPU fallback — if the HIP library is missing or fails to load, importing the
binding's `lib()` raises, and every GPU entry point fails loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MISIFT_LIB: developer override to A/B-test another build of the same library (tools/variants.sh)
LIB_PATH = os.environ.get("MISIFT_LIB") or os.path.join(HERE, "libmisift.so")

POINT_DTYPE = np.dtype([
    ("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("sharpness", "<f4"),
    ("edgeness", "<f4"), ("orientation", "<f4"), ("score", "<f4"), ("ambiguity", "<f4"),
    ("match", "<i4"), ("match_xpos", "<f4"), ("match_ypos", "<f4"), ("match_error", "<f4"),
    ("subsampling", "<f4"), ("empty", "<f4", (3,)), ("data", "<f4", (128,)),
])
assert POINT_DTYPE.itemsize == 576


class Options(C.Structure):
    _fields_ = [("texfrac_bits", C.c_int), ("fix_numpts", C.c_int), ("match_full", C.c_int),
                ("match_exact_top2", C.c_int), ("quiet", C.c_int), ("fused", C.c_int), ("deterministic", C.c_int),
                ("reference_cap", C.c_int)]


class MisiftError(RuntimeError):
    pass


# name -> (restype, argtypes); also the list the symbol-export test checks against the header
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ip, _fp, _up = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint)
SIGNATURES = {
    "misift_device_count": (_i, []),
    "misift_device_info": (_i, [_i, C.c_char_p, _i, _ip, _ip, C.POINTER(_sz), _ip, _ip]),
    "misift_device_arch": (_i, [_i, C.c_char_p, _i]),
    "misift_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "misift_ctx_destroy": (None, [_vp]),
    "misift_ctx_set_stream": (_i, [_vp, _vp]),
    "misift_ctx_set_graph_replay": (_i, [_vp, _i]),
    "misift_ctx_sync": (_i, [_vp]),
    "misift_ctx_set_early_return": (_i, [_vp, _i]),
    "misift_ctx_chain_fallbacks": (_i, [_vp]),
    "misift_ctx_fuse_fallbacks": (_i, [_vp]),
    "misift_ctx_last_call_balanced": (_i, [_vp]),
    "misift_ctx_descr_big_fallbacks": (_i, [_vp]),
    "misift_last_error": (C.c_char_p, []),
    "misift_default_options": (None, [C.POINTER(Options)]),
    "misift_default_options_sized": (None, [C.POINTER(Options), C.c_size_t]),
    "misift_set_options": (_i, [_vp, C.POINTER(Options)]),
    "misift_get_options": (_i, [_vp, C.POINTER(Options)]),
    "misift_set_options_sized": (_i, [_vp, C.POINTER(Options), C.c_size_t]),
    "misift_get_options_sized": (_i, [_vp, C.POINTER(Options), C.c_size_t]),
    "misift_malloc": (_i, [_sz, C.POINTER(_vp)]),
    "misift_free": (_i, [_vp]),
    "misift_memset": (_i, [_vp, _vp, _i, _sz]),
    "misift_copy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "misift_copy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "misift_image_alloc": (_i, [_i, _i, C.POINTER(_vp), _ip]),
    "misift_upload_2d": (_i, [_vp, _vp, _i, _vp, _i, _i, _i]),
    "misift_download_2d": (_i, [_vp, _vp, _i, _vp, _i, _i, _i]),
    "misift_download_fields": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "misift_scratch_floats": (_sz, [_i, _i, _i, _i]),
    "misift_extract": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _i, _vp, _vp, _i, _ip]),
    "misift_extract_batch": (_i, [_vp, _vp, _i, _sz, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _ip]),
    "misift_extract_batch_async": (_i, [_vp, _vp, _i, _sz, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _vp]),
    "misift_get_counters": (_i, [_vp, _i, _up]),
    "misift_get_counter_block": (_i, [_vp, _i, _up]),
    "misift_set_counters": (_i, [_vp, _i, _up]),
    "misift_lowpass": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _f]),
    "misift_scaledown": (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    "misift_scaleup": (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    "misift_laplace_taps": (_i, [_i, _fp]),
    "misift_laplace": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "misift_reset_counters": (_i, [_vp, _i]),
    "misift_findpoints": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _f, _f, _i, _vp, _i]),
    "misift_dog_findpoints": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _i]),
    "misift_orientations": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "misift_descriptors": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _i]),
    "misift_rescale_positions": (_i, [_vp, _vp, _i, _f]),
    "misift_match": (_i, [_vp, _vp, _i, _vp, _i]),
    "misift_match_rows": (_i, [_vp, _vp, _i, _i, _vp, _i]),
    "misift_improve_homography": (_i, [_vp, _vp, _i, _fp, _i, _f, _f, _f, _ip]),
    "misift_malloc_managed": (_i, [_sz, C.POINTER(_vp)]),
    "misift_ctx_set_batches_in_flight": (_i, [_vp, _i]),
    "misift_ctx_get_batches_in_flight": (_i, [_vp]),
    "misift_ctx_wait_batch": (_i, [_vp, _vp]),
    "misift_ctx_record_batch": (_i, [_vp, _vp]),
    "misift_test_elementary": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i]),
    "misift_test_match_split": (_i, [_vp, _vp, _i, _vp, _i, _i, _i]),
    "misift_test_match_plan": (_i, [_i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "misift_test_frame_shares": (_i, [_i, _i, C.c_void_p, C.c_void_p]),
    "misift_test_set_knob": (_i, [_vp, C.c_char_p, C.c_double]),
    "misift_test_knob_names": (C.c_char_p, []),
    "misift_test_set_guard": (_i, [_i]),
    "misift_test_check_guards": (_i, [C.POINTER(_i)]),
    "misift_comm_unique_id": (_i, [_vp]),
    "misift_comm_create": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "misift_comm_adopt": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "misift_loopback_world_create": (_i, [_i, C.POINTER(_vp)]),
    "misift_loopback_world_destroy": (None, [_vp]),
    "misift_comm_create_loopback": (_i, [_vp, _vp, _i, C.POINTER(_vp)]),
    "misift_gather_test": (_i, [_vp, _i, _ip]),
    "misift_comm_destroy": (None, [_vp]),
    "misift_comm_rank": (_i, [_vp]),
    "misift_comm_size": (_i, [_vp]),
    "misift_comm_barrier": (_i, [_vp]),
    "misift_gather_post": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "misift_gather_complete": (_i, [_vp, _i, _i, _vp, _vp, _sz, _vp]),
    "misift_match_sharded": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "misift_comm_wire_bytes": (_i, [_vp, _vp, _vp]),
    "misift_comm_create_host": (_i, [_i, _i, _vp, _vp]),
    "misift_find_homography": (_i, [_vp, _vp, _i, _fp, _ip, _i, _f, _f, _f]),
    "misift_extract_batch_packed_async": (_i, [_vp, _vp, _i, _sz, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "misift_lowpass_scaledown": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _f, _vp, _i]),
    "misift_extract_batch_u8": (_i, [_vp, _vp, _i, _sz, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _ip]),
    "misift_extract_batch_ex": (_i, [_vp, _vp, _i, _i, _sz, _i, _i, _i, _i, _f, _f, _f, _i, _vp, _vp, _i, _ip]),
    "misift_pipe_create": (_i, [_vp, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, C.POINTER(_vp)]),
    "misift_pipe_destroy": (None, [_vp]),
    "misift_pipe_submit": (_i, [_vp, _vp, _i]),
    "misift_pipe_collect": (_i, [_vp, _ip, _ip, _vp, _sz, C.POINTER(_sz)]),
    "misift_pipe_pending": (_i, [_vp]),
    "misift_host_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "misift_host_free": (_i, [_vp]),
    "misift_timer_start": (_i, [_vp]),
    "misift_timer_stop_ms": (_i, [_vp, _fp]),
    "misift_profile_enable": (_i, [_vp, _i]),
    "misift_profile_reset": (_i, [_vp]),
    "misift_profile_read": (_i, [_vp, _i, _vp, _fp, _ip, _ip]),
}

_lib = None


def lib():
    """Load libmisift.so (raises if it is missing: the HIP path is the only path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MisiftError("HIP extension %s is not built — run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` or `make`" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise MisiftError("%s failed (rc=%d): %s" % (what, rc, lib().misift_last_error().decode()))


def device_count():
    return lib().misift_device_count()


def scratch_floats(width, height, num_octaves=5, scale_up=False):
    return int(lib().misift_scratch_floats(width, height, num_octaves, int(scale_up)))


def laplace_taps(num_octaves):
    k = np.zeros(8 * 12 * 16, np.float32)
    check(lib().misift_laplace_taps(num_octaves, k.ctypes.data_as(_fp)), "misift_laplace_taps")
    return k


def knob_names():
    """{knob: environment variable honoured under MISIFT_TUNABLES=1}."""
    return dict(kv.split("=") for kv in lib().misift_test_knob_names().decode().split(","))


def set_guard(on=True):
    """Test mode: every device allocation made from now on (DevBuf and the library's own buffers) gets 64 KiB guard bands
    and a NaN-poisoned payload (misift_test_set_guard).  Returns the previous mode."""
    return bool(lib().misift_test_set_guard(int(on)))


def check_guards():
    """Verify the guard bands of every live guarded allocation; raises MisiftError on damage, returns how many were checked."""
    n = _i(0)
    bad = lib().misift_test_check_guards(C.byref(n))
    if bad != 0:
        raise MisiftError("guard check: %d damaged allocation(s): %s" % (bad, lib().misift_last_error().decode()))
    return n.value


class DevBuf:
    """A raw HBM allocation owned through misift_malloc/misift_free."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().misift_malloc(nbytes, C.byref(p)), "misift_malloc")
        self.ptr = p.value
        self.nbytes = nbytes

    def free(self):
        if self.ptr:
            lib().misift_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per device (+ optional hipStream_t given as an int)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(lib().misift_ctx_create(device, stream, C.byref(h)), "misift_ctx_create")
        self.h = h.value
        self.device = device

    def close(self):
        if self.h:
            lib().misift_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- options
    def get_options(self):
        o = Options()
        check(lib().misift_get_options_sized(self.h, C.byref(o), C.sizeof(o)), "misift_get_options")
        return o

    def set_options(self, **kw):
        o = self.get_options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise KeyError(k)
            setattr(o, k, int(v))
        check(lib().misift_set_options_sized(self.h, C.byref(o), C.sizeof(o)), "misift_set_options")

    def set_batches_in_flight(self, k):
        """K pipelines behind this context (misift_ctx_set_batches_in_flight): consecutive packed-async calls overlap."""
        check(lib().misift_ctx_set_batches_in_flight(self.h, k), "misift_ctx_set_batches_in_flight")

    def record_batch(self, event):
        """Record a HipEvent behind the most recently enqueued batch of this context."""
        check(lib().misift_ctx_record_batch(self.h, event.h), "misift_ctx_record_batch")

    def wait_batch(self, stream):
        """Make `stream` (a raw hipStream_t value) wait for the most recently enqueued batch of this context."""
        check(lib().misift_ctx_wait_batch(self.h, stream), "misift_ctx_wait_batch")

    def set_knob(self, name, value):
        """Developer / test knob of this context (misift_test_set_knob; capi.knob_names() lists them)."""
        check(lib().misift_test_set_knob(self.h, name.encode(), float(value)), "misift_test_set_knob")

    def set_early_return(self, on=True):
        """Synchronous calls return at the last kernel's completion flag instead of after a stream synchronisation."""
        check(lib().misift_ctx_set_early_return(self.h, int(on)), "misift_ctx_set_early_return")

    def last_call_balanced(self):
        return lib().misift_ctx_last_call_balanced(self.h)

    def descr_big_fallbacks(self):
        return lib().misift_ctx_descr_big_fallbacks(self.h)

    def chain_fallbacks(self):
        return lib().misift_ctx_chain_fallbacks(self.h)

    def fuse_fallbacks(self):
        return lib().misift_ctx_fuse_fallbacks(self.h)

    def sync(self):
        check(lib().misift_ctx_sync(self.h), "misift_ctx_sync")

    # ---- memory helpers
    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = DevBuf(max(arr.nbytes, 16))
        if arr.nbytes:
            check(lib().misift_copy_h2d(self.h, buf.ptr, arr.ctypes.data, arr.nbytes), "misift_copy_h2d")
        return buf

    def download(self, buf, shape, dtype):
        out = np.empty(shape, dtype)
        if out.nbytes:
            check(lib().misift_copy_d2h(self.h, out.ctypes.data, buf.ptr if isinstance(buf, DevBuf) else buf,
                                        out.nbytes), "misift_copy_d2h")
        return out

    def upload_image(self, img, pitch=None):
        """Host [h,w] float32 -> pitched device image.  Returns (DevBuf, pitch_floats)."""
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        if pitch is None:
            pitch = (w + 127) // 128 * 128
        buf = DevBuf(4 * pitch * h)
        check(lib().misift_upload_2d(self.h, buf.ptr, pitch, img.ctypes.data, w, w, h), "misift_upload_2d")
        return buf, pitch

    def download_image(self, ptr, w, h, pitch):
        out = np.empty((h, w), np.float32)
        check(lib().misift_download_2d(self.h, out.ctypes.data, w, ptr, pitch, w, h), "misift_download_2d")
        return out

    poison_outputs = False        # tests (guard mode): output buffers start as 0xFF instead of zero

    def zeros(self, nbytes):
        buf = DevBuf(nbytes)
        check(lib().misift_memset(self.h, buf.ptr, 0xFF if self.poison_outputs else 0, nbytes), "misift_memset")
        self.sync()
        return buf

    # ---- counters
    def get_counters(self, frame=0):
        c = (C.c_uint * 17)()
        check(lib().misift_get_counters(self.h, frame, c), "misift_get_counters")
        return np.array(list(c), np.uint32)

    def get_counter_block(self, frame=0):
        """All 64 words of a frame's counter block (diagnostic: candidate / detection / duplicate counts per octave)."""
        c = (C.c_uint * 64)()
        check(lib().misift_get_counter_block(self.h, frame, c), "misift_get_counter_block")
        return np.array(list(c), np.uint32)

    def set_counters(self, counters, frame=0):
        c = (C.c_uint * 17)(*[int(v) for v in counters])
        check(lib().misift_set_counters(self.h, frame, c), "misift_set_counters")

    # ---- stage-level calls on host arrays (upload, run, download) — used by the parity tests
    def lowpass(self, img, sigma):
        h, w = img.shape
        src, p = self.upload_image(img)
        dst = DevBuf(4 * p * h)
        check(lib().misift_lowpass(self.h, src.ptr, w, h, p, dst.ptr, p, sigma), "misift_lowpass")
        return self.download_image(dst.ptr, w, h, p)

    def scaledown(self, img):
        h, w = img.shape
        src, p = self.upload_image(img)
        w2, h2 = w // 2, h // 2
        p2 = (w2 + 127) // 128 * 128
        dst = DevBuf(4 * p2 * max(h2, 1))
        check(lib().misift_scaledown(self.h, src.ptr, w, h, p, dst.ptr, p2), "misift_scaledown")
        return self.download_image(dst.ptr, w2, h2, p2)

    def lowpass_scaledown(self, img, sigma):
        """Fused LowPass + first ScaleDown.  Returns (lowpassed [h,w], decimated [h//2,w//2])."""
        h, w = img.shape
        src, p = self.upload_image(img)
        dst = DevBuf(4 * p * h)
        w2, h2 = w // 2, h // 2
        p2 = (w2 + 127) // 128 * 128
        dst2 = DevBuf(4 * p2 * max(h2, 1))
        check(lib().misift_lowpass_scaledown(self.h, src.ptr, w, h, p, dst.ptr, p, sigma, dst2.ptr, p2),
              "misift_lowpass_scaledown")
        return self.download_image(dst.ptr, w, h, p), self.download_image(dst2.ptr, w2, h2, p2)

    def scaleup(self, img):
        h, w = img.shape
        src, p = self.upload_image(img)
        p2 = (2 * w + 127) // 128 * 128
        dst = DevBuf(4 * p2 * 2 * h)
        check(lib().misift_scaleup(self.h, src.ptr, w, h, p, dst.ptr, p2), "misift_scaleup")
        return self.download_image(dst.ptr, 2 * w, 2 * h, p2)

    def laplace(self, base, num_octaves, octave):
        h, w = base.shape
        src, p = self.upload_image(base)
        dog = DevBuf(4 * 7 * p * h)
        check(lib().misift_laplace(self.h, src.ptr, w, h, p, num_octaves, octave, dog.ptr), "misift_laplace")
        planes = self.download(dog, (7, h, p), np.float32)
        return np.ascontiguousarray(planes[:, :, :w])

    def findpoints(self, dog, thresh, subsampling=1.0, lowest_scale=0.0, max_pts=32768, octave=1,
                   edge_limit=10.0):
        """Unfused detect+refine on host DoG planes [7,h,w]. Returns (points, n)."""
        _, h, w = dog.shape
        p = (w + 127) // 128 * 128
        padded = np.zeros((7, h, p), np.float32)
        padded[:, :, :w] = dog
        d = self.upload(padded)
        pts = self.zeros(576 * max_pts)
        check(lib().misift_reset_counters(self.h, max_pts), "misift_reset_counters")
        check(lib().misift_findpoints(self.h, d.ptr, w, h, p, thresh, edge_limit, lowest_scale, subsampling,
                                      octave, pts.ptr, max_pts), "misift_findpoints")
        cnt = self.get_counters()
        n = int(min(cnt[2 * octave], max_pts))
        return self.download(pts, (max_pts,), POINT_DTYPE), n

    def dog_findpoints(self, base, num_octaves, octave, thresh, subsampling=1.0, lowest_scale=0.0,
                       max_pts=32768, edge_limit=10.0):
        """Fused DoG+detect+refine on a host base image. Returns (points, n)."""
        h, w = base.shape
        src, p = self.upload_image(base)
        pts = self.zeros(576 * max_pts)
        check(lib().misift_reset_counters(self.h, max_pts), "misift_reset_counters")
        check(lib().misift_dog_findpoints(self.h, src.ptr, w, h, p, num_octaves, octave, thresh, edge_limit,
                                          lowest_scale, subsampling, pts.ptr, max_pts), "misift_dog_findpoints")
        cnt = self.get_counters()
        n = int(min(cnt[2 * octave], max_pts))
        return self.download(pts, (max_pts,), POINT_DTYPE), n

    def orient_and_describe(self, base, pts, first, last, octave, subsampling=1.0, max_pts=None):
        """Run orientation + descriptor kernels on host points [first,last) of one octave.
        Returns (points, counters) with duplicates appended from `last`."""
        h, w = base.shape
        if max_pts is None:
            max_pts = len(pts)
        src, p = self.upload_image(base)
        d = self.upload(pts)
        cnt = np.zeros(17, np.uint32)
        cnt[2 * octave - 1] = first
        cnt[2 * octave] = last
        cnt[2 * octave + 1] = last
        check(lib().misift_reset_counters(self.h, max_pts), "misift_reset_counters")
        self.set_counters(cnt)
        check(lib().misift_orientations(self.h, src.ptr, w, h, p, octave, d.ptr, max_pts), "misift_orientations")
        check(lib().misift_descriptors(self.h, src.ptr, w, h, p, subsampling, octave, d.ptr, max_pts),
              "misift_descriptors")
        return self.download(d, (len(pts),), POINT_DTYPE), self.get_counters()

    # ---- whole path
    def extract(self, img, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, scale_up=False,
                max_pts=32768, scratch=True):
        """ExtractSift on a host image.  Returns (points[max_pts], numPts, counters[17])."""
        h, w = img.shape
        src, p = self.upload_image(img)
        sc = DevBuf(4 * scratch_floats(w, h, num_octaves, scale_up)) if scratch else None
        pts = self.zeros(576 * max_pts)
        n = C.c_int(0)
        check(lib().misift_extract(self.h, src.ptr, w, h, p, num_octaves, init_blur, thresh, lowest_scale,
                                   int(scale_up), sc.ptr if sc else None, pts.ptr, max_pts, C.byref(n)),
              "misift_extract")
        return self.download(pts, (max_pts,), POINT_DTYPE), n.value, self.get_counters()

    def extract_batch(self, imgs, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, max_pts=32768):
        """imgs: [B,h,w] host array.  Returns (points[B,max_pts], numPts[B])."""
        imgs = np.ascontiguousarray(imgs, np.float32)
        B, h, w = imgs.shape
        p = (w + 127) // 128 * 128
        padded = np.zeros((B, h, p), np.float32)
        padded[:, :, :w] = imgs
        d = self.upload(padded)
        S = scratch_floats(w, h, num_octaves, False)
        sc = DevBuf(4 * S * B)
        pts = self.zeros(576 * max_pts * B)
        n = (C.c_int * B)()
        check(lib().misift_extract_batch(self.h, d.ptr, B, h * p, w, h, p, num_octaves, init_blur, thresh,
                                         lowest_scale, sc.ptr, pts.ptr, max_pts, n), "misift_extract_batch")
        return self.download(pts, (B, max_pts), POINT_DTYPE), np.array(list(n), np.int32)

    def extract_batch_ex(self, imgs, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, scale_up=False,
                         max_pts=32768):
        """imgs: [B,h,w] uint8 or float32 host array (tightly packed).  Returns (points[B,max_pts], numPts[B])."""
        u8 = imgs.dtype == np.uint8
        imgs = np.ascontiguousarray(imgs, np.uint8 if u8 else np.float32)
        B, h, w = imgs.shape
        d = self.upload(imgs)
        sc = DevBuf(4 * scratch_floats(w, h, num_octaves, scale_up) * B)
        pts = self.zeros(576 * max_pts * B)
        n = (C.c_int * B)()
        check(lib().misift_extract_batch_ex(self.h, d.ptr, int(u8), B, h * w, w, h, w, num_octaves, init_blur, thresh,
                                            lowest_scale, int(scale_up), sc.ptr, pts.ptr, max_pts, n),
              "misift_extract_batch_ex")
        return self.download(pts, (B, max_pts), POINT_DTYPE), np.array(list(n), np.int32)

    def extract_batch_u8(self, imgs, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, max_pts=32768):
        """imgs: [B,h,w] uint8 host array (tightly packed).  Returns (points[B,max_pts], numPts[B])."""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        B, h, w = imgs.shape
        d = self.upload(imgs)
        S = scratch_floats(w, h, num_octaves, False)
        sc = DevBuf(4 * S * B)
        pts = self.zeros(576 * max_pts * B)
        n = (C.c_int * B)()
        check(lib().misift_extract_batch_u8(self.h, d.ptr, B, h * w, w, h, w, num_octaves, init_blur, thresh,
                                            lowest_scale, sc.ptr, pts.ptr, max_pts, n), "misift_extract_batch_u8")
        return self.download(pts, (B, max_pts), POINT_DTYPE), np.array(list(n), np.int32)

    def match(self, pts1, n1, pts2, n2, row_begin=0, row_count=None):
        """MatchSiftData on host structured arrays; returns the updated copy of pts1."""
        d1 = self.upload(pts1)
        d2 = self.upload(pts2)
        if row_count is None:
            check(lib().misift_match(self.h, d1.ptr, n1, d2.ptr, n2), "misift_match")
        else:
            check(lib().misift_match_rows(self.h, d1.ptr, row_begin, row_count, d2.ptr, n2), "misift_match_rows")
        return self.download(d1, (len(pts1),), POINT_DTYPE)

    def match_split(self, pts1, n1, pts2, n2, own_tile_begin, own_tile_end):
        """Test hook: misift_match with the column sweep cut into two launches (the sharded matcher's cut)."""
        d1 = self.upload(pts1)
        d2 = self.upload(pts2)
        check(lib().misift_test_match_split(self.h, d1.ptr, n1, d2.ptr, n2, own_tile_begin, own_tile_end),
              "misift_test_match_split")
        return self.download(d1, (len(pts1),), POINT_DTYPE)

    def find_homography(self, dpts_ptr, npts, num_loops=1000, min_score=0.85, max_ambiguity=0.95, thresh=5.0):
        H = (C.c_float * 9)()
        nm = C.c_int(0)
        check(lib().misift_find_homography(self.h, dpts_ptr, npts, H, C.byref(nm), num_loops, min_score,
                                           max_ambiguity, thresh), "misift_find_homography")
        return np.array(list(H), np.float32).reshape(3, 3), nm.value

    def improve_homography(self, dpts_ptr, npts, H, num_loops=5, min_score=0.0, max_ambiguity=0.80, thresh=3.0):
        h = (C.c_float * 9)(*[float(v) for v in np.asarray(H, np.float32).reshape(9)])
        nf = C.c_int(0)
        check(lib().misift_improve_homography(self.h, dpts_ptr, npts, h, num_loops, min_score, max_ambiguity, thresh,
                                              C.byref(nf)), "misift_improve_homography")
        return np.array(list(h), np.float32).reshape(3, 3), nf.value

    def test_elementary(self, fn, x, y=None):
        """Device copies of det_exp2 (0) / det_atan2(y, x) (1) / det_exp (2) / det_sincos (3) on float32 arrays."""
        x = np.ascontiguousarray(x, np.float32)
        dx = self.upload(x)
        dy = self.upload(np.ascontiguousarray(y, np.float32)) if y is not None else None
        o1, o2 = self.zeros(4 * len(x)), self.zeros(4 * len(x))
        check(lib().misift_test_elementary(self.h, fn, dx.ptr, dy.ptr if dy else None, o1.ptr, o2.ptr, len(x)),
              "misift_test_elementary")
        a = self.download(o1, (len(x),), np.float32)
        return (a, self.download(o2, (len(x),), np.float32)) if fn == 3 else a

    # ---- profiling
    def profile_enable(self, on=True):
        check(lib().misift_profile_enable(self.h, int(on)), "misift_profile_enable")

    def profile_reset(self):
        check(lib().misift_profile_reset(self.h), "misift_profile_reset")

    def profile_read(self):
        cap = 32
        names = C.create_string_buffer(32 * cap)
        ms = (C.c_float * cap)()
        calls = (C.c_int * cap)()
        n = C.c_int(0)
        check(lib().misift_profile_read(self.h, cap, C.cast(names, C.c_void_p), ms, calls, C.byref(n)),
              "misift_profile_read")
        out = {}
        for i in range(n.value):
            nm = names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode()
            out[nm] = {"total_ms": float(ms[i]), "calls": int(calls[i])}
        return out


class HipEvent:
    """A raw hipEvent_t (timing enabled) for the entry points that take one as void*: misift_ctx_record_batch."""
    _hip = None

    def __init__(self):
        if HipEvent._hip is None:
            HipEvent._hip = C.CDLL("libamdhip64.so")
            HipEvent._hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            HipEvent._hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
            HipEvent._hip.hipEventSynchronize.argtypes = [C.c_void_p]
            HipEvent._hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.h = C.c_void_p()
        if HipEvent._hip.hipEventCreate(C.byref(self.h)) != 0:
            raise MisiftError("hipEventCreate failed")

    def synchronize(self):
        if HipEvent._hip.hipEventSynchronize(self.h) != 0:
            raise MisiftError("hipEventSynchronize failed")

    def stream_wait(self, stream):
        """`stream` (raw hipStream_t value) waits for this event."""
        if HipEvent._hip.hipStreamWaitEvent(C.c_void_p(stream), self.h, 0) != 0:
            raise MisiftError("hipStreamWaitEvent failed")

    def elapsed_time(self, later):
        ms = C.c_float(0)
        if HipEvent._hip.hipEventElapsedTime(C.byref(ms), self.h, later.h) != 0:
            raise MisiftError("hipEventElapsedTime failed")
        return ms.value

    def __del__(self):
        try:
            if self.h:
                HipEvent._hip.hipEventDestroy(self.h)
                self.h = None
        except Exception:
            pass


COMM_ID_BYTES = 128
RESULT_DTYPE = np.dtype([("score", "<f4"), ("ambiguity", "<f4"), ("match", "<i4")])      # 12 B/row (misift_match_sharded)
# the 528-byte match column misift_match_sharded ships and leaves in d_set2_all (MISIFT_MATCH_COLUMN_BYTES)
COLUMN_DTYPE = np.dtype([("data", "<f4", (128,)), ("xpos", "<f4"), ("ypos", "<f4"), ("reserved", "<f4", (2,))])
assert COLUMN_DTYPE.itemsize == 528


def comm_unique_id():
    """128 opaque bytes made by rank 0 (ncclGetUniqueId); ship them to the other ranks, then Comm(ctx, n, r, id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().misift_comm_unique_id(C.cast(buf, C.c_void_p)), "misift_comm_unique_id")
    return buf.raw


class LoopbackWorld:
    """misift_loopback_world: the rendezvous object of N in-process communicators (fake N ranks on one GPU)."""

    def __init__(self, nranks):
        h = C.c_void_p()
        check(lib().misift_loopback_world_create(nranks, C.byref(h)), "misift_loopback_world_create")
        self.h, self.size = h, nranks

    def close(self):
        if self.h:
            lib().misift_loopback_world_destroy(self.h)
            self.h = None


class Comm:
    """One communicator per context (misift_comm_*): the multi-GPU entry points of the C-ABI.
    id_bytes: the 128-byte RCCL id — or a LoopbackWorld for the in-process transport."""

    def __init__(self, ctx, nranks, rank, id_bytes):
        h = C.c_void_p()
        if isinstance(id_bytes, LoopbackWorld):
            assert nranks == id_bytes.size
            check(lib().misift_comm_create_loopback(ctx.h, id_bytes.h, rank, C.byref(h)), "misift_comm_create_loopback")
        else:
            assert len(id_bytes) == COMM_ID_BYTES
            buf = C.create_string_buffer(bytes(id_bytes), COMM_ID_BYTES)
            check(lib().misift_comm_create(ctx.h, nranks, rank, C.cast(buf, C.c_void_p), C.byref(h)), "misift_comm_create")
        self.h = h
        self.ctx = ctx
        self.rank, self.size = lib().misift_comm_rank(h), lib().misift_comm_size(h)

    def barrier(self):
        check(lib().misift_comm_barrier(self.h), "misift_comm_barrier")

    def gather_post(self, slot, d_counts, nframes, d_packed, ctx=None):
        """ctx: the context (of the communicator's device) whose stream produced the buffers; default = the communicator's own."""
        check(lib().misift_gather_post((ctx or self.ctx).h, self.h, slot, d_counts, nframes, d_packed), "misift_gather_post")

    def gather_test(self, slot):
        """True once the batch posted in `slot` has finished on the GPU (non-blocking)."""
        r = C.c_int(0)
        check(lib().misift_gather_test(self.h, slot, C.byref(r)), "misift_gather_test")
        return bool(r.value)

    def gather_complete(self, slot, nframes, root=0, d_recv=None, capacity_records=0):
        """Returns (all_counts [size, nframes] int32, rank offsets [size+1] in records)."""
        counts = np.zeros((self.size, nframes), np.int32)
        offs = (C.c_size_t * (self.size + 1))()
        check(lib().misift_gather_complete(self.h, slot, root, counts.ctypes.data, d_recv, capacity_records, offs),
              "misift_gather_complete")
        return counts, np.array(list(offs), np.int64)

    def wire_bytes(self):
        """(received, sent) payload bytes of this rank since the communicator was created."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib().misift_comm_wire_bytes(self.h, C.byref(a), C.byref(b)), "misift_comm_wire_bytes")
        return int(a.value), int(b.value)

    def match_sharded(self, d_rows1, row_count, d_shard2, shard_count, d_set2_all, d_results_all=None):
        check(lib().misift_match_sharded(self.ctx.h, self.h, d_rows1, row_count, d_shard2, shard_count, d_set2_all,
                                         d_results_all), "misift_match_sharded")

    def close(self):
        if self.h:
            lib().misift_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _HostTransport(C.Structure):
    """misift_host_transport (include/misift.h)."""
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    P2P = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
    END = C.CFUNCTYPE(C.c_int, C.c_void_p)
    _fields_ = [("user", C.c_void_p), ("allgather", AG), ("send", P2P), ("recv", P2P), ("group_end", END)]


class HostComm(Comm):
    """A HOST communicator (misift_comm_create_host): no device; counts / packed records / receive buffers are host
    memory (numpy arrays), the exchange primitives are Python callables — tests/test_dist_cpu.py implements them with
    torch.distributed (gloo), so the gather logic of multigpu.hip itself runs on a machine without GPUs.

    allgather(send_ptr, recv_ptr, nbytes), send(ptr, nbytes, peer), recv(ptr, nbytes, peer), group_end(): raise on error."""

    def __init__(self, nranks, rank, allgather, send, recv, group_end):     # noqa: super().__init__ not called on purpose
        def guard(fn):
            def call(_user, *a):
                try:
                    fn(*a)
                    return 0
                except Exception:                    # noqa: BLE001 — reported through the C-ABI's error path
                    import traceback
                    traceback.print_exc()
                    return 1
            return call
        self._cb = _HostTransport(None, _HostTransport.AG(guard(allgather)), _HostTransport.P2P(guard(send)),
                                  _HostTransport.P2P(guard(recv)), _HostTransport.END(guard(group_end)))
        h = C.c_void_p()
        check(lib().misift_comm_create_host(nranks, rank, C.byref(self._cb), C.byref(h)), "misift_comm_create_host")
        self.h = h
        self.ctx = None
        self.rank, self.size = lib().misift_comm_rank(h), lib().misift_comm_size(h)

    def gather_post(self, slot, counts, nframes, packed, ctx=None):
        """counts: int32 numpy [nframes]; packed: uint8 numpy (both must stay alive until gather_complete)."""
        check(lib().misift_gather_post(None, self.h, slot, counts.ctypes.data, nframes, packed.ctypes.data), "misift_gather_post")

    def gather_complete(self, slot, nframes, root=0, recv=None, capacity_records=0):
        counts = np.zeros((self.size, nframes), np.int32)
        offs = (C.c_size_t * (self.size + 1))()
        check(lib().misift_gather_complete(self.h, slot, root, counts.ctypes.data, recv.ctypes.data if recv is not None else None,
                                           capacity_records, offs), "misift_gather_complete")
        return counts, np.array(list(offs), np.int64)


class PinnedArray:
    """numpy view over pinned host memory (misift_host_alloc) — uploads/downloads from it are asynchronous."""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().misift_host_alloc(max(nbytes, 1), C.byref(p)), "misift_host_alloc")
        self.ptr = p.value
        buf = (C.c_char * nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().misift_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pipe:
    """Host-fed extraction pipeline (misift_pipe_*): submit batches of host frames, collect packed records."""

    def __init__(self, ctx, width, height, batch_frames, src_u8=True, num_octaves=5, init_blur=1.0, thresh=3.0,
                 lowest_scale=0.0, max_pts=32768, depth=2):
        h = C.c_void_p()
        check(lib().misift_pipe_create(ctx.h, width, height, batch_frames, int(bool(src_u8)), num_octaves, init_blur,
                                       thresh, lowest_scale, max_pts, depth, C.byref(h)), "misift_pipe_create")
        self.h = h
        self.ctx = ctx
        self.batch = batch_frames
        self.max_pts = max_pts

    def submit(self, host_ptr, nframes):
        check(lib().misift_pipe_submit(self.h, host_ptr, nframes), "misift_pipe_submit")

    def collect(self, out_ptr=None, capacity_records=0):
        """Returns (counts[nframes], nrecords).  Records go to out_ptr (capacity in records) when given."""
        nf = C.c_int(0)
        cnt = (C.c_int * self.batch)()
        nrec = C.c_size_t(0)
        check(lib().misift_pipe_collect(self.h, C.byref(nf), cnt, out_ptr, capacity_records, C.byref(nrec)),
              "misift_pipe_collect")
        return np.array(list(cnt)[:nf.value], np.int32), int(nrec.value)

    def pending(self):
        return lib().misift_pipe_pending(self.h)

    def close(self):
        if self.h:
            lib().misift_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

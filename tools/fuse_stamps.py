#!/usr/bin/env python3
"""Developer aid (r06): where the fused orientation + descriptor launch of a single call spends its time.  Needs a library
built with -DFUSE_STAMPS=1 (tools/variants.sh kernels_points.hip fstamps "-DFUSE_STAMPS=1").
usage (GPU box): MISIFT_LIB=build/variants/libmisift_fstamps.so python tools/fuse_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cudasift_amd import capi  # noqa: E402
from synth import synth_frame  # noqa: E402

ctx = capi.Context(0)
ctx.set_options(quiet=1)
ctx.set_knob("fuse_orient", 1)
img = synth_frame(0, 1920, 1080)
lib = capi.lib()
stamps = lib.misift_debug_fuse_stamps
stamps.argtypes = [C.c_void_p]
stamps.restype = C.c_int
out = np.zeros(16, np.uint32)
for _ in range(10):
    ctx.extract(img, num_octaves=5, thresh=3.0)
stamps(None)
rows = []
for _ in range(30):
    ctx.extract(img, num_octaves=5, thresh=3.0)
    stamps(out.ctypes.data)
    w = out.astype(np.int64)
    rows.append([(w[k] - w[0]) / 100.0 for k in (1, 8, 6, 2, 9, 3, 4, 5)] + [w[7] / 100.0])
med = np.median(np.array(rows), axis=0)
names = ("last workgroup starts", "first workgroup has its orientations", "last COARSE-octave workgroup has its orientations",
         "last workgroup has its orientations", "first workgroup through the wait", "last workgroup through the wait",
         "last descriptors written", "counters exported", "(longest orientation pass of one wavefront, its own duration)")
print("fused orientation + descriptor launch, 1920x1080, fuse_fallbacks %d (us after the first workgroup started; medians of 30 calls)" % ctx.fuse_fallbacks())
for n, v in zip(names, med):
    print("   %-55s +%.2f" % (n, v))
ctx.close()

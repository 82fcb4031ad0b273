#!/usr/bin/env python3
"""Developer aid: where a small MatchSiftData spends its time.  Needs a library built with -DMT_STAMPS=1
(tools/variants.sh kernels_match.hip stamps "-DMT_STAMPS=1"; MISIFT_LIB=build/variants/libmisift_stamps.so).
usage (GPU box): MISIFT_LIB=... [MISIFT_MATCH_CHUNKS=c] python tools/match_stamps.py n1 n2"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cudasift_amd import capi  # noqa: E402
from synth import descriptors_to_points, synth_descriptors  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 2052
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 2163
ctx = capi.Context(0)
ctx.set_options(quiet=1)
a = ctx.upload(descriptors_to_points(synth_descriptors(n1, 12345), capi.POINT_DTYPE))
b = ctx.upload(descriptors_to_points(synth_descriptors(n2, 12346), capi.POINT_DTYPE))
lib = capi.lib()
stamps = lib.misift_debug_match_stamps
stamps.argtypes = [C.c_void_p]
stamps.restype = C.c_int
out = np.zeros(16, np.uint32)
for _ in range(50):
    capi.check(lib.misift_match(ctx.h, a.ptr, n1, b.ptr, n2), "misift_match")
stamps(None)
rows = []
for _ in range(40):
    capi.check(lib.misift_match(ctx.h, a.ptr, n1, b.ptr, n2), "misift_match")
    stamps(out.ctypes.data)
    w = out.astype(np.int64)
    rows.append([(w[k] - w[0]) / 100.0 for k in (1, 2, 3, 4, 5, 8, 9, 10, 11)])
med = np.median(np.array(rows), axis=0)
names = ("last wg starts", "last wg has its operands", "last first-tile staged", "last sweep done", "last partials stored",
         "merge: first wg starts", "merge: last wg starts", "merge: last rows written", "merge: flag stored")
print("match %d x %d  MISIFT_MATCH_CHUNKS=%s  (us after the first match workgroup started; medians of 40 calls)"
      % (n1, n2, os.environ.get("MISIFT_MATCH_CHUNKS", "-")))
print("   " + ", ".join("%s +%.2f" % (n, v) for n, v in zip(names, med)))
ctx.close()

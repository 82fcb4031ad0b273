#!/usr/bin/env python3
"""The 100 k x 100 k matcher (BASELINE config 5 on one GPU) alone, for rocprofv3: MATCH_REPS sweeps of misift_match.
usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python tools/match_prof.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cudasift_amd import capi  # noqa: E402
from synth import descriptors_to_points, synth_descriptors  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = n // 32 * 32
reps = int(os.environ.get("MATCH_REPS", "5"))
ctx = capi.Context(0)
ctx.set_options(quiet=1)
a = ctx.upload(descriptors_to_points(synth_descriptors(n, 12345), capi.POINT_DTYPE))
b = ctx.upload(descriptors_to_points(synth_descriptors(n, 12346), capi.POINT_DTYPE))
capi.check(capi.lib().misift_match(ctx.h, a.ptr, n, b.ptr, n), "misift_match")      # warm-up
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    capi.check(capi.lib().misift_match(ctx.h, a.ptr, n, b.ptr, n), "misift_match")
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print("match %d x %d: %.3f ms per sweep, %.1f Gpairs/s, %.1f TFLOP/s (%.3f of 157.3)"
      % (n, n, dt * 1e3, n * n / dt / 1e9, 256.0 * n * n / dt / 1e12, 256.0 * n * n / dt / 1e12 / 157.3))
ctx.close()

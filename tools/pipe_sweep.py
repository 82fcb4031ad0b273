#!/usr/bin/env python3
"""Developer aid: delivered frames/s of the host-fed pipeline (misift_pipe_*) over batch size, depth and source type."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401  (binds the HIP runtime first)
from cudasift_amd import capi
from synth import synth_frame

W, H = 1920, 1080
ctx = capi.Context(0)
base = np.stack([np.clip(np.rint(synth_frame(i)), 0, 255).astype(np.uint8) for i in range(4)])
for dt in (np.uint8, np.float32):
    for nb, depth in ((16, 3), (32, 3), (64, 2), (64, 3)):
        src = capi.PinnedArray((nb, H, W), dt)
        for i in range(nb):
            src.array[i] = base[i % 4]
        recs = capi.PinnedArray((nb * 4096,), capi.POINT_DTYPE)
        pipe = capi.Pipe(ctx, W, H, nb, src_u8=(dt == np.uint8), max_pts=8192, depth=depth)

        def run(k, fetch=True):
            for i in range(k):
                if pipe.pending() == depth:
                    pipe.collect(recs.ptr if fetch else None, nb * 4096)
                pipe.submit(src.ptr, nb)
            while pipe.pending():
                pipe.collect(recs.ptr if fetch else None, nb * 4096)
        run(depth)
        nbat = max(4, 512 // nb)
        t0 = time.perf_counter(); run(nbat); dt1 = time.perf_counter() - t0
        t0 = time.perf_counter(); run(nbat, fetch=False); dt2 = time.perf_counter() - t0
        print("%-7s batch %3d depth %d : %8.0f frames/s   (without record read-back %8.0f)" % (
            np.dtype(dt).name, nb, depth, nb * nbat / dt1, nb * nbat / dt2), flush=True)
        pipe.close(); src.free(); recs.free()

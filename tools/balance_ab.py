#!/usr/bin/env python3
"""Developer aid: batches whose frames differ in keypoint count, with and without MISIFT_BALANCE=1 (workgroups of
orient_all / descr_all dealt out in proportion to the frames' counts).  64 x 1920x1080 frames made from 4 synthetic base
frames at different contrasts; prints ms per batch for both contexts, alternating, and checks the records are the same.
usage (GPU box): python tools/balance_ab.py [uniform|skewed|one-busy]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cudasift_amd import capi  # noqa: E402
from synth import synth_frame  # noqa: E402

W, H, B, MP = 1920, 1080, 64, 32768
base = [synth_frame(4200 + i, W, H) for i in range(4)]


def batch(kind):
    if kind == "uniform":
        amps = [1.0] * B
    elif kind == "one-busy":
        amps = [1.5] + [0.8] * (B - 1)
    else:
        amps = [1.5 if i % 8 == 0 else (1.0 if i % 8 < 3 else 0.8) for i in range(B)]
    return np.stack([np.clip(128.0 + a * (base[i % 4] - 128.0), 0, 255).astype(np.float32) for i, a in enumerate(amps)])


def make_ctx(balance):
    c = capi.Context(0)
    c.set_knob("balance", 1 if balance else 0)
    c.set_options(quiet=1, fused=1)
    return c


kinds = sys.argv[1:] or ["uniform", "skewed", "one-busy"]
ctxs = {"plain": make_ctx(False), "balanced": make_ctx(True)}
S = capi.scratch_floats(W, H, 5, False)
for kind in kinds:
    frames = batch(kind)
    res = {}
    bufs = {}
    for name, c in ctxs.items():
        bufs[name] = (c.upload(frames), capi.DevBuf(4 * S * B), c.zeros(576 * MP * B), (C.c_int * B)())
    times = {n: [] for n in ctxs}
    for rep in range(12):
        for name, c in ctxs.items():
            d, sc, pts, n = bufs[name]
            t0 = time.perf_counter()
            capi.check(capi.lib().misift_extract_batch(c.h, d.ptr, B, H * W, W, H, W, 5, 1.0, 3.0, 0.0, sc.ptr, pts.ptr, MP, n),
                       "misift_extract_batch")
            times[name].append(time.perf_counter() - t0)
    counts = {}
    for name, c in ctxs.items():
        d, sc, pts, n = bufs[name]
        counts[name] = np.array(list(n))
        res[name] = c.download(pts, (B, MP), capi.POINT_DTYPE)
    assert np.array_equal(counts["plain"], counts["balanced"]), "numPts differ"
    same = True
    for f in range(B):
        k = counts["plain"][f]
        a, b = res["plain"][f, :k], res["balanced"][f, :k]
        key = lambda r: r[np.lexsort([r[x].view(np.uint32) for x in ("orientation", "scale", "ypos", "xpos")])].tobytes()
        same = same and key(a) == key(b)
    n = counts["plain"]
    print("%-9s keypoints per frame min %d / median %d / max %d (sum %d): plain %.3f ms, balanced %.3f ms per batch "
          "(medians of 12, synchronous misift_extract_batch); records identical: %s"
          % (kind, n.min(), int(np.median(n)), n.max(), n.sum(), 1e3 * np.median(times["plain"][2:]),
             1e3 * np.median(times["balanced"][2:]), same))
for c in ctxs.values():
    c.close()

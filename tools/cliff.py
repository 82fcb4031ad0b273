#!/usr/bin/env python3
"""How far do the fallback paths fall?  Batch extraction of 32 frames: 1920x1080 (fast path), 1918x1080 and 1917x1079
(width % 4 != 0), 1000x750 (a multiple of 4 whose coarser levels are not: 250, 125, 62), 1920x1080 with ONE white-noise
frame that floods its candidate list (the exact dense re-run of that frame), and 1920x1080 with MISIFT_FUSED=0 (dense
laplace/detect kernels for the whole batch)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
import bench
from cudasift_amd import capi
dev = torch.device("cuda", 0)
B = 32
base = torch.empty((B, 1080, 1920), dtype=torch.float32, device=dev)
bench.gen_frames_torch(torch, B, 0, dev, out=base)
for (w, h, fused, noise) in ((1920, 1080, 1, 0), (1918, 1080, 1, 0), (1917, 1079, 1, 0), (1000, 750, 1, 0), (1920, 1080, 1, 1),
                            (1920, 1080, 0, 0)):
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_options(quiet=1, fused=fused)
    p = (w + 127) // 128 * 128
    frames = torch.zeros((B, h, p), dtype=torch.float32, device=dev)
    frames[:, :, :w] = base[:, :h, :w]
    if noise:
        g = torch.Generator(device=dev); g.manual_seed(5)
        frames[7, :, :w] = torch.rand((h, w), generator=g, device=dev) * 255.0
    S = capi.scratch_floats(w, h, bench.NUM_OCTAVES, False)
    scratch = torch.empty((B * S,), dtype=torch.float32, device=dev)
    pts = torch.zeros((B * bench.MAX_PTS * 576,), dtype=torch.uint8, device=dev)
    counts = (C.c_int * B)()
    def run(n, prof=False):
        for _ in range(n):
            capi.check(capi.lib().misift_extract_batch(ctx.h, frames.data_ptr(), B, h * p, w, h, p, bench.NUM_OCTAVES,
                                                       bench.INIT_BLUR, bench.THRESH, 0.0, scratch.data_ptr(), pts.data_ptr(),
                                                       bench.MAX_PTS, counts), "extract")
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ctx.profile_reset(); ctx.profile_enable(True); run(3); pr = ctx.profile_read(); ctx.profile_enable(False)
    print("%dx%d fused=%d%s: %.3f ms per %d-frame batch = %.0f frames/s, keypoints %d  %s" % (
        w, h, fused, " +1 overflowing frame" if noise else "", dt * 1e3, B, B / dt, sum(counts),
        {k: round(v["total_ms"] / 3, 3) for k, v in pr.items()}), flush=True)
    ctx.close()

#!/usr/bin/env python3
"""Experiment: the timed entry point alternated over NCTX contexts (own stream, staging and scratch arena each) —
do the HBM-bound front end of one batch and the VALU-bound kernels of another overlap on the GPU?
   tools/two_ctx.py [nctx ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from cudasift_amd import capi
W, H, B, NB = bench.W, bench.H, 64, 4
dev = torch.device("cuda", 0)
frames = torch.empty((NB * B, H, W), dtype=torch.float32, device=dev)
bench.gen_frames_torch(torch, NB * B, 0, dev, out=frames)
S = capi.scratch_floats(W, H, bench.NUM_OCTAVES, False)
for nctx in [int(a) for a in sys.argv[1:]] or [1, 2]:
    streams = [torch.cuda.Stream(device=dev) for _ in range(nctx)]
    ctxs = [capi.Context(0, s.cuda_stream) for s in streams]
    for c in ctxs:
        c.set_options(quiet=1, fused=1)
    scr = [torch.empty((B * S,), dtype=torch.float32, device=dev) for _ in range(nctx)]
    NSLOT = 2 * nctx
    packed = [torch.empty((B * bench.MAX_PTS * 576,), dtype=torch.uint8, device=dev) for _ in range(NSLOT)]
    cnts = [torch.zeros((2 * B + 1,), dtype=torch.int32, device=dev) for _ in range(NSLOT)]

    def run(n):
        for k in range(n):
            i = k % nctx
            slot = k % NSLOT
            capi.check(capi.lib().misift_extract_batch_packed_async(
                ctxs[i].h, frames[(k % NB) * B].data_ptr(), B, H * W, W, H, W, bench.NUM_OCTAVES, bench.INIT_BLUR,
                bench.THRESH, 0.0, scr[i].data_ptr(), None, bench.MAX_PTS, cnts[slot].data_ptr(),
                cnts[slot][B:].data_ptr(), packed[slot].data_ptr()), "extract")
        torch.cuda.synchronize()
    run(6)
    t0 = time.perf_counter()
    n = 60
    run(n)
    dt = time.perf_counter() - t0
    tot = [int(c[2 * B].item()) for c in cnts]
    print("nctx=%d  %.1f frames/s  %.4f ms/step  records per batch %s" % (nctx, n * B / dt, 1e3 * dt / n, tot), flush=True)
    del ctxs, scr, packed

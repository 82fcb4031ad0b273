"""Developer aid (r06): does lowpass_down's duration depend on where its buffers lie?  The kernel is HBM-bound; across
processes it is bimodal (0.24 / 0.27 ms per 64 x 1080p batch).  One process: several allocations and byte offsets of the
frames and of the scratch arena, the kernel's own duration (library profile) for each.
    gpurun -- 'python tools/lowpass_placement.py'"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudasift_amd import capi

W, H, B, NOCT, MAXP = 1920, 1080, 64, 5, 32768
dev = torch.device("cuda:0")
ctx = capi.Context(0)
S = capi.scratch_floats(W, H, NOCT, False)
pts = torch.zeros((B * MAXP * 576,), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.randn((H, W), generator=g, device=dev) * 20 + 100
out = []


def run(fr_ptr, sc_ptr, reps=6):
    counts = (C.c_int * B)()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps):
        capi.check(capi.lib().misift_extract_batch(ctx.h, fr_ptr, B, H * W, W, H, W, NOCT, 1.0, 3.0, 0.0, sc_ptr,
                                                   pts.data_ptr(), MAXP, counts), "extract")
    torch.cuda.synchronize()
    p = ctx.profile_read(); ctx.profile_enable(False)
    return {k: round(v["total_ms"] / reps, 4) for k, v in p.items() if k in ("lowpass_down", "dog_scan", "descr_all", "orient_all")}


mode = sys.argv[1] if len(sys.argv) > 1 else "gap"
FB, SB = B * H * W * 4, B * S * 4
if mode == "gap":
    # ONE allocation holding the frames and, `gap` bytes behind them, the scratch arena: if the physical memory behind it is
    # contiguous, the distance of the read stream from the write stream is the gap
    big = torch.empty((FB + SB + (1 << 30),), dtype=torch.uint8, device=dev)
    f = big[:FB].view(torch.float32).view(B, H, W)
    f[:] = base
    for gap in [0, 4096, 1 << 16, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20,
                48 << 20, 64 << 20, 96 << 20, 128 << 20, 192 << 20, 256 << 20, 384 << 20, 512 << 20, 768 << 20, 1 << 30]:
        r = run(big.data_ptr(), big.data_ptr() + FB + gap)
        row = {"gap_MiB": gap / 2**20, **r}
        print(json.dumps(row), flush=True); out.append(row)
else:
    # frames fixed, scratch re-allocated (and the other way round): which stream's placement matters
    fr = torch.empty((FB,), dtype=torch.uint8, device=dev); fr.view(torch.float32).view(B, H, W)[:] = base
    keep = []
    for i in range(6):
        sc = torch.empty((SB,), dtype=torch.uint8, device=dev); keep.append(sc)
        r = run(fr.data_ptr(), sc.data_ptr())
        row = {"fixed": "frames", "frames_va": hex(fr.data_ptr()), "scratch_va": hex(sc.data_ptr()), **r}
        print(json.dumps(row), flush=True); out.append(row)
    sc = keep[0]
    for i in range(6):
        fr = torch.empty((FB,), dtype=torch.uint8, device=dev); keep.append(fr); fr.view(torch.float32).view(B, H, W)[:] = base
        r = run(fr.data_ptr(), sc.data_ptr())
        row = {"fixed": "scratch", "frames_va": hex(fr.data_ptr()), "scratch_va": hex(sc.data_ptr()), **r}
        print(json.dumps(row), flush=True); out.append(row)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_lowpass_placement_%s.json" % mode, "w"), indent=1)

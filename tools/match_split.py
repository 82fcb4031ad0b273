#!/usr/bin/env python3
"""What does the column cut of the sharded matcher cost on ONE GPU?  A rank of an 8-GPU 100k x 100k job: 12 500 rows
against 100 000 columns, as one sweep and as own-shard tiles [195, 390) + the rest (misift_test_match_split)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from cudasift_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
out = {}
for rows, n2, ranks in ((12500, 100000, 8), (25000, 100000, 4), (50000, 100000, 2)):
    g = torch.Generator(device=dev); g.manual_seed(3)
    def pts(n):
        t = torch.zeros((n, 144), dtype=torch.float32, device=dev)
        d = torch.rand((n, 128), generator=g, device=dev)
        t[:, 16:] = d / d.norm(dim=1, keepdim=True)
        return t
    p1, p2 = pts(rows), pts(n2)
    L = capi.lib()
    shard = n2 // ranks
    t0, t1 = (shard + 63) // 64, (2 * shard) // 64            # rank 1's own tiles
    def run(split, n=6):
        ms = []
        for _ in range(n):
            torch.cuda.synchronize(); a = time.perf_counter()
            if split:
                capi.check(L.misift_test_match_split(ctx.h, p1.data_ptr(), rows, p2.data_ptr(), n2, t0, t1), "split")
            else:
                capi.check(L.misift_match(ctx.h, p1.data_ptr(), rows, p2.data_ptr(), n2), "match")
            ms.append((time.perf_counter() - a) * 1e3)
        return min(ms[1:]), sorted(ms[1:])[len(ms[1:]) // 2]
    run(False, 3)
    one = run(False); ref = p1[:, 8:11].clone()
    two = run(True)
    same = bool(torch.equal(ref, p1[:, 8:11]))
    flop = 2.0 * 128 * rows * n2
    out["%d_ranks" % ranks] = dict(rows=rows, n2=n2, own_tiles=[t0, t1], one_sweep_ms=one, split_ms=two, same_bits=same,
                                   one_sweep_frac=flop / (one[0] * 1e-3) / 157.3e12, split_frac=flop / (two[0] * 1e-3) / 157.3e12)
    print(out["%d_ranks" % ranks], flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)

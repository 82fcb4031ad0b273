#!/usr/bin/env python3
"""Matcher time against the number of column chunks (MISIFT_MATCH_CHUNKS, read once per process: one child per value)."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import time, torch
    from cudasift_amd import capi
    rows, n2 = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(3)
    def pts(n):
        t = torch.zeros((n, 144), dtype=torch.float32, device=dev)
        d = torch.rand((n, 128), generator=g, device=dev)
        t[:, 16:] = d / d.norm(dim=1, keepdim=True)
        return t
    p1, p2 = pts(rows), pts(n2)
    L = capi.lib()
    for _ in range(3):
        capi.check(L.misift_match(ctx.h, p1.data_ptr(), rows, p2.data_ptr(), n2), "m")
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(8):
        capi.check(L.misift_match(ctx.h, p1.data_ptr(), rows, p2.data_ptr(), n2), "m")
    pr = ctx.profile_read()
    ms = pr["match_mfma"]["total_ms"] / pr["match_mfma"]["calls"]
    mg = pr["match_merge"]["total_ms"] / pr["match_merge"]["calls"]
    print(json.dumps(dict(sweep_ms=ms, merge_ms=mg, frac=2.0 * 128 * rows * n2 / (ms * 1e-3) / 157.3e12)))
    sys.exit(0)
out = {}
SHAPES = {"12500x100000": "0,13,26", "25000x100000": "0,13,26", "100000x100000": "0,16,17,18", "2000x2000": "0,8,16,20,31",
          "16384x16384": "0,2,4,6,8,10,12,16", "1500x1500": "0,4,8,12,23"}
if os.environ.get("CHUNKS"):
    SHAPES = {k: os.environ["CHUNKS"] for k in SHAPES}
for shape, cs in SHAPES.items():
    rows, n2 = [int(x) for x in shape.split("x")]
    for c in [int(x) for x in cs.split(",")]:
        env = dict(os.environ)
        if c: env["MISIFT_MATCH_CHUNKS"] = str(c)
        r = subprocess.run([sys.executable, __file__, "child", str(rows), str(n2)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out["%dx%d c=%d" % (rows, n2, c)] = json.loads(line[-1]) if line else r.stderr[-300:]
        print(rows, n2, "chunks", c or "auto", out["%dx%d c=%d" % (rows, n2, c)], flush=True)
json.dump(out, open(os.environ.get("OUT", "/dev/stdout"), "w"), indent=1)

#!/usr/bin/env python3
"""Every input the reference accepts: N random small shapes (width, height in [1, 48], 1-6 octaves, white noise, low
thresholds) through misift_extract on the GPU and through the reference's own ExtractSift on the CPU SIMT emulator
(oracle/_ref, prebuilt) — numPts, the 17 counters (duplicate counters +-1: tests/util.py compare_tiny) and every keypoint.
-> gpurun_out/r06_tiny_sweep.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MISIFT_QUIET", "1")
os.environ.setdefault("SIMT_THREADS", "16")
from cudasift_amd import capi                      # noqa: E402
from oracle import pyrefemul as ref                # noqa: E402
from util import compare_tiny                      # noqa: E402

N = int(os.environ.get("TINY_N", "300"))
rng = np.random.default_rng(2025)
ctx = capi.Context(0)
ctx.set_options(quiet=1)
bad, total_pts, shapes = [], 0, []
for i in range(N):
    w, h = int(rng.integers(1, 49)), int(rng.integers(1, 49))
    noct = int(rng.integers(1, 7))
    th = float(rng.choice([0.05, 0.2, 0.5, 1.0]))
    up = bool(rng.random() < 0.1)
    img = rng.uniform(0, 255, (h, w)).astype(np.float32)
    try:
        r_pts, r_n, r_cnt = ref.extract(img, noct, 1.0, th, scale_up=up, flavour="fast")
        pts, n, cnt = ctx.extract(img, num_octaves=noct, init_blur=1.0, thresh=th, scale_up=up)
        compare_tiny(pts, n, cnt, r_pts, r_n, r_cnt, noct)
        total_pts += int(n)
    except Exception as e:                          # noqa: BLE001 — collected and reported
        bad.append({"w": w, "h": h, "octaves": noct, "thresh": th, "scale_up": up, "error": repr(e)[:300]})
    shapes.append((w, h, noct))
out = {"shapes": N, "failures": len(bad), "failed": bad[:20], "keypoints_total": total_pts,
       "smallest": [int(min(s[0] for s in shapes)), int(min(s[1] for s in shapes))],
       "what": "misift_extract (MI355X) vs the emulated reference on random shapes 1..48 x 1..48, 1-6 octaves, 10 % with scaleUp"}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_tiny_sweep.json"), "w"), indent=1)
print(json.dumps(out)[:1500])

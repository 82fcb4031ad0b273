"""Developer aid: time stamps (100 MHz) of the ScaleDown chain embedded in the scan launch — needs a library built with
-DSCAN_STAMPS=1 (tools/variants.sh kernels_dog.hip stamps "-DSCAN_STAMPS=1"; MISIFT_LIB=build/variants/libmisift_stamps.so)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from synth import synth_frame
from cudasift_amd import capi
ctx = capi.Context(0)
img = synth_frame(0, 1920, 1080)
for it in range(4):
    pts, n, cnt = ctx.extract(img, num_octaves=5, thresh=3.0)
    w = ctx.get_counter_block(1).astype(np.int64)
    t0 = w[8]
    names = {17: "first scan wg starts", 14: "last wg starts", 16: "highest-index wg starts", 9: "last chain wg done", 19: "last ticket returned", 18: "flag raised", 15: "last waiting wg starts to wait", 13: "last waiting wg released", 11: "last fine item done", 12: "last coarse item done"}
    print("call %d: %d points;" % (it, n), ", ".join("%s +%.2f us" % (names[k], (w[k] - t0) / 100.0) for k in (17, 14, 16, 9, 19, 18, 15, 13, 11, 12)))
    print("   candidates per octave", ctx.get_counter_block(0)[20:28], "detections", ctx.get_counter_block(0)[32:40])

#!/usr/bin/env python3
"""The HIP path against the reference's OWN code, no oracle in between, at scale: N synthetic 1920x1080 frames (+ the stereo
pair) extracted by libmisift.so on the GPU and by the emulated reference (oracle/_ref/libcudasift_refemul_fast.so: the
reference's kernels and host code on the CPU SIMT emulator, prebuilt — it travels to the GPU box) -> pooled statistics,
gpurun_out/<HVR_TAG>_hip_vs_refemul.json.  Test infrastructure, like tests/test_gpu_golden.py, which asserts the same on five
committed golden cases."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cudasift_amd import capi                      # noqa: E402
from oracle import pyrefemul as ref                # noqa: E402
from synth import synth_frame                      # noqa: E402
from refemul_report import stats                   # noqa: E402

N = int(os.environ.get("HVR_FRAMES", "32"))
TAG = os.environ.get("HVR_TAG", "r06")
z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_pair_u8.npz"))
cases = [("left.pgm thresh 3.0", z["left"].astype(np.float32), 5, 3.0), ("righ.pgm thresh 3.0", z["right"].astype(np.float32), 5, 3.0)]
cases += [("synthetic 1920x1080 frame %d" % f, None, 5, 3.0) for f in range(N)]
cases = [c + (1.0, 0.0, False) for c in cases]                 # (name, image, octaves, thresh, initBlur, lowestScale, scaleUp)
if os.environ.get("HVR_VARIANTS"):                             # the rest of ExtractSift's argument surface
    cases = [("scaleUp 960x540 frame %d" % f, synth_frame(300 + f, 960, 540), 5, 3.0, 1.0, 0.0, True) for f in range(8)]
    cases += [("initBlur 0.5, 1280x960", synth_frame(310, 1280, 960), 5, 2.0, 0.5, 0.0, False),
              ("initBlur 2.0, 1280x960", synth_frame(311, 1280, 960), 5, 1.0, 2.0, 0.0, False),
              ("initBlur 0 (delta), 1280x960", synth_frame(312, 1280, 960), 5, 3.0, 0.0, 0.0, False),
              ("lowestScale 2.0, 1920x1080", synth_frame(313), 5, 2.0, 1.0, 2.0, False),
              ("6 octaves, 2560x1440", synth_frame(79, 2560, 1440), 6, 2.5, 1.0, 0.0, False),
              ("1 octave, 1920x1080", synth_frame(314), 1, 3.0, 1.0, 0.0, False),
              ("4096x3072", synth_frame(77, 4096, 3072), 5, 3.0, 1.0, 0.0, False),
              ("1917x1079 (ragged)", synth_frame(315, 1917, 1079), 5, 3.0, 1.0, 0.0, False),
              ("scaleUp + lowestScale 1.5, left.pgm crop", z["left"][200:440, 300:620].astype(np.float32), 4, 3.0, 1.0, 1.5, True)]
ctx = capi.Context(0)
ctx.set_options(quiet=1)
out = {"what": "libmisift.so (MI355X) vs the reference's own kernels and host code on the CPU SIMT emulator (-ffp-contract=fast "
               "build), no oracle in between", "images": []}
pooled = {}
t_hip = t_ref = 0.0
for name, img, noct, th, blur, lowest, up in cases:
    if img is None:
        img = synth_frame(int(name.split()[-1]))
    t0 = time.time()
    hp, hn, hc = ctx.extract(img, num_octaves=noct, init_blur=blur, thresh=th, lowest_scale=lowest, scale_up=up)
    t1 = time.time()
    rp, rn, rc = ref.extract(img, noct, blur, th, lowest_scale=lowest, scale_up=up, flavour="fast")
    t2 = time.time()
    t_hip += t1 - t0; t_ref += t2 - t1
    st = stats(hp, hc, rp, rc, noct, img=img, init_blur=blur, scale_up=up)
    out["images"].append({"image": name, "numPts_hip": hn, "numPts_reference": rn, "only_hip": st["only_oracle"], **{k: st[k] for k in (
        "records", "counters_equal", "only_reference", "orientation_flips", "desc_over_0.0001", "desc_over_0.001")},
        "desc_over_bound": st.get("desc_over_bound"), "desc_explained": st.get("desc_explained"),
        "desc_unexplained": st.get("desc_unexplained")})
    for kk, v in st.items():
        if isinstance(v, bool):
            pooled[kk] = pooled.get(kk, True) and v
        elif kk.endswith("_max") or kk == "desc_max":
            pooled[kk] = max(pooled.get(kk, 0.0), v)
        elif kk == "desc_min_cos":
            pooled[kk] = min(pooled.get(kk, 1.0), v)
        else:
            pooled[kk] = pooled.get(kk, 0) + v
    print(name, hn, rn, st["counters_equal"], flush=True)
pooled["only_hip"] = pooled.pop("only_oracle")          # stats() names its first argument "oracle"
out["pooled_hip_vs_reference"] = pooled
out["seconds"] = {"hip_single_frame_calls_incl_upload": round(t_hip, 2), "emulated_reference": round(t_ref, 2)}
path = os.path.join(ROOT, "gpurun_out", TAG + "_hip_vs_refemul_variants.json" if os.environ.get("HVR_VARIANTS") else
                    TAG + "_hip_vs_refemul.json" if N else TAG + "_hip_vs_refemul_match.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(pooled, indent=1))

# ---- matcher: the reference's MatchSiftData (CleanMatches + FindMaxCorr10 on the emulator) vs misift_match, bit for bit
if os.environ.get("HVR_MATCH_N"):
    from synth import descriptors_to_points, synth_descriptors
    n1 = int(os.environ["HVR_MATCH_N"]); n2 = n1 + 37                      # n2 % 32 != 0: the reference drops the last 5 columns
    a = descriptors_to_points(synth_descriptors(n1, 71, l2=True), capi.POINT_DTYPE)
    b = descriptors_to_points(synth_descriptors(n2, 72, l2=True), capi.POINT_DTYPE)
    t0 = time.time()
    want = a.copy()
    ref.match(want, n1, b.copy(), n2, "fast")
    t1 = time.time()
    got = ctx.match(a.copy(), n1, b.copy(), n2)
    m = {"n1": n1, "n2": n2, "emulated_reference_s": round(t1 - t0, 1),
         **{f + "_identical": bool(np.array_equal(got[f], want[f])) for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos")}}
    print("matcher", m)
    out["matcher_hip_vs_reference"] = m
    json.dump(out, open(path, "w"), indent=1)

# ---- FindHomography: ComputeHomographies + TestHomographies on the emulator vs homography.hip, same libc rand() state
if os.environ.get("HVR_HOMOG_N"):
    from oracle import pyoracle as orc
    from synth import synth_matches
    res = []
    for n, loops, seed in ((int(os.environ["HVR_HOMOG_N"]), 10000, 1), (5000, 4000, 7), (900, 1000, 3)):
        m, _, _ = synth_matches(n, seed=11 + seed, dtype=capi.POINT_DTYPE)
        t0 = time.time()
        Hr, nr = ref.find_homography(m.copy(), n, loops, 0.85, 0.95, 5.0, seed=seed, flavour="fast")
        t1 = time.time()
        d = ctx.upload(m)
        orc.srand(seed)                                        # the libc state both sides draw their samples from
        Hh, nh = ctx.find_homography(d.ptr, n, num_loops=loops, min_score=0.85, max_ambiguity=0.95, thresh=5.0)
        res.append({"points": n, "loops": loops, "seed": seed, "emulated_reference_s": round(t1 - t0, 1), "inliers_reference": int(nr),
                    "inliers_hip": int(nh), "H_identical": bool(np.array_equal(Hr, Hh))})
        print("homography", res[-1])
    out["find_homography_hip_vs_reference"] = res
    json.dump(out, open(path, "w"), indent=1)

# ---- the BATCH entry point (misift_extract_batch: all frames in one set of launches, the bench's kernels) vs the reference
if os.environ.get("HVR_BATCH"):
    import util
    B = int(os.environ["HVR_BATCH"])
    frames = np.stack([synth_frame(500 + f) for f in range(B)])
    pts, counts = ctx.extract_batch(frames, thresh=3.0)
    res = {"frames": B, "numPts_equal": 0, "records": 0, "only_hip": 0, "only_reference": 0, "field_relerr_max": 0.0,
           "orientation_flips": 0, "desc_over_1e-4": 0}
    for f in range(B):
        rp, rn, rc = ref.extract(frames[f], 5, 1.0, 3.0, flavour="fast")
        n = int(counts[f])
        res["numPts_equal"] += int(n == rn)
        ia, ib, oh, orr = util.associate(pts[f][:n], rp[:rn])
        A, Rr = pts[f][:n][ia], rp[:rn][ib]
        res["records"] += n; res["only_hip"] += len(oh); res["only_reference"] += len(orr)
        res["field_relerr_max"] = max(res["field_relerr_max"], *[float(util.rel_err(A[k], Rr[k]).max()) for k in ("xpos", "ypos", "scale", "sharpness", "edgeness")])
        od = util.circ_diff_deg(A["orientation"], Rr["orientation"])
        res["orientation_flips"] += int((od > 0.036).sum())
        ok = ~np.isnan(Rr["data"]).any(axis=1) & (od <= 0.036)
        res["desc_over_1e-4"] += int((np.abs(A["data"][ok].astype(np.float64) - Rr["data"][ok]).max(axis=1) > 1e-4).sum())
    print("batch", res)
    out["batch_entry_point_hip_vs_reference"] = res
    json.dump(out, open(path, "w"), indent=1)

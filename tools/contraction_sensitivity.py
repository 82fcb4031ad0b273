#!/usr/bin/env python
"""Error bar on the unpinned extraction oracle: run oracle/sift_oracle.c in both contraction
modes (plain = what the HIP kernels implement; nvcc = LLVM/NVPTX-style fused multiply-adds in the refinement,
orientation and descriptor code) on the same images and report what the choice can change.

    python tools/contraction_sensitivity.py [--out profiles/r02_contraction_sensitivity.json] [--big]

Test infrastructure: uses oracle/ only.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sensitivity(img, **kw):
    """Extract `img` in both modes; pair the keypoints; return the report dict."""
    from oracle import pyoracle as orc
    from util import associate, circ_diff_deg
    a, na, ca = orc.extract(img, **kw)
    with orc.contract(1):
        b, nb, cb = orc.extract(img, **kw)
    ta, tb = int(ca[2 * kw.get("num_octaves", 5) + 1]), int(cb[2 * kw.get("num_octaves", 5) + 1])
    A, B = a[:min(ta, len(a))], b[:min(tb, len(b))]
    ia, ib, only_a, only_b = associate(A, B)          # exact identity first, then 1e-3 px / 1e-3 rel. scale
    PA, PB = A[ia], B[ib]
    rep = {"n_plain": int(len(A)), "n_nvcc": int(len(B)), "paired": int(len(ia)),
           "paired_bit_identical_position": int(associate.last_exact),
           "only_plain": int(len(only_a)), "only_nvcc": int(len(only_b)),
           "jaccard": float(len(ia) / max(1, len(ia) + len(only_a) + len(only_b))),
           "numPts_plain": int(na), "numPts_nvcc": int(nb)}
    if len(ia):
        sub = PA["subsampling"].astype(np.float64)
        rep["max_dpos_octave_px"] = float(max((np.abs(PA["xpos"].astype(np.float64) - PB["xpos"]) / sub).max(),
                                              (np.abs(PA["ypos"].astype(np.float64) - PB["ypos"]) / sub).max()))
        rep["max_rel_dscale"] = float((np.abs(PA["scale"].astype(np.float64) - PB["scale"]) / PA["scale"]).max())
        rep["max_rel_dsharpness"] = float((np.abs(PA["sharpness"].astype(np.float64) - PB["sharpness"]) /
                                           np.maximum(np.abs(PA["sharpness"]), 1.0)).max())
        rep["max_rel_dedgeness"] = float((np.abs(PA["edgeness"].astype(np.float64) - PB["edgeness"]) /
                                          np.maximum(np.abs(PA["edgeness"]), 1.0)).max())
        od = circ_diff_deg(PA["orientation"], PB["orientation"])
        rep["max_dorientation_deg"] = float(od.max())
        rep["orientation_gt_0.036deg"] = int((od > 0.036).sum())
        dd = np.abs(PA["data"].astype(np.float64) - PB["data"]).max(axis=1)
        rep["max_ddescriptor"] = float(dd.max())
        rep["descriptor_gt_1e-4"] = int((dd > 1e-4).sum())
        rep["min_descriptor_cos"] = float((PA["data"].astype(np.float64) * PB["data"]).sum(axis=1).min())
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_contraction_sensitivity.json"))
    ap.add_argument("--big", action="store_true", help="also the 4096x3072 synthetic frame (~1 min)")
    args = ap.parse_args()
    from synth import synth_frame
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_pair_u8.npz"))
    cases = [("left.pgm 1280x960 thresh 4.5", z["left"].astype(np.float32), dict(thresh=4.5)),
             ("righ.pgm 1280x960 thresh 4.5", z["right"].astype(np.float32), dict(thresh=4.5)),
             ("left.pgm 1280x960 thresh 2.0", z["left"].astype(np.float32), dict(thresh=2.0))]
    for f in range(4):
        cases.append(("synthetic 1920x1080 frame %d thresh 3.0" % f, synth_frame(f), dict(thresh=3.0)))
    if args.big:
        cases.append(("synthetic 4096x3072 frame 4242 thresh 3.0", synth_frame(4242, width=4096, height=3072),
                      dict(thresh=3.0)))
    out = {"modes": {"plain": "uncontracted outside the separable filters (HIP kernels, all parity tests)",
                     "nvcc": "LLVM/NVPTX-style contraction of the reference expressions (cudaSiftD.cu:1383-1417, "
                             ":1006-1013, :299-303, :337-345), left multiply first"},
           "cases": {}}
    tot = {"paired": 0, "only_plain": 0, "only_nvcc": 0}
    for name, img, kw in cases:
        rep = sensitivity(img, num_octaves=5, init_blur=1.0, **kw)
        out["cases"][name] = rep
        for k in tot:
            tot[k] += rep[k]
        print(name, json.dumps(rep))
    tot["jaccard"] = tot["paired"] / max(1, tot["paired"] + tot["only_plain"] + tot["only_nvcc"])
    out["total"] = tot
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("total", tot)


if __name__ == "__main__":
    main()

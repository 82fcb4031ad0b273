#!/usr/bin/env python3
"""Oracle (oracle/sift_oracle.c) against the reference's own kernels on the CPU SIMT emulator
(oracle/_ref/libcudasift_refemul_{fast,off}.so) over a set of images -> profiles/r03_refemul_report.json.
CPU only; needs oracle/_ref (built where /root/reference exists).  The assertions live in tests/test_refemul_cpu.py; this
writes the statistics down, per image and pooled."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc, pyrefemul as ref   # noqa: E402
from synth import synth_frame                           # noqa: E402
import util                                             # noqa: E402


def stats(o_pts, o_cnt, r_pts, r_cnt, noct, img=None, init_blur=1.0, scale_up=False):
    """img: the image both sides extracted from -> every descriptor pair over 1e-4 is also EXPLAINED (oracle.descriptor_explain:
    the other side's descriptor reproduced by flipping a few 8-bit texture weights that sit on a rounding tie; tests/util.py
    EXPLAIN_*) and checked element by element against its worst-case texture-weight bound (oracle.descriptor_bounds,
    BOUND_*): desc_unexplained / desc_over_bound count the failures."""
    total = int(o_cnt[2 * noct + 1])
    st = {"counters_equal": bool(np.array_equal(o_cnt, r_cnt)), "records": total}
    O, R = o_pts[:total], r_pts[:int(r_cnt[2 * noct + 1])]
    ia, ib, only_o, only_r = util.associate(O, R)
    st["only_oracle"], st["only_reference"] = len(only_o), len(only_r)
    A, B = O[ia], R[ib]
    for f in ("xpos", "ypos", "scale", "sharpness", "edgeness"):
        st[f + "_relerr_max"] = float(util.rel_err(A[f], B[f]).max()) if len(ia) else 0.0
    od = util.circ_diff_deg(A["orientation"], B["orientation"])
    st["orientation_deg_max"] = float(od.max()) if len(od) else 0.0
    st["orientation_flips"] = int((od > 0.036).sum())
    nan_ref = np.isnan(B["data"]).any(axis=1)
    st["nan_descriptors_reference"] = int(nan_ref.sum())
    ok = ~nan_ref & (od <= 0.036)
    dd = np.abs(A["data"][ok].astype(np.float64) - B["data"][ok]).max(axis=1)
    st["descriptors_compared"] = int(ok.sum())
    for t in (1e-6, 1e-5, 1e-4, 1e-3):
        st["desc_over_%g" % t] = int((dd > t).sum())
    st["desc_max"] = float(dd.max()) if len(dd) else 0.0
    st["desc_min_cos"] = float((A["data"][ok].astype(np.float64) * B["data"][ok]).sum(axis=1).min()) if ok.any() else 1.0
    if img is not None:
        big = np.where(dd > 1e-4)[0]
        st["desc_bound_checked"], st["desc_over_bound"], st["desc_diff_over_bound_max"] = int(len(big)), 0, 0.0
        st["desc_explained"], st["desc_partly_explained"], st["desc_unexplained"], st["desc_residual_max"] = 0, 0, 0, 0.0
        if len(big):
            Ab, Bb = A[ok][big], B[ok][big]
            cs = None
            if scale_up:         # records below numPts were halved by RescalePositions (cudaSiftD.cu:753-761)
                cs = np.where(np.asarray(ia)[ok][big] < int(o_cnt[2 * noct]), 2.0, 1.0).astype(np.float32)
            # the tight form (r06): the other side's descriptor reproduced by flipping a few tie weights / seam decisions
            res, nset, _ = orc.descriptor_explain(img, Ab, Bb["data"], Bb, noct, init_blur, ulps=util.EXPLAIN_ULPS,
                                                  scale_up=scale_up, coord_scale=cs, tol=util.EXPLAIN_TOL)
            st["desc_explained"] = int((res <= util.EXPLAIN_TOL).sum())
            st["desc_partly_explained"] = int(((res > util.EXPLAIN_TOL) & (res <= util.EXPLAIN_PARTIAL)).sum())
            st["desc_unexplained"] = int((res > util.EXPLAIN_PARTIAL).sum())
            st["desc_residual_max"] = float(res.max())
            st["desc_toggles_median"] = float(np.median(nset))
            # the worst case (r05): every candidate fetch flipping the same way
            bound, _, _ = orc.descriptor_bounds(img, Ab, len(big), noct, init_blur, util.BOUND_ULPS,
                                                dtheta_deg=util.circ_diff_deg(Ab["orientation"], Bb["orientation"]),
                                                scale_up=scale_up, coord_scale=cs)
            r = (np.abs(Ab["data"].astype(np.float64) - Bb["data"]) / (bound + util.BOUND_SLACK)).max(axis=1)
            st["desc_over_bound"] = int((r > 1.0).sum())
            st["desc_diff_over_bound_max"] = float(r.max())
    return st


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_pair_u8.npz"))
    L, R = z["left"].astype(np.float32), z["right"].astype(np.float32)
    cases = [("left.pgm thresh 4.5", L, 5, 4.5), ("righ.pgm thresh 4.5", R, 5, 4.5), ("left.pgm thresh 2.0", L, 5, 2.0),
             ("righ.pgm thresh 3.0", R, 5, 3.0)]
    cases += [("synthetic 1920x1080 frame %d" % f, synth_frame(f), 5, 3.0) for f in range(4)]
    cases += [("synthetic 4096x3072", synth_frame(77, 4096, 3072), 5, 3.0), ("synthetic 1000x750", synth_frame(78, 1000, 750), 5, 3.0)]
    large = os.environ.get("REFEMUL_SET") in ("large", "xlarge")    # the long runs: -> profiles/r03_refemul_report_large.json
    xlarge = os.environ.get("REFEMUL_SET") == "xlarge"              #                   profiles/r03_refemul_report_xlarge.json
    if large:
        cases += [("synthetic 1920x1080 frame %d" % f, synth_frame(f), 5, 3.0) for f in range(4, 24)]
        cases += [("left.pgm mirrored to 1920x1080, thresh 3.0", np.pad(z["left"], ((60, 60), (320, 320)), mode="reflect").astype(np.float32), 5, 3.0),
                  ("righ.pgm mirrored to 1920x1080, thresh 2.0", np.pad(z["right"], ((60, 60), (320, 320)), mode="reflect").astype(np.float32), 5, 2.0),
                  ("left.pgm thresh 1.0", L, 5, 1.0), ("synthetic 2560x1440", synth_frame(79, 2560, 1440), 6, 2.5),
                  ("synthetic 641x479 (odd)", synth_frame(80, 641, 479), 4, 2.0)]
    if xlarge:                                                # 232 more 1080p frames: 256 in all (half a bench job)
        cases += [("synthetic 1920x1080 frame %d" % f, None, 5, 3.0) for f in range(24, 256)]
    out = {"what": "oracle (nvcc-contraction mode / plain mode) vs the reference's own kernels on the CPU SIMT emulator "
                   "(-ffp-contract=fast build), and plain oracle vs the -ffp-contract=off build", "images": []}
    pooled = {}
    for name, img, noct, th in cases:
        if img is None:                                       # generated on demand (8 MB each)
            img = synth_frame(int(name.split()[-1]))
        t0 = time.time()
        rp, rn, rc = ref.extract(img, noct, 1.0, th, flavour="fast")
        t_ref = time.time() - t0
        orc.stats_reset()
        with orc.contract(1):
            op, on, oc = orc.extract(img, noct, 1.0, th)
        e = {"image": name, "numPts_oracle": on, "numPts_reference": rn, "emulator_seconds": round(t_ref, 2),
             "oracle_nvcc_vs_fast": stats(op, oc, rp, rc, noct)}
        op2, on2, oc2 = orc.extract(img, noct, 1.0, th)
        e["oracle_plain_vs_fast"] = stats(op2, oc2, rp, rc, noct)
        rp3, rn3, rc3 = ref.extract(img, noct, 1.0, th, flavour="off")
        e["oracle_plain_vs_off"] = stats(op2, oc2, rp3, rc3, noct)
        e["oracle_guards"] = orc.stats()
        out["images"].append(e)
        for k in ("oracle_nvcc_vs_fast", "oracle_plain_vs_fast", "oracle_plain_vs_off"):
            p = pooled.setdefault(k, {})
            for kk, v in e[k].items():
                if isinstance(v, bool):
                    p[kk] = p.get(kk, True) and v
                elif kk.endswith("_max") or kk == "desc_max":
                    p[kk] = max(p.get(kk, 0.0), v)
                elif kk == "desc_min_cos":
                    p[kk] = min(p.get(kk, 1.0), v)
                else:
                    p[kk] = p.get(kk, 0) + v
        print(name, on, rn, e["oracle_nvcc_vs_fast"]["counters_equal"], "%.1fs" % t_ref, flush=True)
    out["pooled"] = pooled
    path = os.path.join(ROOT, "profiles", "r03_refemul_report_xlarge.json" if xlarge else
                        "r03_refemul_report_large.json" if large else "r03_refemul_report.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(pooled, indent=1))


if __name__ == "__main__":
    main()

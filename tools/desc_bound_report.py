#!/usr/bin/env python3
"""The descriptor tail against its per-record bound, pooled (CPU): oracle (= the HIP path to 1e-6, asserted by the GPU suite)
vs the reference's own kernels on the SIMT emulator, N synthetic 1920x1080 frames + the stereo pair.  For every associated
pair that differs by more than 1e-4 in some element: is EVERY element within oracle.descriptor_bounds() (what a last-bit
difference of the sample coordinates can do through the 8-bit texture weights, + the angi = 8 seam)?  -> profiles/r06_desc_bound_report.json
usage: SIMT_THREADS=8 DBR_FRAMES=48 python tools/desc_bound_report.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MISIFT_QUIET", "1")
from oracle import pyoracle as orc, pyrefemul as ref      # noqa: E402
from synth import synth_frame                              # noqa: E402
import util                                                # noqa: E402

N = int(os.environ.get("DBR_FRAMES", "32"))
# DBR_ULPS="6,3,2": the same pairs against the bound at several BOUND_ULPS (first = the one the summary reports)
ULPS = [float(u) for u in os.environ.get("DBR_ULPS", str(util.BOUND_ULPS)).split(",")]
OUT = os.environ.get("DBR_OUT", os.path.join(ROOT, "profiles", "r06_desc_bound_report.json"))
z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_pair_u8.npz"))
cases = [("left.pgm", z["left"].astype(np.float32), 5, 3.0), ("righ.pgm", z["right"].astype(np.float32), 5, 3.0)]
cases += [("synthetic 1920x1080 frame %d" % f, None, 5, 3.0) for f in range(N)]
tot = {"images": 0, "records": 0, "over_1e-4": 0, "over_1e-3": 0, "over_bound": 0, "seam_records_among_over_1e-4": 0,
       "explained": 0, "unexplained": 0}
residuals, toggles = [], []
ratios, worst = [], 0.0
sweep = {u: [] for u in ULPS}
for name, img, noct, th in cases:
    if img is None:
        img = synth_frame(int(name.split()[-1]))
    rp, rn, rc = ref.extract(img, noct, 1.0, th, flavour="fast")
    op, on, oc = orc.extract(img, noct, 1.0, th)
    t = int(oc[2 * noct + 1])
    ia, ib, _, _ = util.associate(op[:t], rp[:int(rc[2 * noct + 1])])
    A, B = op[:t][ia], rp[:int(rc[2 * noct + 1])][ib]
    od = util.circ_diff_deg(A["orientation"], B["orientation"])
    ok = (od <= 0.036) & ~np.isnan(B["data"]).any(axis=1)
    A, B = A[ok], B[ok]
    dd = np.abs(A["data"].astype(np.float64) - B["data"])
    big = np.where(dd.max(axis=1) > 1e-4)[0]
    tot["images"] += 1; tot["records"] += int(len(A)); tot["over_1e-4"] += int(len(big)); tot["over_1e-3"] += int((dd.max(axis=1) > 1e-3).sum())
    if len(big):
        dth = util.circ_diff_deg(A["orientation"][big], B["orientation"][big])
        for u in ULPS:
            bound_u, flips_u, wraps_u = orc.descriptor_bounds(img, A[big], len(big), noct, 1.0, u, dtheta_deg=dth)
            sweep[u] += (dd[big] / (bound_u + util.BOUND_SLACK)).max(axis=1).tolist()
            if u == ULPS[0]:
                bound, flips, wraps = bound_u, flips_u, wraps_u
        r = (dd[big] / (bound + util.BOUND_SLACK)).max(axis=1)
        ratios += r.tolist()
        # the tight form (r06): the reference's descriptor REPRODUCED by flipping a few tie weights / seam decisions
        res, nset, ncand = orc.descriptor_explain(img, A[big], B["data"][big], B[big], noct, 1.0,
                                                  ulps=util.EXPLAIN_ULPS, tol=util.EXPLAIN_TOL)
        residuals += res.tolist(); toggles += nset.tolist()
        tot["explained"] += int((res <= util.EXPLAIN_TOL).sum()); tot["unexplained"] += int((res > util.EXPLAIN_TOL).sum())
        tot["over_bound"] += int((r > 1.0).sum())
        tot["seam_records_among_over_1e-4"] += int((wraps > 0).sum())
    print(name, len(A), len(big), tot["over_bound"], flush=True)
ratios = np.array(ratios)
tot["sweep"] = {str(u): {"over_bound": int((np.array(v) > 1.0).sum()), "max": float(np.max(v)) if v else 0.0,
                         "p99": float(np.percentile(v, 99)) if v else 0.0, "median": float(np.median(v)) if v else 0.0}
                for u, v in sweep.items()}
tot.update({"explain_tol": util.EXPLAIN_TOL, "explain_ulps": util.EXPLAIN_ULPS,
            "explain_residual_max": float(np.max(residuals)) if residuals else 0.0,
            "explain_residual_median": float(np.median(residuals)) if residuals else 0.0,
            "explain_toggles_median": float(np.median(toggles)) if toggles else 0.0,
            "explain_toggles_max": int(np.max(toggles)) if toggles else 0})
tot.update({"bound_ulps": ULPS[0], "bound_slack": util.BOUND_SLACK,
            "diff_over_bound_max": float(ratios.max()) if len(ratios) else 0.0,
            "diff_over_bound_median": float(np.median(ratios)) if len(ratios) else 0.0,
            "diff_over_bound_p99": float(np.percentile(ratios, 99)) if len(ratios) else 0.0,
            "what": "oracle (plain arithmetic: what the HIP kernels implement) vs the emulated reference (-ffp-contract=fast build); "
                    "every record over 1e-4 checked element by element against oracle.descriptor_bounds()"})
json.dump(tot, open(OUT, "w"), indent=1)
print(json.dumps(tot, indent=1))

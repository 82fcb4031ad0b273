"""Summaries for profiles/: (a) the GPU suite's parity report (gpurun_out/parity_report.json, one entry per compared
case) -> one line per GROUP of cases with the worst figures; (b) the per-image lists of the big reference-comparison
reports -> their pooled figures + the image count.  Raw reports stay under gpurun_out/ (scratch).

    python tools/summarise_parity.py parity <parity_report.json> <out.json>
    python tools/summarise_parity.py pooled <report.json> <out.json>
"""
import json
import re
import sys


def parity(src, dst):
    d = json.load(open(src))
    groups = {}
    for key, v in d.items():
        g = re.sub(r"(_f\d+|/\d+|_\d+x\d+.*|_o\d+|_s[\d.]+|/seed\d+.*|_n\d+)$", "", key)
        g = re.sub(r"_f\d+$", "", g)
        e = groups.setdefault(g, {"cases": 0})
        e["cases"] += 1
        if not isinstance(v, dict):
            continue
        for f in ("n", "records"):
            if isinstance(v.get(f), (int, float)):
                e["records"] = e.get("records", 0) + int(v[f])
        for f, agg in (("pos_relerr_max", max), ("scale_relerr_max", max), ("sharp_relerr_max", max), ("desc_maxabs_all", max),
                       ("desc_max", max), ("orient_maxdiff_deg_all", max), ("orientation_deg", max), ("maxabs", max),
                       ("desc_min_cos", min), ("desc_min_cos_same_orient", min), ("orientation_flips", max),
                       ("desc_over_1e-4", max), ("desc_over_1e-3", max), ("desc_bound_checked", max), ("desc_explained", max),
                       ("desc_worst_residual", max), ("allocations_checked", max)):
            if isinstance(v.get(f), (int, float)):
                e[f] = agg(e[f], v[f]) if f in e else v[f]
    out = {"what": "tests -m gpu: parity report summarised per group of cases (worst value of every statistic over the group); "
                   "tools/summarise_parity.py, raw report: gpurun_out/parity_report.json", "cases": len(d), "groups": groups}
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("%d cases -> %d groups" % (len(d), len(groups)))


def pooled(src, dst):
    d = json.load(open(src))
    n = len(d.get("images", []))
    d["images"] = "%d per-image entries dropped (tools/summarise_parity.py); pooled figures above" % n
    json.dump(d, open(dst, "w"), indent=1)
    print("%s: %d images pooled" % (src, n))


if __name__ == "__main__":
    {"parity": parity, "pooled": pooled}[sys.argv[1]](sys.argv[2], sys.argv[3])

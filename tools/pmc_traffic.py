#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, CSV output) of the same bench.py command
into profiles/pmc_traffic.json: HBM bytes per frame per kernel, with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide coalesced reads; WRITE_SIZE is in KB as is).
usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <frames_per_launch> <out.json>"""
import collections, csv, glob, json, sys
fetch_dir, write_dir, frames, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
NAMES = {"lowpass_kernel": "lowpass", "lowpass_down_kernel": "lowpass_down", "scaledown_kernel": "scaledown", "dog_scan_all_kernel": "dog_scan",
         "dog_scan_kernel": "dog_scan", "refine_all_kernel": "refine", "orient_all_kernel": "orient_all",
         "descr_all_kernel": "descr_all", "laplace_kernel": "laplace", "detect_kernel": "detect",
         "match_kernel": "match_mfma"}
def load(d):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        if k in NAMES:
            tot[NAMES[k]] += float(r["Counter_Value"]); n[NAMES[k]] += 1
    return tot, n
ft, fn = load(fetch_dir); wt, wn = load(write_dir)
res = {"frames_per_launch": frames, "bytes_per_frame": {}, "detail": {}}
# launches per step differ per kernel (scaledown: 4); normalise by steps = launches of lowpass
steps = max(1, fn.get("lowpass_down", fn.get("lowpass", fn.get("dog_scan", 1))))     # one prefilter launch per step
for k in sorted(set(ft) | set(wt)):
    rd = 2.0 * ft.get(k, 0.0) * 1024.0 / steps / frames
    wr = wt.get(k, 0.0) * 1024.0 / max(1, wn.get("lowpass_down", wn.get("lowpass", steps))) / frames
    res["bytes_per_frame"][k] = rd + wr
    res["detail"][k] = {"read_bytes_per_frame": rd, "write_bytes_per_frame": wr, "launches": fn.get(k, 0)}
res["note"] = "read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024; separate --pmc passes of the same command"
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res["bytes_per_frame"], indent=1))

"""Budget table of the single-call path from a rocprofv3 --kernel-trace --hip-trace --memory-copy-trace directory
(tools/single_call.sh): per call, every GPU activity (kernel / copy) with its median duration and the median gap in
front of it, the GPU-idle share of a call and the host-side API time.

    python tools/single_call_budget.py <rocprof_dir> <calls>
"""
import csv
import glob
import os
import statistics
import sys


def rows(pattern, root):
    out = []
    for p in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        with open(p, newline="") as f:
            out += list(csv.DictReader(f))
    return out


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "")
    if "<" in name:
        name = name.split("<")[0]
    return name.strip()[-44:]


def main():
    root, calls = sys.argv[1], int(sys.argv[2])
    acts = []
    for r in rows("*kernel_trace.csv", root):
        acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    for r in rows("*memory_copy_trace.csv", root):
        acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")))
    acts.sort()
    api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows("*hip_api_trace.csv", root)]
    api.sort()
    if not acts:
        print("no GPU activity found under", root)
        return
    for marker, label in (("lowpass", "ExtractSift"), ("match_kernel", "MatchSiftData")):
        starts = [i for i, a in enumerate(acts) if marker in a[2] and (i == 0 or marker not in acts[i - 1][2])]
        if marker == "match_kernel":
            # a match call = match_kernel (+ merge ...) ; extraction calls never contain it
            pass
        groups = []
        for gi, s in enumerate(starts):
            e = starts[gi + 1] if gi + 1 < len(starts) else len(acts)
            g = acts[s:e]
            # cut at the first activity of the OTHER kind of call
            other = "match_kernel" if marker == "lowpass" else "lowpass"
            for j, a in enumerate(g):
                if other in a[2]:
                    g = g[:j]
                    break
            groups.append(g)
        groups = [g for g in groups if g]
        if len(groups) < 4:
            continue
        groups = groups[-calls:] if len(groups) > calls else groups
        # the common shape
        shape = statistics.mode(tuple(a[2] for a in g) for g in groups)
        same = [g for g in groups if tuple(a[2] for a in g) == shape]
        period = [same[i + 1][0][0] - same[i][0][0] for i in range(len(same) - 1)
                  if groups.index(same[i + 1]) == groups.index(same[i]) + 1]
        print("== %s: %d calls traced, %d with the common shape of %d GPU activities" % (label, len(groups), len(same), len(shape)))
        print("%-46s %10s %10s" % ("activity", "gap_us", "dur_us"))
        tot_d = tot_g = 0.0
        for k, name in enumerate(shape):
            d = statistics.median((g[k][1] - g[k][0]) / 1e3 for g in same)
            gp = statistics.median((g[k][0] - g[k - 1][1]) / 1e3 for g in same) if k else 0.0
            tot_d += d
            tot_g += max(gp, 0.0) if k else 0.0
            print("%-46s %10.2f %10.2f" % (name, gp, d))
        span = statistics.median((g[-1][1] - g[0][0]) / 1e3 for g in same)
        print("%-46s %10.2f %10.2f" % ("sum (gaps inside a call | busy)", tot_g, tot_d))
        print("GPU span first start -> last end        : %8.2f us" % span)
        if period:
            per = statistics.median(period) / 1e3
            print("call period (start to start)            : %8.2f us" % per)
            print("between calls (host + launch latency)   : %8.2f us" % (per - span))
        # host API time inside one period
        if period and api:
            t0, t1 = same[0][0][0], same[-1][0][0]
            n = len(same) - 1
            by = {}
            for (a, b, fn) in api:
                if a >= t0 and a < t1:
                    by.setdefault(fn, [0, 0.0])
                    by[fn][0] += 1
                    by[fn][1] += (b - a) / 1e3
            print("host API per call (approx., profiler overhead included):")
            for fn, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]:
                print("  %-36s %6.1f calls %8.2f us" % (fn, c / max(n, 1), us / max(n, 1)))
        print()


if __name__ == "__main__":
    main()

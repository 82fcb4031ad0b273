#!/usr/bin/env python3
"""Who runs beside whom: a rocprofv3 --kernel-trace CSV of the pipelined extraction loop (K batches in flight) turned into
the numbers the step budget needs — per kernel the average dispatch duration, the share of wall time it runs, the share of
wall time NOTHING runs, time by number of concurrent dispatches, and for every pair the share of A's time during which B
was running too.
usage: tools/overlap_report.py <kernel_trace.csv> [skip_first_n_steps] [steps]"""
import collections
import csv
import sys

NAMES = {"lowpass_down_kernel": "lpd", "scaledown_kernel": "sd", "dog_scan_all_kernel": "scan", "refine_all_kernel": "refine",
         "orient_all_gather_kernel": "orient", "descr_all_kernel": "descr", "descr_big_kernel": "dbig",
         "bin_detections_kernel": "bin", "frame_shares_kernel": "shares", "export_counts_kernel": "export",
         "scaledown_chain_kernel": "chain", "lowpass_kernel": "lp"}


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
            if k in NAMES:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), NAMES[k]))
    rows.sort()
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    nl = sum(1 for r in rows if r[2] == "lpd")
    # drop the first / last `skip` steps (ramp-up, drain): by lpd launch index
    lpd_starts = [r[0] for r in rows if r[2] == "lpd"]
    t_lo, t_hi = lpd_starts[skip], lpd_starts[nl - skip]
    steps = nl - 2 * skip
    rows = [r for r in rows if r[0] >= t_lo and r[0] < t_hi]
    span = t_hi - t_lo
    print("steps %d  span %.3f ms  -> %.4f ms per step" % (steps, span * 1e-6, span * 1e-6 / steps))
    dur, cnt = collections.defaultdict(float), collections.Counter()
    for a, b, k in rows:
        dur[k] += min(b, t_hi) - a
        cnt[k] += 1
    # sweep line
    ev = []
    for a, b, k in rows:
        ev.append((a, 1, k))
        ev.append((min(b, t_hi), -1, k))
    ev.sort()
    active = collections.Counter()
    by_level = collections.Counter()
    alone = collections.Counter()
    pair = collections.defaultdict(float)
    only_hbm = 0
    last = t_lo
    for t, d, k in ev:
        dt = t - last
        if dt > 0:
            n = sum(active.values())
            by_level[n] += dt
            kinds = [x for x in active if active[x] > 0]
            if kinds and all(x in ("lpd", "sd", "chain", "lp") for x in kinds):
                only_hbm += dt
            if len(kinds) == 1:
                alone[kinds[0]] += dt
            for x in kinds:
                for y in kinds:
                    if x != y:
                        pair[(x, y)] += dt
        active[k] += d
        last = t
    print("kernel      n/step  avg_us   busy_share  alone_share")
    for k in sorted(dur, key=lambda x: -dur[x]):
        print("%-10s %6.2f %8.1f %10.3f %10.3f" % (k, cnt[k] / steps, dur[k] / cnt[k] * 1e-3, dur[k] / span, alone[k] / span))
    print("sum of busy shares %.3f  (> 1 = overlap)" % (sum(dur.values()) / span))
    print("time by concurrent dispatches:", {n: round(v / span, 3) for n, v in sorted(by_level.items())})
    print("only HBM-bound kernels running (lpd / sd): %.3f of the time; nothing running: %.3f" % (only_hbm / span, by_level[0] / span))
    ks = [k for k in sorted(dur, key=lambda x: -dur[x]) if dur[k] / span > 0.02]
    print("share of row's time during which column runs too:")
    print("%-8s" % "" + "".join("%8s" % k for k in ks))
    for a in ks:
        print("%-8s" % a + "".join("%8.2f" % (pair[(a, b)] / dur[a] if a != b else 0) for b in ks))


if __name__ == "__main__":
    main()

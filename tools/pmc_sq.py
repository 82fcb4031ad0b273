#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output directories: per kernel, the sum of every counter over all dispatches.
usage: tools/pmc_sq.py <dir> [<dir> ...] > table.csv"""
import collections, csv, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
names = []
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
            if "at::native" in r["Kernel_Name"] or "elementwise" in k:
                continue
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            tot[k][c] += float(r["Counter_Value"])
print("kernel," + ",".join(names))
for k in sorted(tot):
    print(k + "," + ",".join("%.0f" % tot[k].get(c, 0) for c in names))

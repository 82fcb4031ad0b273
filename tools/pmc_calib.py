#!/usr/bin/env python
"""Run build/pmc_calib under `rocprofv3 --pmc FETCH_SIZE` and derive the FETCH_SIZE -> HBM-bytes factors for the two
access patterns of libmisift.so (wide coalesced reads; scattered 8-byte gathers).  Writes
profiles/r02_pmc_calibration.json, which bench.py uses for the gather kernels' traffic.  GPU box only:
    cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/pmc_calib.py
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = 1 << 30


def main():
    exe = os.path.join(ROOT, "build", "pmc_calib")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "build/pmc_calib"])
    tmp = tempfile.mkdtemp(prefix="pmc_calib_", dir="/tmp")
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    subprocess.check_call([prof, "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", tmp, "-o", "c", "--output-format", "csv",
                           "--", exe], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    f = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    # the two gather launches share a kernel name: they alternate gather64, gather128 in launch order
    stream = rows["calib_stream"]
    g = rows["calib_gather"]
    g64, g128 = g[0::2], g[1::2]
    kb = lambda v: sum(v) / len(v) * 1024.0
    out = {"buffer_bytes": M,
           "raw_FETCH_SIZE_bytes": {"stream_16B_coalesced": kb(stream), "gather_8B_per_64B_sector": kb(g64),
                                    "gather_8B_per_128B_line": kb(g128)},
           "stream_read_factor": M / kb(stream),
           "gather_read_factor": M / kb(g64),
           "gather128_raw_over_gather64_raw": kb(g128) / kb(g64),
           "note": "factor = bytes HBM must deliver / (FETCH_SIZE KB x 1024).  stream: every byte of 1 GiB read once with 16 B "
                   "per lane (the guide's x2 case).  gather: one 8-byte word of every 64-byte sector, scrambled order, each "
                   "sector once -> 1 GiB needed whatever the fetch granularity.  gather128/gather64 raw ratio = 1.0 means a "
                   "gather miss fetches the whole 128-B line, 0.5 means 64-B sectors."}
    dst = os.path.join(ROOT, "gpurun_out", "r02_pmc_calibration.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

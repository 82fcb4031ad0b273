#!/usr/bin/env python3
"""Developer aid: one quick bench.py line per environment / library variant (kernel table included).
   tools/ab.py "X=1" "MISIFT_LIB=build/variants/libmisift_foo.so" ..."""
import json, os, subprocess, sys
Q = "--no-pmc --no-match --no-cpu --no-latency --no-pcie --steps 30 --warmup 5".split()
for v in sys.argv[1:]:
    env = dict(os.environ)
    for kv in v.split():
        k, _, val = kv.partition("=")
        env[k] = val
    p = subprocess.run([sys.executable, "bench.py"] + Q, env=env, capture_output=True, text=True, timeout=600)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        k = {n: round(e["ms_per_step"], 4) for n, e in d["kernels"].items()}
        print("== %s\n%.1f %.4f %s single=%s" % (v, d["value"], d["ms_per_step"], k, d["roofline"].get("single_launch", {}).get("ms")), flush=True)
    except Exception as e:
        print("== %s FAILED %r\n%s" % (v, e, p.stderr[-2000:]), flush=True)

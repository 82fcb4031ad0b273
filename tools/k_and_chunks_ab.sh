#!/bin/bash
# (a) K = 2 against K = 4 batches in flight, alternating; (b) the matcher's chunk plan at the shape ONE rank of
# BASELINE config 5 sweeps (12 500 x 100 000) and at 100 k x 100 k
export TMPDIR=/tmp; mkdir -p gpurun_out/r05_sweep
L="--no-match --no-cpu --no-latency --no-pcie --no-pmc --no-skewed"
for rep in 1 2 3; do
  for K in 2 4 3; do
    timeout 200 python bench.py $L --batches-in-flight $K > gpurun_out/r05_sweep/kab_${K}_$rep.json 2>/dev/null
    python - $K $rep <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05_sweep/kab_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("K=%s rep %s: %8.1f frames/s %.4f ms  no_preroll %.1f" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["no_preroll"]["value"]))
PY
  done
done
for K in 2 4; do
  timeout 200 python bench.py $L --steps 20 --warmup 5 --batches-in-flight $K > gpurun_out/r05_sweep/kab20_${K}.json 2>/dev/null
  python - $K <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05_sweep/kab20_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("steps 20, K=%s: %8.1f frames/s %.4f ms  no_preroll %.1f" % (sys.argv[1], d["value"], d["ms_per_step"], d["no_preroll"]["value"]))
PY
done
python - <<'PY'
import json, os, subprocess, sys
out = {}
for shape, cs in (("12512x100000", [0, 13, 20, 24, 26, 28, 32, 39, 52]), ("100000x100000", [0, 14, 16, 17, 20, 23])):
    rows, n2 = [int(x) for x in shape.split("x")]
    for c in cs:
        env = dict(os.environ)
        if c:
            env["MISIFT_MATCH_CHUNKS"] = str(c)
        r = subprocess.run([sys.executable, "tools/match_chunks.py", "child", str(rows), str(n2)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out["%s c=%d" % (shape, c)] = json.loads(line[-1]) if line else r.stderr[-300:]
        print(shape, "chunks", c or "auto", out["%s c=%d" % (shape, c)], flush=True)
json.dump(out, open("gpurun_out/r05_sweep/match_chunks.json", "w"), indent=1)
PY

export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2 3; do
for cfg in "32:32" "16:32" "12:32" "16:16" "8:32"; do
  sw=${cfg%%:*}; st=${cfg#*:}
  MISIFT_SCAN_WAVES=$sw MISIFT_STRIP_WAVES=$st timeout 300 python bench.py --steps 60 --no-pmc --no-match --no-cpu --no-pcie --no-latency > gpurun_out/r03_sw.json 2> gpurun_out/r03_sw.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_sw.json').read().strip().splitlines()[-1]); print("rep$rep SCAN_WAVES=$sw STRIP_WAVES=$st fps",d["value"])
except Exception as e: print("ERR",e)
PY
done
done

#!/bin/bash
# Work-decomposition knobs at the new default of 2 batches in flight (their optima date from the in-order / K = 4 eras):
# light bench line, one box, default first / in the middle / last as the noise reference
export TMPDIR=/tmp; mkdir -p gpurun_out/r05_sweep
L="--no-match --no-cpu --no-latency --no-pcie --no-pmc --no-skewed"
run() { name=$1; shift
  env "$@" timeout 200 python bench.py $L > gpurun_out/r05_sweep/k2_$name.json 2>/dev/null
  python - "$name" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r05_sweep/k2_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.1f frames/s  %.4f ms  no_preroll %.1f" % (sys.argv[1], d["value"], d["ms_per_step"], d["no_preroll"]["value"]))
except Exception as e:
    print(sys.argv[1], "ERR", repr(e))
PY
}
run default_a A=1
run scan_waves16 MISIFT_SCAN_WAVES=16
run scan_waves24 MISIFT_SCAN_WAVES=24
run scan_waves48 MISIFT_SCAN_WAVES=48
run strip_waves16 MISIFT_STRIP_WAVES=16
run strip_waves48 MISIFT_STRIP_WAVES=48
run default_b A=1
run orient_blocks4 MISIFT_ORIENT_BLOCKS=4
run orient_blocks6 MISIFT_ORIENT_BLOCKS=6
run point_blocks6 MISIFT_POINT_BLOCKS=6
run point_blocks12 MISIFT_POINT_BLOCKS=12
run hwq4 GPU_MAX_HW_QUEUES=4
run hwq16 GPU_MAX_HW_QUEUES=16
run bin0 MISIFT_BIN=0
run default_c A=1

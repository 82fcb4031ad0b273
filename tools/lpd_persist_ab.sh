#!/bin/bash
# MISIFT_LPD_PERSIST (lowpass_down with a capped number of workgroups per CU, each walking over several items) against the
# default one-workgroup-per-item launch, alternating on one box; then the 8-rank driver loops on one GPU (loopback transport)
export TMPDIR=/tmp; mkdir -p gpurun_out/r05_sweep
L="--no-match --no-cpu --no-latency --no-pcie --no-pmc --no-skewed"
for rep in 1 2; do
  for P in 0 2 3 1; do
    MISIFT_LPD_PERSIST=$P timeout 200 python bench.py $L > gpurun_out/r05_sweep/lpdp_${P}_$rep.json 2>/dev/null
    python - $P $rep <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05_sweep/lpdp_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("MISIFT_LPD_PERSIST=%s rep %s: %8.1f frames/s %.4f ms  no_preroll %.1f  lowpass_down %.4f ms (events, in flight)" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["no_preroll"]["value"], d["kernels"]["lowpass_down"]["ms_per_step"]))
PY
  done
done
MISIFT_LPD_PERSIST=2 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "timed_path or batch" 2>&1 | tail -2
timeout 600 python bench.py --emulate-ranks 8 > gpurun_out/r05_emulate_ranks8.json 2> gpurun_out/r05_emulate_ranks8.err; echo "emulate rc=$?"; tail -c 600 gpurun_out/r05_emulate_ranks8.json

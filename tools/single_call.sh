#!/bin/bash
# Single-call path (BASELINE configs 2 and 3) on the GPU box: the reference demo's loop through libcudasift.so,
# plain (wall-clock percentiles) and under rocprofv3 --kernel-trace --hip-trace --memory-copy-trace; the budget table
# (kernel time, inter-launch gaps, count read-back, host time) is written by tools/single_call_budget.py.
#   gpurun -- 'bash tools/single_call.sh <tag>'        -> gpurun_out/<tag>_single_call_*.{json,txt}
tag=${1:-r04}
calls=${2:-200}
export TMPDIR=/tmp; mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from synth import synth_frame
for (w, h) in ((1920, 1080), (1280, 960)):
    for f in (0, 1):
        synth_frame(f, w, h).tofile("/tmp/frame%d_%dx%d.f32" % (f, w, h))
PY
for wh in "1920 1080" "1280 960"; do
  set -- $wh; w=$1; h=$2
  for host in 1 0; do
    GPU_MAX_HW_QUEUES=8 build/single_call /tmp/frame0_${w}x${h}.f32 /tmp/frame1_${w}x${h}.f32 $w $h $calls 5 3.0 $host | grep '^{' >> gpurun_out/${tag}_single_call_wall.jsonl
  done
  (cd /tmp && rm -rf /tmp/sc_$w && timeout 300 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace -d /tmp/sc_$w -o sc --output-format csv -- \
     $R/build/single_call /tmp/frame0_${w}x${h}.f32 /tmp/frame1_${w}x${h}.f32 $w $h $calls 5 3.0 0 > /tmp/sc_$w.out 2>/tmp/sc_$w.err)
  python tools/single_call_budget.py /tmp/sc_$w $calls > gpurun_out/${tag}_single_call_budget_${w}x${h}.txt 2>&1
done
cat gpurun_out/${tag}_single_call_wall.jsonl
cat gpurun_out/${tag}_single_call_budget_1920x1080.txt

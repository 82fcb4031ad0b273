#!/bin/bash
# Developer aid: kernel durations of the single-call path (1920x1080) under rocprofv3 for a list of environment settings.
#   gpurun -- 'bash tools/single_call_sweep.sh tag "A=1 B=2" "C=3" ...'   (one run per argument; "" = defaults;
#   an argument starting with thresh=<t> sets the extraction threshold instead of an environment variable)
tag=$1; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from synth import synth_frame
for f in (0, 1):
    synth_frame(f, 1920, 1080).tofile("/tmp/frame%d_1920x1080.f32" % f)
PY
out=gpurun_out/${tag}_sweep.txt; : > $out
for cfg in "$@"; do
  thresh=3.0
  envs=""
  for kv in $cfg; do
    case $kv in thresh=*) thresh=${kv#thresh=};; *) envs="$envs $kv";; esac
  done
  echo "##### [$cfg]" >> $out
  env GPU_MAX_HW_QUEUES=8 $envs build/single_call /tmp/frame0_1920x1080.f32 /tmp/frame1_1920x1080.f32 1920 1080 200 5 $thresh 0 | grep '^{' >> $out
  (cd /tmp && rm -rf /tmp/sw && env $envs timeout 300 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace -d /tmp/sw -o sw --output-format csv -- \
     $R/build/single_call /tmp/frame0_1920x1080.f32 /tmp/frame1_1920x1080.f32 1920 1080 100 5 $thresh 0 > /dev/null 2>&1)
  python tools/single_call_budget.py /tmp/sw 100 2>&1 | sed -n '1,/^GPU span/p' | grep -v "^==" >> $out
done
cat $out

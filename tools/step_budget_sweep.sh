#!/bin/bash
# where does the step go with K batches in flight?  kernel traces at K = 4 and K = 1 (tools/overlap_report.py) and a
# sweep of launch-structure / residency knobs on the light bench line
export TMPDIR=/tmp; mkdir -p gpurun_out/r05_sweep
cd /tmp
for K in 4 1; do
  BENCH_CHILD_RING=$K BENCH_CHILD_STEPS=60 BENCH_CHILD_PIPELINED=1 BENCH_PMC_FRAMES=64 GPU_MAX_HW_QUEUES=8 timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_$K -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --pmc-child > /dev/null 2>&1
  f=$(find /tmp/tr_$K -name "*kernel_trace.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r05_sweep/trace_K$K.csv
  echo "=== K=$K"; python $GRAFT_REPO_ROOT/tools/overlap_report.py $f 10
done
cd $GRAFT_REPO_ROOT
L="--no-match --no-cpu --no-latency --no-pcie --no-pmc --no-skewed"
run() { # name, env..., -- extra args
  name=$1; shift
  env "$@" timeout 200 python bench.py $L $EXTRA > gpurun_out/r05_sweep/$name.json 2> gpurun_out/r05_sweep/$name.err
  python - "$name" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r05_sweep/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %8.1f frames/s  %.4f ms  no_preroll %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["no_preroll"]["value"]))
except Exception as e:
    print(sys.argv[1], "ERR", repr(e))
PY
}
EXTRA=""
run default_a A=1
run split0 MISIFT_SPLIT_TAIL=0
run split0_chain64 MISIFT_SPLIT_TAIL=0 MISIFT_CHAIN_FRAMES=64
run scan3 MISIFT_LDS_PAD_SCAN=3000
run descr3 MISIFT_LDS_PAD_DESCR=2000
run scan3_descr3 MISIFT_LDS_PAD_SCAN=3000 MISIFT_LDS_PAD_DESCR=2000
run scan2 MISIFT_LDS_PAD_SCAN=16500
run default_b A=1
EXTRA="--batches-in-flight 6"; run K6 A=1
EXTRA="--batches-in-flight 8"; run K8 A=1
EXTRA="--batches-in-flight 1"
run K1_default A=1
run K1_split0 MISIFT_SPLIT_TAIL=0
run K1_split0_chain64 MISIFT_SPLIT_TAIL=0 MISIFT_CHAIN_FRAMES=64
EXTRA="--batches-in-flight 2"; run K2 A=1

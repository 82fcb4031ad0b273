#!/bin/bash
# r03 GPU call 20: the driver's command line on the final kernels, and the paths beside the hot one
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driverlike.json 2> gpurun_out/r03_bench_driverlike.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_driverlike.json').read().strip().splitlines()[-1])
print("driver-like fps", d["value"], "ms", d["ms_per_step"], "K", d["config"].get("batches_in_flight"), "frac", d["roofline"]["frac"], "single", d["roofline"]["single_launch"]["frac"], "traffic", d["roofline"].get("traffic"))
PY
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/cliff.py > gpurun_out/r03_fallback_paths.txt 2>&1; cat gpurun_out/r03_fallback_paths.txt | cut -c1-200

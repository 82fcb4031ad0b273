#!/bin/bash
# r03 GPU call 23 (last): direct-vs-reference tests on the final tree and a third sample of the default bench line
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_refemul.py -q -m gpu > gpurun_out/pytest_gpu23.log 2>&1; grep -aE "passed|failed|error" gpurun_out/pytest_gpu23.log | tail -1
timeout 300 python bench.py > gpurun_out/r03_bench_final_c.json 2> gpurun_out/r03_bench_final_c.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_final_c.json').read().strip().splitlines()[-1])
print("fps", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "single", d["roofline"]["single_launch"]["frac"], "match", d["match"]["roofline"]["frac"], "validated", d["validated_frames"])
PY

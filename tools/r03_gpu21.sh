#!/bin/bash
# r03 GPU call 21: sanity of the last library build (matcher + golden tests) and one more sample of the default bench line
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_multi.py -q -m gpu -x -k "golden or reference_kernels or match" > gpurun_out/pytest_gpu21.log 2>&1; tail -2 gpurun_out/pytest_gpu21.log
timeout 600 python bench.py > gpurun_out/r03_bench_final_b.json 2> gpurun_out/r03_bench_final_b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_final_b.json').read().strip().splitlines()[-1])
print("fps", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "single", d["roofline"]["single_launch"]["frac"], "match", d["match"]["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
PY

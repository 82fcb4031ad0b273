#!/bin/bash
# r05 call 1: the whole GPU suite on the new defaults (MISIFT_BALANCE on, tiny calls accepted, bounded chain wait), then the
# default bench both ways.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r05_pytest_gpu_1.log 2>&1
grep -E "passed|failed|error" gpurun_out/r05_pytest_gpu_1.log | tail -5
for b in 1 0 1 0; do
  MISIFT_BALANCE=$b timeout 300 python bench.py --no-match --no-cpu --no-latency --no-pcie > gpurun_out/r05_balance_bench_${b}_$RANDOM.json 2>gpurun_out/r05_bench_err.txt
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05_balance_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k: round(v["ms_per_step"], 4) for k, v in d["kernels"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 400 python bench.py > gpurun_out/r05_bench_first.json 2> gpurun_out/r05_bench_first.err; tail -c 3000 gpurun_out/r05_bench_first.json

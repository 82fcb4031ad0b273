export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest_a.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_pytest_a.log | tail -3
timeout 600 python bench.py --emulate-ranks 8 > gpurun_out/r03_emulate8.json 2> gpurun_out/r03_emulate8.err; echo "emulate rc=$?"; tail -c 600 gpurun_out/r03_emulate8.json
timeout 900 python bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03_bench_a.json').read().strip().splitlines()[-1])
    print("fps",d["value"],"ms/step",d["ms_per_step"],"frac",d["roofline"]["frac"],"validated",d["validated_frames"])
    print({k:v["ms_per_step"] for k,v in d["kernels"].items()})
    print("match",d["match"]["value"],d["match"]["roofline"]["frac"], "cpu", d["cpu_baseline"])
except Exception as e: print("ERR",e)
PY

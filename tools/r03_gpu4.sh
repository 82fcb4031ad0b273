export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/cliff.py > gpurun_out/r03_fallback_paths.txt 2>&1; cat gpurun_out/r03_fallback_paths.txt
timeout 900 python bench.py > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03_bench_b.json').read().strip().splitlines()[-1])
    print("fps",d["value"],"ms/step",d["ms_per_step"],"frac",d["roofline"]["frac"],"single",d["roofline"].get("single_launch"),"validated",d["validated_frames"])
    print({k:(v["ms_per_step"], v.get("hip_event_ms_per_step")) for k,v in d["kernels"].items()})
    print("pipeline", d["roofline"]["pipeline"])
    print("match",d["match"]["value"],d["match"]["roofline"]["frac"], d["match"].get("cpu_baseline"))
    print("cpu", d["cpu_baseline"])
    print("pcie", d["pcie_inclusive"])
except Exception as e: print("ERR",e); print(open('gpurun_out/r03_bench_b.err').read()[-1500:])
PY

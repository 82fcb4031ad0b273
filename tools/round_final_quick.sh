#!/bin/bash
# The short form of tools/round_final.sh for a late re-verification: the GPU suite, smoke, the default bench line, HIP vs
# the emulated reference at scale (single-frame calls: the small-batch kernels), the single-call budget.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r04_pytest_final.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_pytest_final.log | tail -3
cp gpurun_out/parity_report.json gpurun_out/r04_parity_report_final.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke OK"
timeout 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
SIMT_THREADS=16 HVR_FRAMES=256 timeout 900 python tools/hip_vs_refemul.py > gpurun_out/r04_hip_vs_refemul.log 2>&1; echo "hip_vs_refemul rc=$?"
SIMT_THREADS=16 HVR_VARIANTS=1 timeout 600 python tools/hip_vs_refemul.py > gpurun_out/r04_hip_vs_refemul_variants.log 2>&1; echo "variants rc=$?"
bash tools/single_call.sh r04_final 200 > /dev/null 2>&1; cat gpurun_out/r04_final_single_call_wall.jsonl
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print("fps",d["value"],"ms/step",d["ms_per_step"],"frac",d["roofline"]["frac"],"validated",d["validated_frames"])
print(d["single_frame"])
for f in ("r04_hip_vs_refemul","r04_hip_vs_refemul_variants"):
    try:
        p=json.load(open('gpurun_out/%s.json'%f)); k=[x for x in p if x.startswith("pooled")][0]; print(f, json.dumps(p[k])[:700])
    except Exception as e: print(f,"ERR",e)
PY

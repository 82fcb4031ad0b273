#!/bin/bash
# r03 GPU call 15: dog_scan instruction diet (scalar DoG subtractions, one-instruction |.|max, compare-mask votes)
export TMPDIR=/tmp; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_variants.py -q -m gpu -x -k "not match and not homography" > gpurun_out/pytest_gpu15.log 2>&1; tail -2 gpurun_out/pytest_gpu15.log
bash tools/pmc_pass.sh r03_pmc_sq2 "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > /dev/null 2>&1; grep -E "^kernel|_kernel" gpurun_out/r03_pmc_sq2.csv | grep -v "fft\|rocclr" | cut -d, -f1-4
for rep in 1 2; do
timeout 400 python bench.py --no-match --no-cpu --no-pcie --no-latency --no-pmc 2>/dev/null | tail -1 > gpurun_out/r03_diet_k4_$rep.json
timeout 400 python bench.py --batches-in-flight 1 --no-match --no-cpu --no-pcie --no-latency --no-pmc 2>/dev/null | tail -1 > gpurun_out/r03_diet_k1_$rep.json
python - <<PY
import json
for f in ("r03_diet_k4_$rep","r03_diet_k1_$rep"):
    d=json.load(open("gpurun_out/%s.json"%f)); r=d["roofline"]
    print(f, "fps", d["value"], "frac", r["frac"], "single", (r.get("single_launch") or {}).get("frac"), {k:v.get("ms_per_step") for k,v in d["kernels"].items() if k in ("dog_scan","descr_all","lowpass_down")})
PY
done

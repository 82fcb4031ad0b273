export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/pmc_pass.sh r03_sq_a "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" 2>&1 | tail -14
for sw in 32 24 16; do
  MISIFT_SCAN_WAVES=$sw timeout 300 python bench.py --no-pmc --no-match --no-cpu --no-pcie --no-latency > gpurun_out/r03_sw$sw.json 2> gpurun_out/r03_sw.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_sw$sw.json').read().strip().splitlines()[-1]); print("SCAN_WAVES=$sw fps",d["value"],"ms",d["ms_per_step"])
except Exception as e: print("ERR",e)
PY
done

export TMPDIR=/tmp; mkdir -p gpurun_out
(MISIFT_DESCR_OCC=4 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_j.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_j.log); tail -4 gpurun_out/r02_pytest_j.log
Q="--no-pmc --no-match --no-cpu --no-pcie --no-latency --steps 50 --warmup 5"
for v in "MISIFT_DESCR_OCC=4" "MISIFT_DESCR_OCC=4 MISIFT_POINT_BLOCKS=16" "MISIFT_DESCR_OCC=3 MISIFT_POINT_BLOCKS=16" "MISIFT_TILE_DESCR=0"; do
  echo "== $v"; env $v timeout 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done
MISIFT_DESCR_OCC=4 bash tools/pmc_pass.sh r02_sq_j "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" 2>&1 | grep -E "^kernel|descr|orient|refine"

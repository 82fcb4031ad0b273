export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extract or orient or timed or pipe or findpoints" > gpurun_out/r02_pytest_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_e.log); tail -3 gpurun_out/r02_pytest_e.log
(MISIFT_SCAN=2 MISIFT_DESCR_OCC=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extract or timed or findpoints" > gpurun_out/r02_pytest_e2.log 2>&1; echo "pytest scan2 rc=$?" >> gpurun_out/r02_pytest_e2.log); tail -3 gpurun_out/r02_pytest_e2.log
Q="--no-pmc --no-match --no-cpu --no-pcie --no-latency --steps 50 --warmup 5"
for v in "MISIFT_DESCR_OCC=3" "MISIFT_DESCR_OCC=4" "MISIFT_DESCR_OCC=3 MISIFT_SCAN=1" "MISIFT_DESCR_OCC=3 MISIFT_SCAN=2" "MISIFT_DESCR_OCC=3 MISIFT_SCAN=2 MISIFT_SPLIT_TAIL=0" "MISIFT_DESCR_OCC=3 MISIFT_SCAN=0 MISIFT_SPLIT_TAIL=0" "MISIFT_DESCR_OCC=3 MISIFT_POINT_BLOCKS=6"; do
  echo "== $v"; env $v timeout 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done
MISIFT_SCAN=2 bash tools/pmc_pass.sh r02_sq_e "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" 2>&1 | grep -E "^kernel|descr|orient|refine|dog_scan|lowpass|scaledown|bin_"

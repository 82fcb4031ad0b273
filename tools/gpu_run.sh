export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; tail -c 400 gpurun_out/bench_final.err | grep -v amdgpu.ids
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms'], d['validated_frames'], d['cpu_baseline']['value'], d['match'].get('value'))
print({k:v['ms_per_step'] for k,v in d['kernels'].items()})
r=d['roofline']; print(r['frac'], r.get('single_launch'), r['traffic'], r['hbm_bound_kernels'], r['pipeline'])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu --no-match --no-pcie --no-latency > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
cp /tmp/rp/*kernel_stats.csv gpurun_out/r02_kernel_stats_final.csv 2>/dev/null || find /tmp/rp -name "*stats*" | head
tail -1 /tmp/rp.json > gpurun_out/r02_bench_under_rocprof.json
timeout 600 python bench.py --contexts 4 --no-pmc --no-match --cpu-frames 64 > gpurun_out/bench_ctx4.json 2>/dev/null; echo "ctx4 rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/bench_ctx4.json').read().strip().splitlines()[-1]); print('ctx4', d['value'], d['ms_per_step'], d['validated_frames'], d['roofline']['frac'], d['roofline'].get('alone'))"
bash tools/pmc_pass.sh r02_sq_final "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" > /dev/null 2>&1; echo "pmc rc=$?"

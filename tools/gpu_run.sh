export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_n.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_n.log); tail -4 gpurun_out/r02_pytest_n.log
timeout 600 python bench.py > gpurun_out/r02_bench_n.json 2> gpurun_out/r02_bench_n.err; echo "bench rc=$?"; tail -3 gpurun_out/r02_bench_n.err
python -c "
import json
d=json.load(open('gpurun_out/r02_bench_n.json'))
print(d['value'], d['ms_per_step'], d['validated_frames'], d['step_ms']); print(json.dumps(d['kernels'])); print(json.dumps(d['roofline'])[:1500]); print(d['cpu_baseline']); print(d['match']['value'], d['match']['roofline']['frac'], d['match'].get('validated_rows'))"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu --no-pcie --no-latency --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/prof_r02.log 2>&1); ls gpurun_out/prof_r02 | head; head -30 gpurun_out/prof_r02/*kernel_stats.csv 2>/dev/null | cut -c1-200
bash tools/pmc_pass.sh r02_pmc_traffic "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -E "^kernel|descr|orient|refine|dog_scan|lowpass|scaledown|bin_"

export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/bench_final.err | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms']['p10'], d['step_ms']['p50'], d['step_ms']['p90'], d['validated_frames'], d['cpu_baseline']['value'], d['match'].get('value'))
print({k:(v['ms_per_step'], v.get('hip_event_ms_per_step')) for k,v in d['kernels'].items()})
r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r.get('single_launch'), r['hbm_bound_kernels'], r['pipeline'].get('traffic_frac'), r['copy_ceiling_GBps'])
print(d['pcie_inclusive'], d['single_frame'])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-match --no-pcie --no-latency > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
cp /tmp/rp/*kernel_stats.csv gpurun_out/r02_kernel_stats_final.csv
tail -1 /tmp/rp.json > gpurun_out/r02_bench_under_rocprof.json
python - <<'PY'
import json,csv
d=json.loads(open('gpurun_out/r02_bench_under_rocprof.json').read())
print(d['value'], {k:(v['ms_per_step'], v['launches_per_step']) for k,v in d['kernels'].items()}, d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for r in csv.DictReader(open('gpurun_out/r02_kernel_stats_final.csv')):
    n=r['Name']
    if any(k in n for k in ('dog_scan','descr_all','lowpass_down','orient_all','refine_all','scaledown')) and 'native' not in n:
        print(n[:40], r['Calls'], float(r['AverageNs'])/1e6)
PY
timeout 600 python bench.py --contexts 4 --no-pmc --no-match --cpu-frames 64 > gpurun_out/bench_ctx4.json 2>/dev/null; echo "ctx4 rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/bench_ctx4.json').read().strip().splitlines()[-1]); print('ctx4', d['value'], d['ms_per_step'], d['validated_frames'], d['roofline']['frac'], d['roofline'].get('alone'))"

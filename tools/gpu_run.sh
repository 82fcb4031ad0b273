export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/bench_final.err | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms']['p10'], d['step_ms']['p50'], d['step_ms']['p90'], d['validated_frames'], d['cpu_baseline']['value'], d['match'].get('value'))
print({k:(v['ms_per_step'], v.get('hip_event_ms_per_step')) for k,v in d['kernels'].items()})
r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r.get('single_launch'), r['hbm_bound_kernels'], r['pipeline'].get('traffic_frac'), r['copy_ceiling_GBps'])
print(d['pcie_inclusive']['frames_per_s_u8'], d['pcie_inclusive']['frames_per_s_f32'], d['single_frame']['extract_1920x1080_ms'])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-match --no-pcie --no-latency > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
cp /tmp/rp/*kernel_stats.csv gpurun_out/r02_kernel_stats_final.csv
tail -1 /tmp/rp.json > gpurun_out/r02_bench_under_rocprof.json

export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py --no-pmc --no-match --cpu-frames 8 > gpurun_out/bench_k4.json 2> gpurun_out/bench_k4.err; echo rc=$?
tail -3 gpurun_out/bench_k4.err | grep -v amdgpu.ids
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_k4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms'], d['validated_frames'])
print({k:(v['ms_per_step'], v.get('alone_ms_per_step')) for k,v in d['kernels'].items()})
r=d['roofline']; print(r['frac'], r.get('alone'), r.get('single_launch'), r['hbm_bound_kernels'])
PY
timeout 300 python bench.py --no-pmc --no-match --no-cpu --no-latency --no-pcie --selftest-dist --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('selftest-dist', d['value'], d['ms_per_step'])"

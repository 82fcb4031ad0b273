export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/pytest_multi.log 2>&1; grep -E "passed|failed|error|Error" gpurun_out/pytest_multi.log | tail -3
for m in "--selftest-dist" "--selftest-dist --contexts 4"; do
BENCH_TRACE=1 timeout 300 python bench.py --no-pmc --no-match --no-cpu --no-latency --steps 100 --warmup 10 $m 2> gpurun_out/t.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], d['pcie_inclusive']['frames_per_s_u8'], d['pcie_inclusive']['frames_per_s_f32'])"
grep "host time" gpurun_out/t.err
done

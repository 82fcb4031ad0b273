export TMPDIR=/tmp; mkdir -p gpurun_out
for e in "BENCH_RB_PRIO=-1" "BENCH_RB_PRIO=0" "BENCH_RB_PRIO=-1" "BENCH_RB_PRIO=0"; do
env $e timeout 300 python bench.py --no-pmc --no-match --no-cpu --no-latency --no-pcie 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e ctx1', d['value'], d['ms_per_step'], d['step_ms']['p50'])"
done
for e in "BENCH_RB_PRIO=0"; do
env $e timeout 300 python bench.py --no-pmc --no-match --no-cpu --no-latency --no-pcie --selftest-dist 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e selftest-dist', d['value'], d['ms_per_step'])"
done

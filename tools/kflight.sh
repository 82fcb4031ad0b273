export TMPDIR=/tmp
for cfg in "4 8" "6 8" "8 8" "6 16" "4 16" "3 8"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py --steps 100 --warmup 10 --batches-in-flight $1 --no-cpu --no-match --no-pcie --no-pmc --no-latency 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$1 queues=$2 fps', d['value'], 'ms', d['ms_per_step'], d['step_ms']['p50'])"
done

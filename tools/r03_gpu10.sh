export TMPDIR=/tmp
for rep in 1 2; do
for k in 1 2 4; do
  timeout 300 python bench.py --steps 20 --warmup 5 --batches-in-flight $k --no-pmc --no-match --no-cpu --no-pcie --no-latency 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rep$rep K=$k steps=20 fps', d['value'], 'ms', d['ms_per_step'])"
done
done

export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/ab.py "X=1" "GPU_MAX_HW_QUEUES=8" "X=2" "GPU_MAX_HW_QUEUES=8 X=2"
for e in "GPU_MAX_HW_QUEUES=8"; do
env $e timeout 300 python bench.py --no-pmc --no-match --no-cpu --no-latency --no-pcie --steps 100 --warmup 10 --contexts 4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e ctx4', d['value'], d['ms_per_step'])"
done

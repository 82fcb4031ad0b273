#!/bin/bash
# r03 GPU call 22: batches in flight for the driver's short command line (20 timed steps) on the final kernels
mkdir -p gpurun_out
for rep in 1 2 3; do
for k in 2 3 4; do
  timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --batches-in-flight $k --no-pmc --no-match --no-cpu --no-pcie --no-latency 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rep$rep K=$k steps=20 fps', d['value'], 'ms', d['ms_per_step'])"
done
done

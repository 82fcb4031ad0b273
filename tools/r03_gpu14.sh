#!/bin/bash
# r03 GPU call 14: (a) the column cut of the sharded matcher with the final chunk plan; (b) ring K=4 with and without the
# split-tail scan (one dog_scan launch per batch instead of two)
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 300 python tools/match_split.py gpurun_out/r03_match_split.json 2>&1 | grep rows | cut -c1-330
for rep in 1 2; do
for st in 8 0; do
  MISIFT_SPLIT_TAIL=$st timeout 400 python bench.py --no-match --no-cpu --no-pcie --no-latency --no-pmc 2>/dev/null | tail -1 > gpurun_out/r03_split_tail_${st}_$rep.json
  python -c "
import json; d=json.load(open('gpurun_out/r03_split_tail_${st}_$rep.json')); r=d['roofline']
print('rep $rep split_tail=$st fps', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'single', (r.get('single_launch') or {}).get('frac'), 'K', d['config'].get('batches_in_flight'))"
done
done
timeout 600 python -m pytest tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -4

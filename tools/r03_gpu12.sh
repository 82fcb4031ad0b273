#!/bin/bash
# r03 GPU call 12: chunk plan of the matcher at the row counts of an 8/4/2-GPU job (and the full 100k x 100k)
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for cfg in "24 2" "12 4" "8 8" "6 8" "4 16" "3 32"; do
  set -- $cfg
  echo "== rounds $1 min_tiles $2"
  MISIFT_MATCH_ROUNDS=$1 MISIFT_MATCH_MIN_TILES=$2 timeout 300 python tools/match_split.py gpurun_out/r03_match_split_$1_$2.json 2>&1 | grep rows | cut -c1-400
  MISIFT_MATCH_ROUNDS=$1 MISIFT_MATCH_MIN_TILES=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu --no-pcie --no-latency 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get("matcher",{}); print("100k x 100k:", {k:m[k] for k in ("match_ms","call_ms","mfma_util_fp32") if k in m})"
done
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x -k "loopback or match" > gpurun_out/pytest_gpu12.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu12.log
tail -3 gpurun_out/pytest_gpu12.log

#!/bin/bash
# r03 GPU call 11: the overlapped sharded matcher (loopback worlds) + the cost of its column cut on one GPU
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x -k "loopback or match" > gpurun_out/pytest_gpu11.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu11.log
tail -5 gpurun_out/pytest_gpu11.log
timeout 300 python tools/match_split.py gpurun_out/r03_match_split.json 2>&1 | tail -5

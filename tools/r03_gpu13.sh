#!/bin/bash
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x -k "loopback or match" > gpurun_out/pytest_gpu13.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu13.log
tail -3 gpurun_out/pytest_gpu13.log
OUT=gpurun_out/r03_match_chunks2.json timeout 900 python tools/match_chunks.py 2>&1 | grep chunks

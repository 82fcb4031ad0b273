#!/bin/bash
# r03 GPU call 19: orientation histogram with masked adds: focused parity subset + SQ counters + timing
export TMPDIR=/tmp; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -q -m gpu -x -k "golden or reference_kernels or stereo or timed_path or determin or 1080" > gpurun_out/pytest_gpu19.log 2>&1; tail -2 gpurun_out/pytest_gpu19.log
bash tools/pmc_pass.sh r03_pmc_sq6 "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > /dev/null 2>&1; grep -E "^kernel|_kernel" gpurun_out/r03_pmc_sq6.csv | grep -v "fft\|rocclr" | cut -d, -f1-4
timeout 400 python bench.py --batches-in-flight 1 --no-match --no-cpu --no-pcie --no-latency --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=1 fps', d['value'], {k:v.get('ms_per_step') for k,v in d['kernels'].items()})"

#!/bin/bash
# r03 GPU call 18: footprint weights from an LDS table: parity + SQ counters
export TMPDIR=/tmp; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x -k "not match and not homography" > gpurun_out/pytest_gpu18.log 2>&1; tail -2 gpurun_out/pytest_gpu16.log
bash tools/pmc_pass.sh r03_pmc_sq5 "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > /dev/null 2>&1; grep -E "^kernel|_kernel" gpurun_out/r03_pmc_sq5.csv | grep -v "fft\|rocclr" | cut -d, -f1-4

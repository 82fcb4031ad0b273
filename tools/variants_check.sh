#!/bin/bash
# The parity suite under the path-variant switches (dense kernels, descriptor from global memory, hipGraph replay,
# orientation from an LDS window): gpurun -- 'bash tools/variants_check.sh'
export TMPDIR=/tmp; mkdir -p gpurun_out
for e in "MISIFT_FUSED=0" "MISIFT_TILE_DESCR=0" "MISIFT_GRAPH=1" "MISIFT_TILE_ORIENT=1"; do
env $e timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_var.log 2>&1; echo "$e: $(grep -E 'passed|failed|error' gpurun_out/pytest_var.log | tail -1)"
done

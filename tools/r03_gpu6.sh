export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multi.py -q -m gpu -x -k "match" 2>&1 | tail -3
for i in 1 2 3; do python tools/match_prof.py; done

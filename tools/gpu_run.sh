export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/ab.py "X=1" "X=2"

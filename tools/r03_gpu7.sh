export TMPDIR=/tmp
for v in "" wg8 "" wg8; do
  if [ -n "$v" ]; then export MISIFT_LIB=$PWD/build/variants/libmisift_$v.so; else unset MISIFT_LIB; fi
  echo -n "${v:-default}: "; MATCH_REPS=8 python tools/match_prof.py | tail -1
done
export MISIFT_LIB=$PWD/build/variants/libmisift_wg8.so
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x -k match 2>&1 | tail -1

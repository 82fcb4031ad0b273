export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x -k "lowpass or scaledown or laplace or findpoints or ragged or golden or stereo or timed or odd" 2>&1 | tail -3
for k in 1 4; do
  timeout 300 python bench.py --batches-in-flight $k --no-pmc --no-match --no-cpu --no-pcie --no-latency > gpurun_out/r03_ab_k$k.json 2> gpurun_out/r03_ab.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_ab_k$k.json').read().strip().splitlines()[-1]); print("K=$k fps",d["value"],"ms",d["ms_per_step"], {n:v["ms_per_step"] for n,v in d["kernels"].items()})
except Exception as e: print("K=$k ERR",e); print(open('gpurun_out/r03_ab.err').read()[-600:])
PY
done

export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batches_in_flight or three_contexts" 2>&1 | tail -5
for k in 1 2 3 4; do
  for q in 8 16; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --batches-in-flight $k --no-pmc --no-match --no-cpu --no-pcie --no-latency > gpurun_out/r03_ring_k${k}_q${q}.json 2> gpurun_out/r03_ring.err
    python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_ring_k${k}_q${q}.json').read().strip().splitlines()[-1]); print("K=$k Q=$q fps",d["value"],"ms",d["ms_per_step"], d["step_ms"] and d["step_ms"]["p50"])
except Exception as e: print("K=$k Q=$q ERR",e); print(open('gpurun_out/r03_ring.err').read()[-800:])
PY
  done
done

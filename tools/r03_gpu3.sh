export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "k1:--batches-in-flight 1:8" "ring2_q8:--batches-in-flight 2:8" "ring3_q8:--batches-in-flight 3:8" "ring4_q8:--batches-in-flight 4:8" "ring4_qdef:--batches-in-flight 4:" "ring4_q16:--batches-in-flight 4:16" "ring3_q16:--batches-in-flight 3:16" "ring4_q24:--batches-in-flight 4:24" "ring5_q16:--batches-in-flight 5:16" "ctx4_qdef:--contexts 4:"; do
  name=${cfg%%:*}; rest=${cfg#*:}; args=${rest%%:*}; q=${rest#*:}
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; unset BENCH_NO_QUEUE_DEFAULT; else unset GPU_MAX_HW_QUEUES; export BENCH_NO_QUEUE_DEFAULT=1; fi
  timeout 300 python bench.py $args --no-pmc --no-match --no-cpu --no-pcie --no-latency > gpurun_out/r03_$name.json 2> gpurun_out/r03_ring.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_$name.json').read().strip().splitlines()[-1]); print("$name fps",d["value"],"ms",d["ms_per_step"])
except Exception as e: print("$name ERR",e); print(open('gpurun_out/r03_ring.err').read()[-600:])
PY
done

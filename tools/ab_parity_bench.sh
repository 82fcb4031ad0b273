#!/bin/bash
# Developer aid: for each libmisift.so build — a short parity run (the scan-facing GPU tests) and, alternating over two
# repetitions, the default batch bench.  gpurun -- 'bash tools/ab_parity_bench.sh tag "" build/variants/libmisift_x.so ...'
export MISIFT_TUNABLES=1      # the library reads its launch-shape / path variables only under this switch
tag=$1; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt; : > $out
for lib in "$@"; do
  MISIFT_LIB=$lib timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
    -k "findpoints or extract_synthetic or extract_stereo_left or batch_equals_single or ragged or small_batches or packed_async_equals" \
    > /tmp/par.log 2>&1
  echo "parity ${lib:-(in-tree)}: $(tail -1 /tmp/par.log)" >> $out
done
for rep in 1 2; do
  for lib in "$@"; do
    MISIFT_LIB=$lib timeout 600 python bench.py --no-cpu --no-match --no-pcie --no-latency --no-pmc --steps 100 --warmup 20 > /tmp/ab.json 2>/tmp/ab.err
    python - "$lib" >> $out <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
    print("%-44s fps %8.0f  ms/step %.4f  " % (sys.argv[1] or "(in-tree)", d["value"], d["ms_per_step"]) +
          " ".join("%s=%.3f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()) +
          " single=%s" % d["roofline"].get("single_launch", {}).get("ms"))
except Exception as e:
    print("%-44s FAILED %r %s" % (sys.argv[1], e, open('/tmp/ab.err').read()[-600:]))
PY
  done
done
cat $out

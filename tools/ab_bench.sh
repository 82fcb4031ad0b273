#!/bin/bash
# Developer aid: A/B the default batch bench between libmisift.so builds on ONE box, alternating (clock / box drift).
#   gpurun -- 'bash tools/ab_bench.sh tag "" build/variants/libmisift_x.so ...'   ("" = the in-tree library)
export MISIFT_TUNABLES=1      # the library reads its launch-shape / path variables only under this switch
tag=$1; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt; : > $out
for rep in 1 2; do
  for lib in "$@"; do
    MISIFT_LIB=$lib timeout 600 python bench.py --no-cpu --no-match --no-pcie --no-latency --no-pmc --steps 100 --warmup 20 > /tmp/ab.json 2>/tmp/ab.err
    python - "$lib" >> $out <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print("%-44s fps %8.0f  ms/step %.4f  " % (sys.argv[1] or "(in-tree)", d["value"], d["ms_per_step"]) +
      " ".join("%s=%.3f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
PY
  done
done
cat $out

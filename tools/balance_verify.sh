#!/bin/bash
# What had to be green before MISIFT_BALANCE became the default (r05: profiles/r05_balance_default_ab.txt): the whole GPU suite with
# every context of the session balanced, the A/B of tools/balance_ab.py, and the default bench line both ways.
#   gpurun --timeout 1500 -- 'bash tools/balance_verify.sh'   -> gpurun_out/balance_*.{log,txt,json}
export MISIFT_TUNABLES=1      # the library reads its launch-shape / path variables only under this switch
export TMPDIR=/tmp; mkdir -p gpurun_out
MISIFT_BALANCE=1 timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/balance_pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/balance_pytest_gpu.log | tail -3
timeout 120 python tools/balance_ab.py > gpurun_out/balance_ab.txt 2>&1; cat gpurun_out/balance_ab.txt
for b in 0 1 0 1; do
  MISIFT_BALANCE=$b timeout 300 python bench.py --no-match --no-cpu --no-latency --no-pcie > gpurun_out/balance_bench_$b.json 2>/dev/null
  python - "$b" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/balance_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("MISIFT_BALANCE=%s" % sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done

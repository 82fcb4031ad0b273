#!/bin/bash
# r05 call 2: whole GPU suite (no -x) on the new defaults, then the refactored bench (default line + the driver's command)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1100 python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu_2.log 2>&1
grep -E "passed|failed|error" gpurun_out/r05_pytest_gpu_2.log | tail -5
grep -E "^FAILED|^ERROR" gpurun_out/r05_pytest_gpu_2.log | head -40
timeout 500 python bench.py > gpurun_out/r05_bench_refactor.json 2> gpurun_out/r05_bench_refactor.err; echo "bench rc $?"; tail -5 gpurun_out/r05_bench_refactor.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc > gpurun_out/r05_bench_driverlike_a.json 2> gpurun_out/r05_bench_driverlike_a.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("r05_bench_refactor", "r05_bench_driverlike_a"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "no_preroll", d["no_preroll"], "frac", r["frac"], r.get("frac_basis"), "summed", r.get("summed", {}).get("frac"),
              "single", r.get("single_launch", {}).get("frac"), "hbm", {k: r["hbm"].get(k) for k in ("traffic_frac", "floor_frac", "traffic_over_floor")},
              "issue", {k: r["issue"].get(k) for k in ("issue_frac_of_step", "valu_active_frac_of_step", "clock_GHz", "insts_per_step")},
              "skewed", d["skewed_batch"], "stages", d["stages_s"])
        print({k: (v["ms_per_step"], v.get("union_ms_per_step")) for k, v in d["kernels"].items()})
        print("match", d["match"] and (d["match"]["value"], d["match"]["roofline"]["frac"], d["match"].get("rank_shard_12500x100000")))
    except Exception as e:
        print(f, "ERR", repr(e))
PY

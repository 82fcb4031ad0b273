#!/bin/bash
# Last confirmation of the round's final tree: the whole GPU suite, smoke, the default line + the driver's command + the in-order
# context, the 1-rank RCCL self-test of the N > 1 code path.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu_confirm.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05_pytest_gpu_confirm.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r05_pytest_gpu_confirm.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke OK"
timeout 900 python bench.py > gpurun_out/r05_bench_confirm.json 2> gpurun_out/r05_bench_confirm.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc > gpurun_out/r05_bench_confirm_driverlike.json 2>/dev/null; echo "driver-like rc=$?"
timeout 300 python bench.py --batches-in-flight 1 --no-match --no-pcie --no-latency --no-cpu --no-pmc --no-skewed > gpurun_out/r05_bench_confirm_inorder.json 2>/dev/null; echo "K=1 rc=$?"
timeout 300 python bench.py --selftest-dist --steps 20 --warmup 5 --no-pmc --no-match --no-pcie --no-latency > gpurun_out/r05_selftest_dist.json 2> gpurun_out/r05_selftest_dist.err; echo "selftest-dist rc=$?"
python - <<'PY'
import json
for f in ("r05_bench_confirm", "r05_bench_confirm_driverlike", "r05_bench_confirm_inorder", "r05_selftest_dist"):
    try:
        d = json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "fps", d["value"], "ms/step", d["ms_per_step"], "no_preroll", d["no_preroll"] and d["no_preroll"]["value"], "frac", r["frac"], "hbm", r["hbm"].get("traffic_frac"),
              "issue", r["issue"].get("valu_active_frac_of_step"), "validated", d["validated_frames"], "rccl_ranks", d["rccl_ranks"])
        print("  ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
# The verification round 5's numbers come from (GPU box: gpurun -- 'bash tools/r05_final.sh'): the whole GPU suite, smoke, the
# default bench line, the driver's command line, the in-order context under rocprofv3 --kernel-trace --stats, the HIP path
# against the emulated reference at scale (with the per-record descriptor bound), the single-call budget.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05_pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r05_pytest_gpu.log | head
cp gpurun_out/parity_report.json gpurun_out/r05_parity_report_raw.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke OK"
timeout 900 python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driverlike.json 2> gpurun_out/r05_bench_driverlike.err; echo "bench driver-like rc=$?"
timeout 600 python bench.py --batches-in-flight 1 --no-match --no-pcie --no-latency --cpu-frames 64 > gpurun_out/r05_bench_inorder.json 2> gpurun_out/r05_bench_inorder.err; echo "bench K=1 rc=$?"
(cd /tmp && rm -rf /tmp/rp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batches-in-flight 1 --no-cpu --no-match --no-pcie --no-latency --no-pmc --no-skewed > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
find /tmp/rp -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_kernel_stats.csv \;
tail -1 /tmp/rp.json > gpurun_out/r05_bench_under_rocprof.json
bash tools/pmc_pass.sh r05_pmc_sq "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" > /dev/null 2>&1; grep -E "^kernel|_kernel" gpurun_out/r05_pmc_sq.csv | grep -v "fft\|rocclr" | cut -d, -f1-5
SIMT_THREADS=16 HVR_FRAMES=256 timeout 900 python tools/hip_vs_refemul.py > gpurun_out/r05_hip_vs_refemul.log 2>&1; echo "hip_vs_refemul rc=$?"
SIMT_THREADS=16 HVR_VARIANTS=1 timeout 600 python tools/hip_vs_refemul.py > gpurun_out/r05_hip_vs_refemul_variants.log 2>&1; echo "variants rc=$?"
bash tools/single_call.sh r05 200 > /dev/null 2>&1; cat gpurun_out/r05_single_call_wall.jsonl
python - <<'PY'
import json
for f in ("r05_bench_final", "r05_bench_driverlike", "r05_bench_inorder"):
    try:
        d = json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "fps", d["value"], "ms/step", d["ms_per_step"], "no_preroll", d["no_preroll"] and d["no_preroll"]["value"], "frac", r["frac"], "summed", r.get("summed", {}).get("frac"),
              "single", (r.get("single_launch") or {}).get("frac"), "hbm", {k: r["hbm"].get(k) for k in ("traffic_frac", "floor_frac", "traffic_over_floor")},
              "issue", {k: r["issue"].get(k) for k in ("issue_frac_of_step", "valu_active_frac_of_step", "clock_GHz", "clock_raw_GHz")}, "validated", d["validated_frames"])
        print("  ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
        if d.get("match"): print("  match", d["match"]["value"], d["match"]["roofline"]["frac"], d["match"].get("rank_shard_12500x100000"))
        print("  cpu", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]), "skewed", d.get("skewed_batch") and d["skewed_batch"]["ratio_to_uniform"])
        print("  single", d.get("single_frame") and {k: v for k, v in d["single_frame"].items() if k.endswith("_ms")})
    except Exception as e: print(f, "ERR", e)
for f in ("r05_hip_vs_refemul", "r05_hip_vs_refemul_variants"):
    try:
        p = json.load(open('gpurun_out/%s.json' % f)); k = [x for x in p if x.startswith("pooled")][0]; print(f, json.dumps(p[k])[:900])
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
# The verification a round's numbers come from, for any round tag:   gpurun -- 'bash tools/round.sh r06 [quick]'
#   GPU test-suite + smoke | the default bench line, the driver's command line, the in-order context | the same under
#   rocprofv3 --kernel-trace --stats and the SQ counter passes | the matcher alone (plain + under rocprofv3) | the HIP path
#   against the emulated reference at scale (with the per-record descriptor explanation) | the single-call budget |
#   the 1-rank RCCL self-test and 8 emulated ranks.   `quick`: suite, smoke, bench lines, HIP vs reference only.
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:?usage: tools/round.sh <tag> [quick]}; quick=$2
export TMPDIR=/tmp; mkdir -p gpurun_out
root=${GRAFT_REPO_ROOT:-$(pwd)}
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/${tag}_pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/${tag}_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${tag}_pytest_gpu.log | head
echo "suite wall seconds: $(( $(date +%s) - t0 ))"
cp gpurun_out/parity_report.json gpurun_out/${tag}_parity_report_raw.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke OK"
timeout 900 python bench.py > gpurun_out/${tag}_bench_final.json 2> gpurun_out/${tag}_bench_final.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driverlike.json 2> gpurun_out/${tag}_bench_driverlike.err; echo "bench driver-like rc=$?"
SIMT_THREADS=16 HVR_TAG=$tag HVR_FRAMES=256 timeout 1200 python tools/hip_vs_refemul.py > gpurun_out/${tag}_hip_vs_refemul.log 2>&1; echo "hip_vs_refemul rc=$?"
SIMT_THREADS=16 HVR_TAG=$tag HVR_VARIANTS=1 timeout 900 python tools/hip_vs_refemul.py > gpurun_out/${tag}_hip_vs_refemul_variants.log 2>&1; echo "variants rc=$?"
if [ -z "$quick" ]; then
  timeout 600 python bench.py --batches-in-flight 1 --no-match --no-pcie --no-latency --cpu-frames 64 > gpurun_out/${tag}_bench_inorder.json 2> gpurun_out/${tag}_bench_inorder.err; echo "bench K=1 rc=$?"
  (cd /tmp && rm -rf /tmp/rp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $root/bench.py --batches-in-flight 1 --no-cpu --no-match --no-pcie --no-latency --no-pmc --no-skewed > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
  find /tmp/rp -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
  tail -1 /tmp/rp.json > gpurun_out/${tag}_bench_under_rocprof.json
  bash tools/pmc_pass.sh ${tag}_pmc_sq "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" > /dev/null 2>&1
  grep -E "^kernel|_kernel" gpurun_out/${tag}_pmc_sq.csv | grep -v "fft\|rocclr" | cut -d, -f1-5
  MATCH_REPS=8 python tools/match_prof.py | tail -1 | tee gpurun_out/${tag}_match_plain.txt
  (cd /tmp && rm -rf /tmp/mt && rocprofv3 --kernel-trace --stats -d /tmp/mt -o m --output-format csv -- python $root/tools/match_prof.py > /dev/null 2>&1); find /tmp/mt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_match_kernel_stats.csv \;
  head -3 gpurun_out/${tag}_match_kernel_stats.csv
  (cd /tmp && rm -rf /tmp/mp && MATCH_REPS=2 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d /tmp/mp -o m --output-format csv -- python $root/tools/match_prof.py > /dev/null 2>&1)
  python tools/pmc_sq.py /tmp/mp > gpurun_out/${tag}_match_pmc.csv; cat gpurun_out/${tag}_match_pmc.csv | grep -v rocclr
  bash tools/single_call.sh $tag 200 > /dev/null 2>&1; cat gpurun_out/${tag}_single_call_wall.jsonl
  timeout 300 python bench.py --selftest-dist --steps 20 --warmup 5 --no-pmc --no-match --no-pcie --no-latency > gpurun_out/${tag}_selftest_dist.json 2> gpurun_out/${tag}_selftest_dist.err; echo "selftest-dist rc=$?"
  timeout 600 python bench.py --emulate-ranks 8 > gpurun_out/${tag}_emulate_ranks8.json 2> gpurun_out/${tag}_emulate_ranks8.err; echo "emulate rc=$?"
fi
python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench_final", "bench_driverlike", "bench_inorder", "selftest_dist"):
    try:
        d = json.loads(open('gpurun_out/%s_%s.json' % (tag, f)).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "fps", d["value"], "ms/step", d["ms_per_step"], "no_preroll", d["no_preroll"] and d["no_preroll"]["value"], "frac", r["frac"],
              "single", (r.get("single_launch") or {}).get("frac"), "hbm", {k: r["hbm"].get(k) for k in ("traffic_frac", "floor_frac", "traffic_over_floor")},
              "issue", {k: r["issue"].get(k) for k in ("issue_frac_of_step", "valu_active_frac_of_step", "clock_GHz")}, "validated", d["validated_frames"])
        print("  ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
        if d.get("match"): print("  match", d["match"]["value"], d["match"]["roofline"]["frac"], d["match"].get("rank_shard_12500x100000"))
        print("  cpu", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]), "skewed", d.get("skewed_batch") and d["skewed_batch"]["ratio_to_uniform"])
        print("  single", d.get("single_frame") and {k: v for k, v in d["single_frame"].items() if k.endswith("_ms")})
        print("  T1/T2/T3", json.dumps(d.get("timing_definitions"))[:600]); print("  config2", d.get("config2_1280x960"))
    except Exception as e: print(f, "ERR", e)
for f in ("hip_vs_refemul", "hip_vs_refemul_variants"):
    try:
        p = json.load(open('gpurun_out/%s_%s.json' % (tag, f))); k = [x for x in p if x.startswith("pooled")][0]; print(f, json.dumps(p[k])[:1200])
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
# The verification a round's numbers come from (GPU box: gpurun -- 'bash tools/round_final.sh'):
# GPU test-suite, smoke, the default bench line, the same command under rocprofv3 --kernel-trace --stats (in-order
# context, so that per-kernel averages mean something), SQ instruction counters, the 8 emulated ranks, the matcher alone.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r04_pytest_final.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_pytest_final.log | tail -3
cp gpurun_out/parity_report.json gpurun_out/r04_parity_report_final.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke OK"
timeout 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pmc > gpurun_out/r04_bench_driverlike.json 2> gpurun_out/r04_bench_driverlike.err; echo "bench driver-like rc=$?"
timeout 600 python bench.py --batches-in-flight 1 --no-match --no-pcie --no-latency --cpu-frames 64 > gpurun_out/r04_bench_inorder.json 2> gpurun_out/r04_bench_inorder.err; echo "bench K=1 rc=$?"
(cd /tmp && rm -rf /tmp/rp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batches-in-flight 1 --no-cpu --no-match --no-pcie --no-latency --no-pmc > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
find /tmp/rp -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_kernel_stats.csv \;
tail -1 /tmp/rp.json > gpurun_out/r04_bench_under_rocprof.json
bash tools/pmc_pass.sh r04_pmc_sq "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" > /dev/null 2>&1; grep -E "^kernel|_kernel" gpurun_out/r04_pmc_sq.csv | grep -v "fft\|rocclr" | cut -d, -f1-4
timeout 600 python bench.py --emulate-ranks 8 > gpurun_out/r04_emulate8.json 2> gpurun_out/r04_emulate8.err; echo "emulate rc=$?"
MATCH_REPS=8 python tools/match_prof.py | tail -1 | tee gpurun_out/r04_match_plain.txt
(cd /tmp && rm -rf /tmp/mt && rocprofv3 --kernel-trace --stats -d /tmp/mt -o m --output-format csv -- python $GRAFT_REPO_ROOT/tools/match_prof.py > /dev/null 2>&1); find /tmp/mt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_match_kernel_stats.csv \;
head -3 gpurun_out/r04_match_kernel_stats.csv
SIMT_THREADS=16 HVR_FRAMES=256 timeout 900 python tools/hip_vs_refemul.py > gpurun_out/r04_hip_vs_refemul.log 2>&1; echo "hip_vs_refemul rc=$?"
SIMT_THREADS=16 HVR_VARIANTS=1 timeout 600 python tools/hip_vs_refemul.py > gpurun_out/r04_hip_vs_refemul_variants.log 2>&1; echo "variants rc=$?"
bash tools/single_call.sh r04_final 200 > /dev/null 2>&1; cat gpurun_out/r04_final_single_call_wall.jsonl
python - <<'PY'
import json
for f in ("r04_bench_final", "r04_bench_driverlike", "r04_bench_inorder"):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
        print(f, "fps",d["value"],"ms/step",d["ms_per_step"],"frac",d["roofline"]["frac"],"single",(d["roofline"].get("single_launch") or {}).get("frac"),"validated",d["validated_frames"])
        print("  ", {k:v["ms_per_step"] for k,v in d["kernels"].items()})
        if d.get("match"): print("  match",d["match"]["value"],d["match"]["roofline"]["frac"])
        print("  cpu", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
    except Exception as e: print(f, "ERR",e)
PY

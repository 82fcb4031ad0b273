#!/bin/bash
# The verification this round's numbers come from (run on the GPU box: gpurun -- 'bash tools/gpu_run.sh'):
# GPU test-suite, smoke, the default bench line, the same command under rocprofv3 --stats, two / four batches in flight.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke OK"
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-match --no-pcie --no-latency > /tmp/rp.json 2>/tmp/rp.err); echo "rocprof rc=$?"
cp /tmp/rp/*kernel_stats.csv gpurun_out/r02_kernel_stats_final.csv
tail -1 /tmp/rp.json > gpurun_out/r02_bench_under_rocprof.json
for k in 2 4; do
  timeout 600 python bench.py --contexts $k --no-pmc --no-match --cpu-frames 64 > gpurun_out/bench_ctx$k.json 2>/dev/null; echo "ctx$k rc=$?"
done

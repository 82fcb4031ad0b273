#!/bin/bash
# usage (GPU box): tools/pmc_pass.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
# Runs `python bench.py --pmc-child` (1 warm-up + 3 steps of the timed entry point, 64 x 1080p) under
# rocprofv3 --pmc once per counter list and prints the per-kernel sums (tools/pmc_sq.py) -> gpurun_out/<tag>.csv
set -e
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
dirs=""
i=0
for c in "$@"; do
  d=/tmp/pmc_${tag}_$i; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- python $root/bench.py --pmc-child > $out/${tag}_$i.log 2>&1) || { tail -5 $out/${tag}_$i.log; exit 1; }
  dirs="$dirs $d"; i=$((i+1))
done
python $root/tools/pmc_sq.py $dirs > $out/$tag.csv
cat $out/$tag.csv

export TMPDIR=/tmp; mkdir -p gpurun_out
Q="--no-pmc --no-match --no-cpu --no-latency --steps 20 --warmup 3"
for v in "X=1" "MISIFT_TILE_DESCR=0" "MISIFT_BIN=0" "MISIFT_TILE_DESCR=0 MISIFT_BIN=0 MISIFT_ORIENT_BLOCKS=4"; do
  echo "== $v"; env $v timeout 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['pcie_inclusive'])"
done

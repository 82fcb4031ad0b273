#!/bin/bash
# Developer aid: build A/B variants of libmisift.so that differ in -D macros of ONE source file.
#   tools/variants.sh kernels_points.hip name1 "-DDESCR_OCC=3" name2 "-DDESCR_OCC=2" ...
# -> build/variants/libmisift_<name>.so ; select at run time with MISIFT_LIB=<path>.
export MISIFT_TUNABLES=1      # the library reads its launch-shape / path variables only under this switch
set -e
cd "$(dirname "$0")/.."
make -s -j cudasift_amd/libmisift.so >/dev/null
src=$1; shift
base=${src%.hip}
mkdir -p build/variants build/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Icudasift_amd/csrc -Wno-unused-result -Wno-unused-value"
while [ $# -gt 0 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -c cudasift_amd/csrc/$src -o build/var/${base}_$name.o
  objs=""
  for o in build/*.o; do
    if [ "$(basename $o)" = "$base.o" ]; then objs="$objs build/var/${base}_$name.o"; else objs="$objs $o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libmisift_$name.so $objs
  echo "built build/variants/libmisift_$name.so ($defs)"
done

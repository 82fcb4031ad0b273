#!/bin/bash
# Developer aid: tools/single_call_sweep.sh over alternative libmisift.so builds (MISIFT_LIB), via libcudasift's loader path
tag=$1; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
for lib in "$@"; do
  if [ -n "$lib" ]; then cp cudasift_amd/libmisift.so /tmp/libmisift_saved.so; cp $lib cudasift_amd/libmisift.so; fi
  bash tools/single_call_sweep.sh ${tag}_$(basename "${lib:-intree}" .so) "" > /dev/null 2>&1
  if [ -n "$lib" ]; then cp /tmp/libmisift_saved.so cudasift_amd/libmisift.so; fi
done

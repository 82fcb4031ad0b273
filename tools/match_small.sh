#!/bin/bash
# Developer aid: the single-call MatchSiftData (2 x ~2000 and 2 x ~1100 records) under forced chunk counts, variant
# builds of kernels_match.hip (build/variants/libmisift_<name>.so) and the time-stamp build.
#   gpurun -- 'bash tools/match_small.sh tag'  -> gpurun_out/<tag>_match_small.txt
tag=${1:-r04}
export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/${tag}_match_small.txt; : > $out
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from synth import synth_frame
for (w, h) in ((1920, 1080), (1280, 960)):
    for f in (0, 1):
        synth_frame(f, w, h).tofile("/tmp/frame%d_%dx%d.f32" % (f, w, h))
PY
run() {  # $1 = label, rest = env
  label=$1; shift
  for wh in "1920 1080" "1280 960"; do
    set -- $wh "$@"; w=$1; h=$2; shift 2
    echo -n "[$label] " >> $out
    env GPU_MAX_HW_QUEUES=8 "$@" build/single_call /tmp/frame0_${w}x${h}.f32 /tmp/frame1_${w}x${h}.f32 $w $h 300 5 3.0 0 | grep '^{"what": "Match' >> $out
  done
}
run default A=1
for c in ${MATCH_SMALL_CHUNKS:-4 6 9 12 17 19 34}; do run "chunks=$c" MISIFT_MATCH_CHUNKS=$c; done
cp cudasift_amd/libmisift.so /tmp/libmisift_saved.so
for v in ${MATCH_SMALL_VARIANTS:-pipe0 wg2 noswap}; do
  if [ -f build/variants/libmisift_$v.so ]; then
    cp build/variants/libmisift_$v.so cudasift_amd/libmisift.so
    run "variant $v" A=1
    if [ $v = wg2 ]; then for c in 9 17 34; do run "variant $v chunks=$c" MISIFT_MATCH_CHUNKS=$c; done; fi
    cp /tmp/libmisift_saved.so cudasift_amd/libmisift.so
  fi
done
run "default again" A=1
if [ -f build/variants/libmisift_stamps.so ]; then
  for c in "" ${MATCH_SMALL_STAMP_CHUNKS:-6 17 34}; do
    MISIFT_LIB=build/variants/libmisift_stamps.so MISIFT_MATCH_CHUNKS=$c python tools/match_stamps.py 2052 2163 >> $out 2>&1
  done
  MISIFT_LIB=build/variants/libmisift_stamps.so python tools/match_stamps.py 1106 1205 >> $out 2>&1
  MISIFT_LIB=build/variants/libmisift_stamps.so MISIFT_MATCH_CHUNKS=1 python tools/match_stamps.py 128 64 >> $out 2>&1
fi
cat $out

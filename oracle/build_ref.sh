#!/usr/bin/env bash
# build_ref.sh — compile the only CPU-runnable piece of the reference's hot path
# (the matcher study's host routines in match.cu) straight from the reference tree.
#
# TEST INFRASTRUCTURE ONLY.  Output goes to oracle/_ref/ (git-ignored, travels to
# the GPU box with gpurun).  No reference source is copied into the repository:
# the needed line ranges are streamed from $REF/match.cu through a pipe into g++.
#
#   match.cu:57-141    MatchC1 (scalar), MatchC2/MatchC3 (AVX2, OpenMP), CheckMatches
#   match.cu:945-957   the data generator of main() (uniform rand()/RAND_MAX,
#                      every vector scaled by sqrt(128)/sum)
#
# NPTS is a compile-time constant in that code, so one library is built per size.
# The rest of the reference (extraction, MatchSiftData) is CUDA and cannot be
# built here — see DESIGN.md "Oracle".
set -euo pipefail
REF="${REF:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF/match.cu" ]; then
  echo "build_ref: $REF/match.cu not present (GPU box?) — keeping prebuilt oracle/_ref" >&2
  exit 0
fi
mkdir -p "$OUT"
for N in 1024 16384; do
  {
    cat <<EOF
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <iostream>
#include <immintrin.h>
#define NPTS $N
#define NDIM 128
EOF
    sed -n '57,141p' "$REF/match.cu"
    echo 'extern "C" void ref_generate(float *h_pts1, float *h_pts2, unsigned seed) { srand(seed);'
    sed -n '945,957p' "$REF/match.cu"
    echo '}'
    cat <<'EOF'
extern "C" int  ref_npts(void) { return NPTS; }
extern "C" void ref_match_c1(float *a, float *b, float *score, int *index) { MatchC1(a, b, score, index); }
extern "C" void ref_match_c3(float *a, float *b, float *score, int *index) { MatchC3(a, b, score, index); }
EOF
  } | g++ -x c++ -O2 -mavx2 -mfma -fopenmp -shared -fPIC -w -o "$OUT/libmatchref_$N.so" -
  echo "build_ref: built $OUT/libmatchref_$N.so"
done

# ImproveHomography: the reference's OWN geomFuncs.cpp (host-only C++), compiled where it lies against include/cudaSift.h
# (same SiftData / SiftPoint layout) and the mini-OpenCV stand-in (OpenCV itself is absent from this image).  Pins
# orc_improve_homography (tests/test_oracle_cpu.py).  Exported through a C wrapper appended on the fly.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
{
  cat "$REF/geomFuncs.cpp"
  cat <<'EOF2'
extern "C" int ref_improve_homography(void *pts, int numPts, float *homography, int numLoops, float minScore,
                                      float maxAmbiguity, float thresh)
{
  SiftData d;
  d.numPts = numPts; d.maxPts = numPts; d.h_data = (SiftPoint *)pts; d.d_data = 0;
  return ImproveHomography(d, homography, numLoops, minScore, maxAmbiguity, thresh);
}
EOF2
} | g++ -x c++ -O2 -ffp-contract=off -shared -fPIC -w -I"$ROOT/include" -I"$ROOT/cudasift_amd/compat" -o "$OUT/libgeomref.so" -
echo "build_ref: built $OUT/libgeomref.so"

# ---------------------------------------------------------------------------------------------------------------
# The reference's OWN extraction + matching code on a CPU SIMT emulator (VERDICT r2 "pin the extraction oracle").
# cudaImage.cu, cudaSiftH.cu (which #includes cudaSiftD.cu: all 25 kernels) and matching.cu are compiled where
# they lie, with oracle/simt_emul.h force-included as the "CUDA toolkit".  The only textual changes, applied in the
# pipe, are the two rewrites of oracle/launch_rewrite.sed (the launch syntax g++ cannot parse; one float->int conversion).
# Two flavours: -ffp-contract=off (every product rounded) and -ffp-contract=fast (g++ fuses multiply-adds the way a
# -fmad=true compiler may) — nvcc's actual choice is not observable here, the pair brackets it.
REWRITE="$HERE/launch_rewrite.sed"     # the two rewrites (launch syntax; the GPU's NaN -> 0 float-to-int conversion)
# emit one reference .cu with the rewrites applied; cudaSiftH.cu #includes cudaSiftD.cu textually: inline it in the pipe
ref_tu() {
  sed -f "$REWRITE" "$REF/$1.cu" | while IFS= read -r LINE; do
    if [ "$LINE" = '#include "cudaSiftD.cu"' ]; then
      sed -f "$REWRITE" "$REF/cudaSiftD.cu"
      # Everything of cudaSiftH.cu below this point is HOST code (ExtractSift, PrepareLaplaceKernels, the tap generators
      # of LowPass / ScaleDown): nvcc hands it to the host compiler (g++ -O2 -msse2, CMakeLists.txt:29: no FMA
      # instructions at all), so it is never contracted, whatever -fmad does to the kernels.  r03: with the host code
      # contracted too, the Laplace taps of a 6th / 7th octave came out different in 5 / 10 table entries.
      printf '%s\n' '#pragma GCC optimize ("fp-contract=off")'
    else printf '%s\n' "$LINE"; fi
  done
}
# -fno-toplevel-reorder: __shared__ variables (function-local statics here) are laid out in DECLARATION order, as
# nvcc lays out static shared memory.  It matters once: the descriptor kernel's vote with angle bin 8 in the last
# cell writes buffer[128]; in declaration order that is sums[0], which is overwritten before
# it is read (harmless, as on the GPU) — g++'s default reverse order would make it gauss[0] and poison every later
# descriptor of the block.
EMUFLAGS="-x c++ -std=c++17 -O2 -fPIC -fopenmp -mavx2 -mfma -fno-math-errno -fno-toplevel-reorder -w -include $HERE/simt_emul.h -I$REF"
for FLAVOUR in off fast; do
  OBJ="$OUT/refemul_$FLAVOUR"
  mkdir -p "$OBJ"
  PIDS=""
  for SRC in cudaImage cudaSiftH matching; do
    ref_tu $SRC | g++ $EMUFLAGS -ffp-contract=$FLAVOUR -c -o "$OBJ/$SRC.o" - & PIDS="$PIDS $!"
  done
  g++ $EMUFLAGS -ffp-contract=$FLAVOUR -c -o "$OBJ/wrap.o" "$HERE/refemul_wrap.cpp" & PIDS="$PIDS $!"
  g++ -std=c++17 -O2 -fPIC -fopenmp -c -o "$OBJ/engine.o" "$HERE/simt_emul.cpp" & PIDS="$PIDS $!"
  for P in $PIDS; do wait $P; done
  g++ -shared -fopenmp -o "$OUT/libcudasift_refemul_$FLAVOUR.so" "$OBJ"/cudaImage.o "$OBJ"/cudaSiftH.o "$OBJ"/matching.o \
      "$OBJ"/wrap.o "$OBJ"/engine.o
  echo "build_ref: built $OUT/libcudasift_refemul_$FLAVOUR.so (reference kernels on the CPU SIMT emulator)"
done
# ... and the reference's unmodified demo on top of it: mainSift.cpp + geomFuncs.cpp + the emulated library = the
# whole reference program running here without a GPU (prints the feature / match counts of mainSift.cpp:80-81).
# (its 1000 timing repetitions of ExtractSift, mainSift.cpp:66, are cut to 1 in the pipe: seconds instead of an hour)
sed 's/i<1000;i++/i<1;i++/' "$REF/mainSift.cpp" | g++ -x c++ -std=c++17 -O2 -w -fopenmp -I"$REF" -I"$ROOT/cudasift_amd/compat" \
    -c -o "$OUT/refemul_fast/main.o" -
g++ -std=c++17 -O2 -w -fopenmp -I"$REF" -I"$ROOT/cudasift_amd/compat" -o "$OUT/cudasift_refemul_main" \
    "$OUT/refemul_fast/main.o" "$REF/geomFuncs.cpp" -L"$OUT" -lcudasift_refemul_fast -Wl,-rpath,'$ORIGIN'
echo "build_ref: built $OUT/cudasift_refemul_main"

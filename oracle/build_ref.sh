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

export TMPDIR=/tmp
for v in "" noearly early6 early12 nopipe "" noearly; do
  if [ -n "$v" ]; then export MISIFT_LIB=$PWD/build/variants/libmisift_$v.so; else unset MISIFT_LIB; fi
  echo -n "${v:-default(early10)}: "; MATCH_REPS=8 python tools/match_prof.py | tail -1
done

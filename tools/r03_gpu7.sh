export TMPDIR=/tmp
for v in "" nobar nobar_nostore nobar_nostore_nogl ""; do
  if [ -n "$v" ]; then export MISIFT_LIB=$PWD/build/variants/libmisift_$v.so; else unset MISIFT_LIB; fi
  echo -n "${v:-default}: "; MATCH_REPS=8 python tools/match_prof.py | tail -1
done

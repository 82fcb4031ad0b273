#!/bin/bash
# Developer aid (GPU box): the SQ counters that say what bounds the kernels, for several libmisift.so builds.
#   tools/pmc_variants.sh <tag> "" build/variants/libmisift_x.so ...   -> gpurun_out/<tag>_<name>.csv (+ kernel durations)
export MISIFT_TUNABLES=1      # the library reads its launch-shape / path variables only under this switch
tag=$1; shift
for lib in "$@"; do
  name=$(basename "${lib:-intree}" .so); name=${name#libmisift_}
  MISIFT_LIB=${lib:+$(pwd)/$lib} bash tools/pmc_pass.sh ${tag}_$name \
    "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
    "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
    "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" > /dev/null 2>&1
  # dispatch durations of the first pass (kernels run one at a time under --pmc)
  python - /tmp/pmc_${tag}_${name}_0 >> gpurun_out/${tag}_$name.csv <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k in sorted(d):
    if "at::" in k or "elementwise" in k: continue
    print("duration_us,%s,launches=%d,sum=%.1f" % (k, len(d[k]), sum(d[k])))
PY
  echo "== $name"; cat gpurun_out/${tag}_$name.csv
done

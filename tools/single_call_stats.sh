export TMPDIR=/tmp; mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from synth import synth_frame
for f in (0, 1):
    synth_frame(f, 1920, 1080).tofile("/tmp/frame%d_1920x1080.f32" % f)
PY
(cd /tmp && rm -rf /tmp/scs && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/scs -o scs --output-format csv -- $R/build/single_call /tmp/frame0_1920x1080.f32 /tmp/frame1_1920x1080.f32 1920 1080 1000 5 3.0 0 > /tmp/scs.out 2>/tmp/scs.err)
find /tmp/scs -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_single_call_kernel_stats.csv \;
cat /tmp/scs.out | grep '^{'
cat gpurun_out/r04_single_call_kernel_stats.csv | cut -c1-150

export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/cliff.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_cliff_before.txt

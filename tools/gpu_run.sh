export TMPDIR=/tmp; mkdir -p gpurun_out
for q in 8 16; do
echo "GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q python tools/two_ctx.py 1 2 3 4 6 2>&1 | grep -v amdgpu.ids | cut -c1-60
done

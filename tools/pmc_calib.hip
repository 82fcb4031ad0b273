// pmc_calib.hip — known-byte kernels that calibrate rocprofv3's FETCH_SIZE on gfx950 for the two access patterns of
// libmisift.so: wide coalesced streaming reads (16 B per lane: the pyramid / scan kernels) and
// scattered 8-byte gathers (the bilinear texel pairs of the per-keypoint kernels).  Every kernel touches each part
// of a 1 GiB buffer (4x the 256 MiB Infinity Cache) exactly once, so the bytes HBM must deliver are known:
//   calib_stream     : every byte, 16 B per lane, consecutive lanes consecutive          -> M bytes
//   calib_gather64   : ONE 8-byte word of every 64-byte sector, sectors in a scrambled order (each wavefront's 64
//                      lanes hit 64 unrelated sectors)                                   -> M bytes if HBM is fetched
//                      in 64-B sectors or 128-B lines alike (every sector is needed)
//   calib_gather128  : one 8-byte word of every 128-byte line, scrambled                 -> M/2 (64-B sector fetch) or
//                      M (128-B line fetch): tells the fetch granularity of a gather miss
// tools/pmc_calib.py runs this under `rocprofv3 --pmc FETCH_SIZE` and writes profiles/r02_pmc_calibration.json.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void calib_fill(float4 *buf, size_t n16)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    buf[i] = make_float4((float)(i & 255), 1.0f, 2.0f, 3.0f);
}

__global__ void calib_stream(const float4 *__restrict__ buf, size_t n16, float *out)
{
  float acc = 0.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = buf[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == -1.0f) out[0] = acc;            // never true: keeps the loads alive
}

// unit = bytes between the words that are read (64 or 128); n units in the buffer (a power of two)
__global__ void calib_gather(const char *__restrict__ buf, size_t nunits, int unit, float *out)
{
  float acc = 0.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nunits; i += (size_t)gridDim.x * blockDim.x) {
    // odd multiplier: a bijection mod 2^k, and neighbouring lanes land 2.5 MB apart (40503 units)
    const size_t p = (i * 40503ull + 12345ull) & (nunits - 1);
    const float2 v = *reinterpret_cast<const float2 *>(buf + p * (size_t)unit);
    acc += v.x + v.y;
  }
  if (acc == -1.0f) out[0] = acc;
}

int main()
{
  const size_t M = 1ull << 30;
  char *buf = nullptr;
  float *out = nullptr;
  CHECK(hipMalloc((void **)&buf, M));
  CHECK(hipMalloc((void **)&out, 64));
  hipLaunchKernelGGL(calib_fill, dim3(4096), dim3(256), 0, 0, (float4 *)buf, M / 16);
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(calib_stream, dim3(8192), dim3(256), 0, 0, (const float4 *)buf, M / 16, out);
    hipLaunchKernelGGL(calib_gather, dim3(8192), dim3(256), 0, 0, (const char *)buf, M / 64, 64, out);
    hipLaunchKernelGGL(calib_gather, dim3(8192), dim3(256), 0, 0, (const char *)buf, M / 128, 128, out);
    CHECK(hipDeviceSynchronize());
  }
  printf("pmc_calib: buffer %zu bytes, 3 repetitions of stream / gather64 / gather128\n", M);
  CHECK(hipFree(buf));
  CHECK(hipFree(out));
  return 0;
}

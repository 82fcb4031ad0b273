// valu_rates.hip — issue cost of the instruction kinds the kernels of libmisift.so are made of, measured on the GPU
// (cycles per wave64 instruction per SIMD at 1, 2 and 4 resident waves per SIMD).  One workgroup of 64*W*4 threads per
// CU (W waves per SIMD), every wave runs UNROLL independent copies of one instruction ITER times.
//   build/valu_rates            -> table on stdout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

#define ITER 16384
#define REP8(x) x x x x x x x x

typedef float v2f __attribute__((ext_vector_type(2)));

// each body: 8 independent instructions on registers a0..a7 (chains of dependent ops per register, independent
// across registers: ILP 8)
// every kernel brackets its loop with s_memtime (shader clock) and s_memrealtime (100 MHz): block 0, thread 0 reports
#define TIC const unsigned long long c0_ = __builtin_readcyclecounter(), r0_ = __builtin_amdgcn_s_memrealtime()
#define TOC if (blockIdx.x == 0 && threadIdx.x == 0) { ((unsigned long long *)out)[4] = __builtin_readcyclecounter() - c0_; \
                                                      ((unsigned long long *)out)[5] = __builtin_amdgcn_s_memrealtime() - r0_; }
#define KERNEL(name, decl, body, fin)                                                     \
  __global__ void name(float *out, int iters)                                             \
  {                                                                                       \
    decl;                                                                                 \
    TIC;                                                                                  \
    for (int it = 0; it < iters; it++) { body }                                           \
    fin;                                                                                  \
    TOC;                                                                                  \
  }

#define F8 float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; float k = out[0], c = out[1]
#define OUT8 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[2] = a0
#define ASM8(ins) \
  asm volatile(ins : "+v"(a0) : "v"(k), "v"(c)); asm volatile(ins : "+v"(a1) : "v"(k), "v"(c)); \
  asm volatile(ins : "+v"(a2) : "v"(k), "v"(c)); asm volatile(ins : "+v"(a3) : "v"(k), "v"(c)); \
  asm volatile(ins : "+v"(a4) : "v"(k), "v"(c)); asm volatile(ins : "+v"(a5) : "v"(k), "v"(c)); \
  asm volatile(ins : "+v"(a6) : "v"(k), "v"(c)); asm volatile(ins : "+v"(a7) : "v"(k), "v"(c));

KERNEL(k_fma, F8, ASM8("v_fma_f32 %0, %0, %1, %2"), OUT8)
KERNEL(k_fmac, F8, ASM8("v_fmac_f32 %0, %1, %2"), OUT8)
KERNEL(k_mul, F8, ASM8("v_mul_f32 %0, %0, %1"), OUT8)
KERNEL(k_add, F8, ASM8("v_add_f32 %0, %0, %1"), OUT8)
KERNEL(k_floor, F8, ASM8("v_floor_f32 %0, %0"), OUT8)
KERNEL(k_cvt_i32, F8, ASM8("v_cvt_i32_f32 %0, %0"), OUT8)
KERNEL(k_rcp, F8, ASM8("v_rcp_f32 %0, %0"), OUT8)
KERNEL(k_sqrt, F8, ASM8("v_sqrt_f32 %0, %0"), OUT8)
KERNEL(k_sin, F8, ASM8("v_sin_f32 %0, %0"), OUT8)
KERNEL(k_exp, F8, ASM8("v_exp_f32 %0, %0"), OUT8)
KERNEL(k_mul_lo_u32, F8, ASM8("v_mul_lo_u32 %0, %0, %1"), OUT8)
KERNEL(k_mad_u32_u24, F8, ASM8("v_mad_u32_u24 %0, %0, %1, %2"), OUT8)
KERNEL(k_add_u32, F8, ASM8("v_add_u32 %0, %0, %1"), OUT8)
KERNEL(k_lshl_add_u32, F8, ASM8("v_lshl_add_u32 %0, %0, 2, %1"), OUT8)
// 64-bit integer multiply-add (what hipcc emits for an int index into a generic pointer: r03 found 38 in descr_all)
#define U8 unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; unsigned k = (unsigned)out[0] + 3u, c = (unsigned)out[1] + 5u
#define OUTU8 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345678ull) out[2] = (float)a0
#define ASMU8(ins) \
  asm volatile(ins : "+v"(a0) : "v"(k), "v"(c) : "vcc"); asm volatile(ins : "+v"(a1) : "v"(k), "v"(c) : "vcc"); \
  asm volatile(ins : "+v"(a2) : "v"(k), "v"(c) : "vcc"); asm volatile(ins : "+v"(a3) : "v"(k), "v"(c) : "vcc"); \
  asm volatile(ins : "+v"(a4) : "v"(k), "v"(c) : "vcc"); asm volatile(ins : "+v"(a5) : "v"(k), "v"(c) : "vcc"); \
  asm volatile(ins : "+v"(a6) : "v"(k), "v"(c) : "vcc"); asm volatile(ins : "+v"(a7) : "v"(k), "v"(c) : "vcc");
KERNEL(k_mad_u64_u32, U8, ASMU8("v_mad_u64_u32 %0, vcc, %1, %2, %0"), OUTU8)
KERNEL(k_lshl_add_u64, U8, ASMU8("v_lshl_add_u64 %0, %0, 2, %0"), OUTU8)
KERNEL(k_max3, F8, ASM8("v_max3_f32 %0, %0, %1, %2"), OUT8)
KERNEL(k_med3_i32, F8, ASM8("v_med3_i32 %0, %0, %1, %2"), OUT8)
KERNEL(k_cndmask, F8, ASM8("v_cndmask_b32 %0, %0, %1, vcc"), OUT8)
KERNEL(k_cmp_cnd, F8, ASM8("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc"), OUT8)
KERNEL(k_mov_dpp_shr, F8, ASM8("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), OUT8)
KERNEL(k_mov_dpp_rowshr, F8, ASM8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), OUT8)
KERNEL(k_add_dpp_shr, F8, ASM8("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), OUT8)
KERNEL(k_readlane, F8, ASM8("v_readlane_b32 s20, %0, 3\n v_mov_b32 %0, s20"), OUT8)

#define P8 v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; v2f k = {out[0], out[1]}, c = {out[1], out[0]}
#define OUTP8 if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y == 12345.678f) out[2] = a0.x
KERNEL(k_pk_fma, P8, ASM8("v_pk_fma_f32 %0, %0, %1, %2"), OUTP8)
KERNEL(k_pk_add, P8, ASM8("v_pk_add_f32 %0, %0, %1"), OUTP8)
KERNEL(k_pk_mul, P8, ASM8("v_pk_mul_f32 %0, %0, %1"), OUTP8)

// LDS reads: 8 independent loads per iteration from lane-dependent addresses, consumed by one add each
__global__ void k_ds_read_b32(float *out, int iters)
{
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = i;
  __syncthreads();
  float acc = 0;
  unsigned a = (threadIdx.x * 4u) & 16383u;                 // conflict-free: consecutive dwords
  TIC;
  for (int it = 0; it < iters; it++) {
    float v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                 "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a));
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  if (acc == 12345.678f) out[2] = acc;
  TOC;
}
__global__ void k_ds_read2_b32_rand(float *out, int iters)
{
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = i;
  __syncthreads();
  float acc = 0;
  unsigned a = ((threadIdx.x * 2654435761u) >> 20) & 0x3ffcu;   // scattered dword addresses (bank conflicts like a bilinear fetch)
  if (a > 16000u) a -= 2048u;
  TIC;
  for (int it = 0; it < iters; it++) {
    float v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n ds_read2_b32 %1, %4 offset0:40 offset1:41\n"
                 "ds_read2_b32 %2, %4 offset0:80 offset1:81\n ds_read2_b32 %3, %4 offset0:120 offset1:121\n s_waitcnt lgkmcnt(0)"
                 : "=v"(*(v2f *)&v0), "=v"(*(v2f *)&v2), "=v"(*(v2f *)&v4), "=v"(*(v2f *)&v6) : "v"(a));
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  if (acc == 12345.678f) out[2] = acc;
  TOC;
}
// a dependent chain: latency of one instruction (ILP 1)
KERNEL(k_fma_dep, float a0 = threadIdx.x; float k = out[0]; float c = out[1], REP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(k), "v"(c));), if (a0 == 12345.678f) out[2] = a0)

struct Entry { const char *name; void (*fn)(float *, int); int per_iter; };

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int clk_khz = 0;
  CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  float *out;
  CHECK(hipMalloc((void **)&out, 256));
  float h[4] = {1.0000001f, 0.5f, 0, 0};
  CHECK(hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice));
  Entry e[] = {
#define E(n, c) {#n, n, c}
    E(k_fma, 8), E(k_fmac, 8), E(k_mul, 8), E(k_add, 8), E(k_pk_fma, 8), E(k_pk_add, 8), E(k_pk_mul, 8), E(k_floor, 8), E(k_cvt_i32, 8),
    E(k_rcp, 8), E(k_sqrt, 8), E(k_sin, 8), E(k_exp, 8), E(k_mul_lo_u32, 8), E(k_mad_u32_u24, 8), E(k_add_u32, 8),
    E(k_lshl_add_u32, 8), E(k_mad_u64_u32, 8), E(k_lshl_add_u64, 8), E(k_max3, 8), E(k_med3_i32, 8), E(k_cndmask, 8), E(k_cmp_cnd, 16), E(k_mov_dpp_shr, 8),
    E(k_mov_dpp_rowshr, 8), E(k_add_dpp_shr, 8), E(k_readlane, 16), E(k_ds_read_b32, 8), E(k_ds_read2_b32_rand, 4), E(k_fma_dep, 8),
  };
  hipEvent_t t0, t1;
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  printf("device: %s, %d CUs, clock %d MHz (attribute)\n", prop.gcnArchName, cus, clk_khz / 1000);
  printf("shader cycles per wave64 instruction per SIMD with W wavefronts per SIMD on every SIMD of the chip (one workgroup of 256*W\n"
         "threads per CU).  Cycles = whole-kernel time (HIP events) x shader clock / (instructions per wavefront x W); the shader clock\n"
         "is s_memtime / s_memrealtime (100 MHz) of the same launch.  [wave0] = the same quantity from the s_memtime interval of\n"
         "wavefront 0 alone / W: the issue arbiter favours the oldest wavefront, so wavefront 0 runs at its single-wavefront speed\n"
         "whatever W is and finishes early -- that column is NOT a throughput (it is what r02's first table mistook for one).\n");
  printf("%-22s %8s %8s %8s %8s %8s   %8s %8s\n", "instruction", "W=1", "W=2", "W=3", "W=4", "MHz(4)", "[wave0]1", "[wave0]4");
  for (auto &k : e) {
    printf("%-22s", k.name);
    double w0[5] = {0, 0, 0, 0, 0}, mhz4 = 0;
    for (int w = 1; w <= 4; w++) {
      const int threads = 64 * 4 * w;            // w waves per SIMD, one workgroup per CU
      hipLaunchKernelGGL(k.fn, dim3(cus), dim3(threads), 0, 0, out, 64);
      CHECK(hipDeviceSynchronize());
      const int iters = ITER * 4;
      CHECK(hipEventRecord(t0));
      hipLaunchKernelGGL(k.fn, dim3(cus), dim3(threads), 0, 0, out, iters);
      CHECK(hipEventRecord(t1));
      CHECK(hipEventSynchronize(t1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, t0, t1));
      unsigned long long tc[2];
      CHECK(hipMemcpy(tc, (char *)out + 32, sizeof(tc), hipMemcpyDeviceToHost));
      const double mhz = 100.0 * (double)tc[0] / (double)tc[1];
      const double instr_per_simd = (double)iters * k.per_iter * w;
      printf(" %8.2f", ms * 1e-3 * mhz * 1e6 / instr_per_simd);
      w0[w] = (double)tc[0] / instr_per_simd;
      if (w == 4) mhz4 = mhz;
    }
    printf(" %8.0f   %8.2f %8.2f\n", mhz4, w0[1], w0[4]);
  }
  return 0;
}

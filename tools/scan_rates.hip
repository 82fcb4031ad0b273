// scan_rates.hip — what limits the blur part of dog_scan: the instruction stream of one scan row (three scale pairs,
// vertical + horizontal 9-tap passes, DoG, |.| maximum) run from registers in isolation, at 1..8 wavefronts per SIMD,
// in the order the compiler emits for kernels_dog.hip ("chain": every 5-deep fmaf chain back to back) and with the
// four / eight chains of a pass interleaved by hand ("ilv"), plus dependent-chain latencies of the packed ops.
//   build/scan_rates            -> table on stdout (cycles per wave-row per SIMD, from whole-kernel time)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
template <int CTRL> __device__ __forceinline__ float dpp(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_left(float v) { return dpp<0x138>(v); }    // wave_shr:1
__device__ __forceinline__ float from_right(float v) { return dpp<0x130>(v); }   // wave_shl:1
__device__ __forceinline__ v2f from_left2(v2f v) { return mk2(from_left(v.x), from_left(v.y)); }
__device__ __forceinline__ v2f from_right2(v2f v) { return mk2(from_right(v.x), from_right(v.y)); }
struct Taps2 { v2f k0, k1, k2, k3, k4; };
struct Pair4 { v2f x, y, z, w; };
__device__ __forceinline__ Taps2 load_taps2(const v2f *tk) { Taps2 t; t.k0 = tk[0]; t.k1 = tk[1]; t.k2 = tk[2]; t.k3 = tk[3]; t.k4 = tk[4]; return t; }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// ---- order A: as in kernels_dog.hip (one chain after the other; the compiler keeps that order)
__device__ __forceinline__ v2f conv9p(const Taps2 &t, v2f c, v2f p1, v2f p2, v2f p3, v2f p4)
{
  v2f s = t.k0 * c;
  s = pk_fma(t.k1, p1, s); s = pk_fma(t.k2, p2, s); s = pk_fma(t.k3, p3, s); s = pk_fma(t.k4, p4, s);
  return s;
}
__device__ __forceinline__ Pair4 blur_pair_chain(const Taps2 &t, float4 c, float4 p1, float4 p2, float4 p3, float4 p4)
{
  Pair4 v;
  v.x = conv9p(t, mk2(c.x, c.x), mk2(p1.x, p1.x), mk2(p2.x, p2.x), mk2(p3.x, p3.x), mk2(p4.x, p4.x));
  v.y = conv9p(t, mk2(c.y, c.y), mk2(p1.y, p1.y), mk2(p2.y, p2.y), mk2(p3.y, p3.y), mk2(p4.y, p4.y));
  v.z = conv9p(t, mk2(c.z, c.z), mk2(p1.z, p1.z), mk2(p2.z, p2.z), mk2(p3.z, p3.z), mk2(p4.z, p4.z));
  v.w = conv9p(t, mk2(c.w, c.w), mk2(p1.w, p1.w), mk2(p2.w, p2.w), mk2(p3.w, p3.w), mk2(p4.w, p4.w));
  const v2f lx = from_left2(v.x), ly = from_left2(v.y), lz = from_left2(v.z), lw = from_left2(v.w);
  const v2f rx = from_right2(v.x), ry = from_right2(v.y), rz = from_right2(v.z), rw = from_right2(v.w);
  Pair4 h;
  h.x = conv9p(t, v.x, lw + v.y, lz + v.z, ly + v.w, lx + rx);
  h.y = conv9p(t, v.y, v.x + v.z, lw + v.w, lz + rx, ly + ry);
  h.z = conv9p(t, v.z, v.y + v.w, v.x + rx, lw + ry, lz + rz);
  h.w = conv9p(t, v.w, v.z + rx, v.y + ry, v.x + rz, lw + rw);
  return h;
}
// ---- order B: the four chains of a pass advance together (tap by tap); a scheduling barrier after every tap keeps it so
#define SB __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ Pair4 blur_pair_ilv(const Taps2 &t, float4 c, float4 p1, float4 p2, float4 p3, float4 p4)
{
  Pair4 v;
  v.x = t.k0 * mk2(c.x, c.x); v.y = t.k0 * mk2(c.y, c.y); v.z = t.k0 * mk2(c.z, c.z); v.w = t.k0 * mk2(c.w, c.w); SB;
#define VSTEP(K, P) v.x = pk_fma(K, mk2(P.x, P.x), v.x); v.y = pk_fma(K, mk2(P.y, P.y), v.y); v.z = pk_fma(K, mk2(P.z, P.z), v.z); v.w = pk_fma(K, mk2(P.w, P.w), v.w); SB
  VSTEP(t.k1, p1); VSTEP(t.k2, p2); VSTEP(t.k3, p3); VSTEP(t.k4, p4);
  const v2f lx = from_left2(v.x), ly = from_left2(v.y), lz = from_left2(v.z), lw = from_left2(v.w);
  const v2f rx = from_right2(v.x), ry = from_right2(v.y), rz = from_right2(v.z), rw = from_right2(v.w);
  SB;
  Pair4 h;
  h.x = t.k0 * v.x; h.y = t.k0 * v.y; h.z = t.k0 * v.z; h.w = t.k0 * v.w;
  const v2f a1 = lw + v.y, b1 = v.x + v.z, c1 = v.y + v.w, d1 = v.z + rx; SB;
  h.x = pk_fma(t.k1, a1, h.x); h.y = pk_fma(t.k1, b1, h.y); h.z = pk_fma(t.k1, c1, h.z); h.w = pk_fma(t.k1, d1, h.w);
  const v2f a2 = lz + v.z, b2 = lw + v.w, c2 = v.x + rx, d2 = v.y + ry; SB;
  h.x = pk_fma(t.k2, a2, h.x); h.y = pk_fma(t.k2, b2, h.y); h.z = pk_fma(t.k2, c2, h.z); h.w = pk_fma(t.k2, d2, h.w);
  const v2f a3 = ly + v.w, b3 = lz + rx, c3 = lw + ry, d3 = v.x + rz; SB;
  h.x = pk_fma(t.k3, a3, h.x); h.y = pk_fma(t.k3, b3, h.y); h.z = pk_fma(t.k3, c3, h.z); h.w = pk_fma(t.k3, d3, h.w);
  const v2f a4 = lx + rx, b4 = ly + ry, c4 = lz + rz, d4 = lw + rw; SB;
  h.x = pk_fma(t.k4, a4, h.x); h.y = pk_fma(t.k4, b4, h.y); h.z = pk_fma(t.k4, c4, h.z); h.w = pk_fma(t.k4, d4, h.w); SB;
  return h;
}

// ---- order C: the neighbour lanes' vertical sums come through LDS (2 ds_write_b128 + 4 ds_read_b128 per scale pair)
// instead of 16 DPP moves: LDS instructions do not take VALU issue slots.  xch = this wavefront's 2 x 66 float4 slots
// (slot 0 and 65 stay zero = what DPP bound_ctrl gives lanes 0 and 63), already offset by the lane.
__device__ __forceinline__ Pair4 blur_pair_lds(const Taps2 &t, float4 c, float4 p1, float4 p2, float4 p3, float4 p4, float4 *xch)
{
  Pair4 v;
  v.x = conv9p(t, mk2(c.x, c.x), mk2(p1.x, p1.x), mk2(p2.x, p2.x), mk2(p3.x, p3.x), mk2(p4.x, p4.x));
  v.y = conv9p(t, mk2(c.y, c.y), mk2(p1.y, p1.y), mk2(p2.y, p2.y), mk2(p3.y, p3.y), mk2(p4.y, p4.y));
  xch[1] = make_float4(v.x.x, v.x.y, v.y.x, v.y.y);
  v.z = conv9p(t, mk2(c.z, c.z), mk2(p1.z, p1.z), mk2(p2.z, p2.z), mk2(p3.z, p3.z), mk2(p4.z, p4.z));
  v.w = conv9p(t, mk2(c.w, c.w), mk2(p1.w, p1.w), mk2(p2.w, p2.w), mk2(p3.w, p3.w), mk2(p4.w, p4.w));
  xch[66 + 1] = make_float4(v.z.x, v.z.y, v.w.x, v.w.y);
  const float4 la = xch[0], lb = xch[66], ra = xch[2], rb = xch[66 + 2];
  const v2f lx = mk2(la.x, la.y), ly = mk2(la.z, la.w), lz = mk2(lb.x, lb.y), lw = mk2(lb.z, lb.w);
  const v2f rx = mk2(ra.x, ra.y), ry = mk2(ra.z, ra.w), rz = mk2(rb.x, rb.y), rw = mk2(rb.z, rb.w);
  Pair4 h;
  h.x = conv9p(t, v.x, lw + v.y, lz + v.z, ly + v.w, lx + rx);
  h.y = conv9p(t, v.y, v.x + v.z, lw + v.w, lz + rx, ly + ry);
  h.z = conv9p(t, v.z, v.y + v.w, v.x + rx, lw + ry, lz + rz);
  h.w = conv9p(t, v.w, v.z + rx, v.y + ry, v.x + rz, lw + rw);
  return h;
}

// ---- order D (r04, VERDICT r03 "next" #5): the VERTICAL pass on the matrix pipe.  V_s[x] = sum_j k_s[j] p_j[x] is a
// shared-operand contraction: v_mfma_f32_4x4x1_16b_f32 with A = tap j of four scales (lane l holds the tap of scale
// l % 4: the same in all 16 blocks), B = this lane's pixel, chained over the five taps through C — output register r of
// every lane is scale r's sum for the lane's OWN pixel.  40 MFMAs per wave-row (4 pixel components x 2 scale groups x 5
// taps) replace 60 packed multiply-adds; whether the chain rounds like the fmaf chain is checked by k_mfma_check.
typedef float v4f __attribute__((ext_vector_type(4)));
struct TapsA { float g0[5], g1[5]; };          // per lane: tap j of scale (l % 4) + 1 (group 0) / (l % 4) + 5 (group 1)
__device__ __forceinline__ v4f vert_mfma(const float (&a)[5], float c, float p1, float p2, float p3, float p4)
{
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], c, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], p3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[4], p4, acc, 0, 0, 0);
  return acc;
}
// horizontal pass of one scale pair from its vertical sums (the second half of blur_pair_chain)
__device__ __forceinline__ Pair4 horiz_pair(const Taps2 &t, const Pair4 &v)
{
  const v2f lx = from_left2(v.x), ly = from_left2(v.y), lz = from_left2(v.z), lw = from_left2(v.w);
  const v2f rx = from_right2(v.x), ry = from_right2(v.y), rz = from_right2(v.z), rw = from_right2(v.w);
  Pair4 h;
  h.x = conv9p(t, v.x, lw + v.y, lz + v.z, ly + v.w, lx + rx);
  h.y = conv9p(t, v.y, v.x + v.z, lw + v.w, lz + rx, ly + ry);
  h.z = conv9p(t, v.z, v.y + v.w, v.x + rx, lw + ry, lz + rz);
  h.w = conv9p(t, v.w, v.z + rx, v.y + ry, v.x + rz, lw + rw);
  return h;
}

#define TIC const unsigned long long c0_ = __builtin_readcyclecounter(), r0_ = __builtin_amdgcn_s_memrealtime()
#define TOC if (blockIdx.x == 0 && threadIdx.x == 0) { ((unsigned long long *)out)[4] = __builtin_readcyclecounter() - c0_; \
                                                      ((unsigned long long *)out)[5] = __builtin_amdgcn_s_memrealtime() - r0_; }

template <int ORDER>
__global__ void k_row(float *out, int iters)
{
  __shared__ v2f s_taps[16][16];
  __shared__ float4 s_xch[16][2 * 66];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 2 * 66; i += 64) s_xch[wave][i] = make_float4(0, 0, 0, 0);
  float4 *xch = &s_xch[wave][lane];
  if (lane < 15) s_taps[wave][lane] = mk2(out[0] * (1.0f + lane), out[1] * (2.0f + lane));
  __syncthreads();
  const v2f *tk = s_taps[wave];
  float f = (float)threadIdx.x;
  float4 c = make_float4(f, f + 1, f + 2, f + 3), p1 = make_float4(f * 2, f * 3, f * 4, f * 5), p2 = p1, p3 = c, p4 = p1;
  p2.x += 1; p3.y += 2; p4.z += 3;
  float acc = 0;
  TIC;
  for (int it = 0; it < iters; it++) {
    asm volatile("" ::: "memory");
    float4 d[5];
    Taps2 t = load_taps2(tk);
    __builtin_amdgcn_sched_barrier(0);
    const Pair4 b0 = ORDER == 2 ? blur_pair_lds(t, c, p1, p2, p3, p4, xch) : ORDER ? blur_pair_ilv(t, c, p1, p2, p3, p4) : blur_pair_chain(t, c, p1, p2, p3, p4);
    asm volatile("" ::: "memory");
    t = load_taps2(tk + 5);
    __builtin_amdgcn_sched_barrier(0);
    d[0] = make_float4(b0.x.y - b0.x.x, b0.y.y - b0.y.x, b0.z.y - b0.z.x, b0.w.y - b0.w.x);
    const Pair4 b1 = ORDER == 2 ? blur_pair_lds(t, c, p1, p2, p3, p4, xch) : ORDER ? blur_pair_ilv(t, c, p1, p2, p3, p4) : blur_pair_chain(t, c, p1, p2, p3, p4);
    asm volatile("" ::: "memory");
    t = load_taps2(tk + 10);
    __builtin_amdgcn_sched_barrier(0);
    d[1] = make_float4(b1.x.x - b0.x.y, b1.y.x - b0.y.y, b1.z.x - b0.z.y, b1.w.x - b0.w.y);
    d[2] = make_float4(b1.x.y - b1.x.x, b1.y.y - b1.y.x, b1.z.y - b1.z.x, b1.w.y - b1.w.x);
    const Pair4 b2 = ORDER == 2 ? blur_pair_lds(t, c, p1, p2, p3, p4, xch) : ORDER ? blur_pair_ilv(t, c, p1, p2, p3, p4) : blur_pair_chain(t, c, p1, p2, p3, p4);
    d[3] = make_float4(b2.x.x - b1.x.y, b2.y.x - b1.y.y, b2.z.x - b1.z.y, b2.w.x - b1.w.y);
    d[4] = make_float4(b2.x.y - b2.x.x, b2.y.y - b2.y.x, b2.z.y - b2.z.x, b2.w.y - b2.w.x);
    float amax = 0.0f;
#pragma unroll
    for (int p = 0; p < 5; p++) amax = max3f(max3f(amax, fabsf(d[p].x), fabsf(d[p].y)), fabsf(d[p].z), fabsf(d[p].w));
    acc = fmaxf(acc, amax);
    // keep the next iteration dependent on this one's inputs only through the (never true) branch below
    if (__builtin_expect(amax == 12345.678f, 0)) { c.x += 1.0f; p1.y += 1.0f; }
  }
  if (acc == 12345.678f) out[2] = acc;
  TOC;
}

__global__ void k_row_mfma(float *out, int iters)
{
  __shared__ v2f s_taps[16][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 15) s_taps[wave][lane] = mk2(out[0] * (1.0f + lane), out[1] * (2.0f + lane));
  __syncthreads();
  const v2f *tk = s_taps[wave];
  TapsA A;
#pragma unroll
  for (int j = 0; j < 5; j++) {                 // (timing: any per-lane constants do)
    A.g0[j] = out[0] * (1.0f + (lane & 3) + j);
    A.g1[j] = out[1] * (5.0f + (lane & 3) + j);
  }
  float f = (float)threadIdx.x;
  float4 c = make_float4(f, f + 1, f + 2, f + 3), p1 = make_float4(f * 2, f * 3, f * 4, f * 5), p2 = p1, p3 = c, p4 = p1;
  p2.x += 1; p3.y += 2; p4.z += 3;
  float acc = 0;
  TIC;
  for (int it = 0; it < iters; it++) {
    asm volatile("" ::: "memory");
    float4 d[5];
    // vertical sums of all six scales: 8 independent 5-deep MFMA chains
    const v4f ax = vert_mfma(A.g0, c.x, p1.x, p2.x, p3.x, p4.x), ay = vert_mfma(A.g0, c.y, p1.y, p2.y, p3.y, p4.y);
    const v4f az = vert_mfma(A.g0, c.z, p1.z, p2.z, p3.z, p4.z), aw = vert_mfma(A.g0, c.w, p1.w, p2.w, p3.w, p4.w);
    const v4f bx = vert_mfma(A.g1, c.x, p1.x, p2.x, p3.x, p4.x), by = vert_mfma(A.g1, c.y, p1.y, p2.y, p3.y, p4.y);
    const v4f bz = vert_mfma(A.g1, c.z, p1.z, p2.z, p3.z, p4.z), bw = vert_mfma(A.g1, c.w, p1.w, p2.w, p3.w, p4.w);
    Taps2 t = load_taps2(tk);
    __builtin_amdgcn_sched_barrier(0);
    Pair4 v;
    v.x = mk2(ax[0], ax[1]); v.y = mk2(ay[0], ay[1]); v.z = mk2(az[0], az[1]); v.w = mk2(aw[0], aw[1]);
    const Pair4 b0 = horiz_pair(t, v);
    asm volatile("" ::: "memory");
    t = load_taps2(tk + 5);
    __builtin_amdgcn_sched_barrier(0);
    d[0] = make_float4(b0.x.y - b0.x.x, b0.y.y - b0.y.x, b0.z.y - b0.z.x, b0.w.y - b0.w.x);
    v.x = mk2(ax[2], ax[3]); v.y = mk2(ay[2], ay[3]); v.z = mk2(az[2], az[3]); v.w = mk2(aw[2], aw[3]);
    const Pair4 b1 = horiz_pair(t, v);
    asm volatile("" ::: "memory");
    t = load_taps2(tk + 10);
    __builtin_amdgcn_sched_barrier(0);
    d[1] = make_float4(b1.x.x - b0.x.y, b1.y.x - b0.y.y, b1.z.x - b0.z.y, b1.w.x - b0.w.y);
    d[2] = make_float4(b1.x.y - b1.x.x, b1.y.y - b1.y.x, b1.z.y - b1.z.x, b1.w.y - b1.w.x);
    v.x = mk2(bx[0], bx[1]); v.y = mk2(by[0], by[1]); v.z = mk2(bz[0], bz[1]); v.w = mk2(bw[0], bw[1]);
    const Pair4 b2 = horiz_pair(t, v);
    d[3] = make_float4(b2.x.x - b1.x.y, b2.y.x - b1.y.y, b2.z.x - b1.z.y, b2.w.x - b1.w.y);
    d[4] = make_float4(b2.x.y - b2.x.x, b2.y.y - b2.y.x, b2.z.y - b2.z.x, b2.w.y - b2.w.x);
    float amax = 0.0f;
#pragma unroll
    for (int p = 0; p < 5; p++) amax = max3f(max3f(amax, fabsf(d[p].x), fabsf(d[p].y)), fabsf(d[p].z), fabsf(d[p].w));
    acc = fmaxf(acc, amax);
    if (__builtin_expect(amax == 12345.678f, 0)) { c.x += 1.0f; p1.y += 1.0f; }
  }
  if (acc == 12345.678f) out[2] = acc;
  TOC;
}

// Does the MFMA chain round like the fmaf chain of conv9 (first product rounded, four fused multiply-adds)?  Every lane
// evaluates both on pseudo-random operands (taps in (0, 1), pixels in (0, 255), and a few denormal / huge ones);
// res[0] = evaluations, res[1] = bit mismatches.
__global__ void k_mfma_check(unsigned *res, int iters)
{
  const int lane = threadIdx.x & 63;
  unsigned seed = 12345u + 7919u * threadIdx.x + 104729u * blockIdx.x, bad = 0, n = 0;
  auto rnd = [&]() -> float { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) * (1.0f / 16777216.0f); };
  for (int it = 0; it < iters; it++) {
    float a[5], px[5];
    for (int j = 0; j < 5; j++) { a[j] = rnd() * (j ? 0.3f : 1.0f); px[j] = rnd() * (j ? 510.0f : 255.0f); }
    if ((it & 63) == 7) px[2] = 1e-41f;              // a denormal operand
    if ((it & 63) == 9) a[3] = 3e-39f;               // a denormal tap
    if ((it & 63) == 11) px[4] = 3e37f;
    const v4f m = vert_mfma(a, px[0], px[1], px[2], px[3], px[4]);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      // register r = the taps held by lane (lane & ~3) + r, this lane's pixels
      float s = __shfl(a[0], (lane & ~3) + r) * px[0];
      for (int j = 1; j < 5; j++) s = __builtin_fmaf(__shfl(a[j], (lane & ~3) + r), px[j], s);
      n++;
      if (__float_as_uint(s) != __float_as_uint(m[r])) bad++;
    }
  }
  atomicAdd(&res[0], n);
  atomicAdd(&res[1], bad);
}

// dependent chains of packed fmas: ILP 1, 2, 4, 8 (8 instructions per iteration each)
#define PKF(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(c));
#define PKF_OS(r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(r) : "v"(k), "v"(c));
#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(k1), "v"(c1));
#define DPPM(r, s) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(s));
#define DECL v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
  v2f k = {out[0], out[1]}, c = {out[1], out[0]}; float k1 = out[0], c1 = out[1]; float m0 = 0, m1 = 0, m2 = 0, m3 = 0
#define FIN if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y + m0 + m1 + m2 + m3 == 12345.678f) out[2] = a0.x
#define KERNEL(name, body) __global__ void name(float *out, int iters) { DECL; TIC; for (int it = 0; it < iters; it++) { body } FIN; TOC; }
KERNEL(k_pk_dep1, PKF(a0) PKF(a0) PKF(a0) PKF(a0) PKF(a0) PKF(a0) PKF(a0) PKF(a0))
KERNEL(k_pk_dep2, PKF(a0) PKF(a1) PKF(a0) PKF(a1) PKF(a0) PKF(a1) PKF(a0) PKF(a1))
KERNEL(k_pk_dep4, PKF(a0) PKF(a1) PKF(a2) PKF(a3) PKF(a0) PKF(a1) PKF(a2) PKF(a3))
KERNEL(k_pk_dep8, PKF(a0) PKF(a1) PKF(a2) PKF(a3) PKF(a4) PKF(a5) PKF(a6) PKF(a7))
KERNEL(k_pk_dep1_opsel, PKF_OS(a0) PKF_OS(a0) PKF_OS(a0) PKF_OS(a0) PKF_OS(a0) PKF_OS(a0) PKF_OS(a0) PKF_OS(a0))
KERNEL(k_pk_dep4_opsel, PKF_OS(a0) PKF_OS(a1) PKF_OS(a2) PKF_OS(a3) PKF_OS(a0) PKF_OS(a1) PKF_OS(a2) PKF_OS(a3))
// a packed chain with an (independent) DPP move after every link: what the vertical pass looks like in the kernel
KERNEL(k_pk_dep1_dpp, PKF(a0) DPPM(m0, a1.x) PKF(a0) DPPM(m1, a1.y) PKF(a0) DPPM(m2, a2.x) PKF(a0) DPPM(m3, a2.y))
// DPP move of a value the previous instruction produced (VALU write -> DPP read hazard)
KERNEL(k_pk_then_dpp, PKF(a0) DPPM(m0, a0.x) PKF(a1) DPPM(m1, a1.x) PKF(a2) DPPM(m2, a2.x) PKF(a3) DPPM(m3, a3.x))

struct Entry { const char *name; void (*fn)(float *, int); double per_iter; };

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  CHECK(hipMalloc((void **)&out, 256));
  float h[4] = {1.0000001f, 0.5f, 0, 0};
  CHECK(hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice));
  Entry e[] = {
    {"row_chain (per row)", k_row<0>, 1}, {"row_ilv (per row)", k_row<1>, 1}, {"row_lds (per row)", k_row<2>, 1},
    {"row_mfma_vert (per row)", k_row_mfma, 1},
    {"pk_fma dep ILP1", k_pk_dep1, 8}, {"pk_fma dep ILP2", k_pk_dep2, 8}, {"pk_fma dep ILP4", k_pk_dep4, 8}, {"pk_fma dep ILP8", k_pk_dep8, 8},
    {"pk_fma opsel ILP1", k_pk_dep1_opsel, 8}, {"pk_fma opsel ILP4", k_pk_dep4_opsel, 8},
    {"pk ILP1 + indep dpp", k_pk_dep1_dpp, 8}, {"pk -> dpp of result", k_pk_then_dpp, 8},
  };
  printf("device: %s, %d CUs; shader cycles per item per SIMD = whole-kernel time (HIP events) x shader clock / (items per wavefront x W),\n"
         "W wavefronts per SIMD on every SIMD (W <= 4: one workgroup per CU; 6, 8: two).  [wave0] = s_memtime interval of wavefront 0 / W:\n"
         "the oldest wavefront is favoured by the issue arbiter and runs at its solo speed -- not a throughput.\n", prop.gcnArchName, cus);
  printf("%-24s %9s %9s %9s %9s %9s %9s %8s %9s\n", "item", "W=1", "W=2", "W=3", "W=4", "W=6", "W=8", "MHz(4)", "[wave0]4");
  {
    unsigned *res, hres[2] = {0, 0};
    CHECK(hipMalloc((void **)&res, 8));
    CHECK(hipMemset(res, 0, 8));
    hipLaunchKernelGGL(k_mfma_check, dim3(256), dim3(256), 0, 0, res, 4096);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hres, res, 8, hipMemcpyDeviceToHost));
    printf("mfma_f32_4x4x1 chain vs the fmaf chain of conv9: %u evaluations, %u bit mismatches\n", hres[0], hres[1]);
    CHECK(hipFree(res));
  }
  hipEvent_t t0, t1;
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  const int ws[] = {1, 2, 3, 4, 6, 8};
  for (auto &k : e) {
    printf("%-24s", k.name);
    double mhz = 0, w0_4 = 0;
    for (int w : ws) {
      const int per_block = w <= 4 ? w : w / 2;
      const int threads = 64 * 4 * per_block, blocks = cus * (w <= 4 ? 1 : 2);
      const int iters = k.per_iter == 1 ? 2048 : 16384;
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, 64);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(t0));
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, iters);
      CHECK(hipEventRecord(t1));
      CHECK(hipDeviceSynchronize());
      unsigned long long tc[2];
      CHECK(hipMemcpy(tc, (char *)out + 32, sizeof(tc), hipMemcpyDeviceToHost));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, t0, t1));
      const double mhz_w = 100.0 * (double)tc[0] / (double)tc[1];
      printf(" %9.2f", ms * 1e-3 * mhz_w * 1e6 / ((double)iters * k.per_iter * w));
      if (w == 4) { mhz = mhz_w; w0_4 = (double)tc[0] / ((double)iters * k.per_iter * w); }
    }
    printf(" %8.0f %9.2f\n", mhz, w0_4);
  }
  return 0;
}

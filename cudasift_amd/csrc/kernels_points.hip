// kernels_points.hip — per-keypoint kernels for gfx950 (one wavefront per keypoint).
//
//   orient_kernel / orient_all_kernel  replace ComputeOrientationsCONST (reference cudaSiftD.cu:972-1057,
//                                      host cudaSiftH.cu:353-369)
//   descr_kernel / descr_all_kernel    replace ExtractSiftDescriptorsCONSTNew (reference cudaSiftD.cu:308-417 +
//                                      FastAtan2 :295-306, host cudaSiftH.cu:371-382)
//   rescale_kernel                     replaces RescalePositions (reference cudaSiftD.cu:753-761)
//   (the ..._all kernels take every octave of every frame in one launch from the Detection staging area and
//    lay the final SiftPoint array out in the reference's segment order)
//
// The reference uses 121- and 128-thread blocks (32-lane warps, texture unit fetches, shared-memory float
// atomics).  Here a 64-lane wavefront owns a keypoint: bilinear "texture" fetches are manual (tex2d in
// common.hpp: one 8-byte load per texel pair, optional 8-bit weight quantisation like the CUDA texture unit,
// clamp-free path for interior patches); orientation differences come out of ONE shared 13x13 grid of
// fetches; both histograms are accumulated WITHOUT atomics from wave-private LDS tables (descriptor: per-bin
// planes, 16 b128 loads + 64 literal-weight FMAs per lane); the norms are 64-lane butterfly reductions.
// Four wavefronts share a workgroup only to fill the CU — they never synchronise.
#include "common.hpp"

#define WAVES_PER_BLOCK 4

__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide reductions on the DPP path (quad_perm / row_half_mirror / row_mirror / row_bcast15 / row_bcast31): six
// full-rate VALU instructions with VALU latency.  The __shfl_xor butterfly they replace compiles to ds_bpermute_b32 —
// six dependent trips through the LDS crossbar, ~100 cycles each, on the critical path of every keypoint.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)          // lanes outside ROW_MASK (and invalid sources) receive 0
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
#define DPP_QUAD_XOR1 0xB1        // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E        // quad_perm:[2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142     // lane 15 of every row -> all lanes of the next row
#define DPP_ROW_BCAST31 0x143     // lane 31 -> all lanes of rows 2 and 3

__device__ __forceinline__ float wave_sum(float v)         // sum over the 64 lanes, returned in every lane
{
  v += dpp_f<DPP_QUAD_XOR1, 0xf>(v);
  v += dpp_f<DPP_QUAD_XOR2, 0xf>(v);
  v += dpp_f<DPP_ROW_HALF_MIRROR, 0xf>(v);
  v += dpp_f<DPP_ROW_MIRROR, 0xf>(v);                     // every lane: sum of its 16-lane row
  v += dpp_f<DPP_ROW_BCAST15, 0xa>(v);                    // rows 1, 3 += row 0, 2
  v += dpp_f<DPP_ROW_BCAST31, 0xc>(v);                    // rows 2, 3 += rows 0 + 1: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// maximum over lanes 0..31 of non-negative values (lanes 32..63 must hold 0), returned in every lane
__device__ __forceinline__ float wave_max32_nonneg(float v)
{
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR1, 0xf>(v));
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR2, 0xf>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR, 0xf>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_MIRROR, 0xf>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_BCAST15, 0xa>(v));           // row 1 = max(row 0, row 1): lane 31 holds the maximum of 0..31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
}

// ---------------------------------------------------------------- LDS-staged image patch
// r01 read orient_all / descr_all as bound by the L1's one-cache-line-per-clock lookup rate (every one of a
// descriptor's 2048 texel-pair gathers, 338 for an orientation, is its own L1 lookup); the r02 instruction-cost
// measurements say they are bound by VALU instruction issue, and what the window buys is the
// address arithmetic and clamping of the gathers.  The patch a keypoint samples is small and square: it is fetched ONCE with row-contiguous
// loads (13 dwordx2 per lane for the descriptor's 40x40 window, 4 dwords for the orientation's 16x16: ~15x fewer
// lookups), parked in a wave-private LDS tile with clamp-to-edge already applied (so border keypoints need no
// selects either), and all bilinear fetches read the tile: one address, two ds_read2_b32 (offsets {0,1} and
// {W,W+1}).  Same texels, same weights, same fmaf chain as tex2d() — bit-identical results.  The next keypoint's
// tile is loaded into registers while the current one is being sampled (its Detection record one step earlier
// still), so the global-memory latency is off the critical path.
#define PW 40                               // descriptor window: 40 x 40 texels around floor(xpos), floor(ypos)
#define PATCH_FLOATS (PW * PW)              // 1600 = 25 texels per lane
#define PATCH_LOADS 25
#define PATCH_REACH 17.9f                   // largest sample distance (in texels) the window covers, see patch_geom()
#define OW 16                               // orientation window: 16 x 16 texels

struct PatchGeom {
  int x0, y0;            // window origin (texel coordinates of tile[0][0])
  bool fits;             // every texel the descriptor touches lies inside the window
};

// The descriptor samples tex(x, y) with |x - xpos|, |y - ypos| <= 7.5*sqrt(2)*S + 1 (rotated 16x16 grid of spacing
// S = 0.75*scale, +-1 for the central differences; the +0.5 of the sample position cancels against tex2d's -0.5), so
// the texel columns run from floor(xpos - reach) to floor(xpos + reach) + 1.  With reach <= 17.9 that is inside
// [floor(xpos) - 18, floor(xpos) + 19], one column short of the window [floor(xpos) - 19, floor(xpos) + 20] on either
// side (slack for the few ulps the float coordinate arithmetic can move a sample).  Larger scales (scale > 2.12:
// only reachable through the refinement's unclamped fallback step) take the global-memory path.
__device__ __forceinline__ PatchGeom patch_geom(float xpos, float ypos, float pscale, int w, int h, float max_reach)
{
  PatchGeom g;
  const float reach = 10.6067f * (0.75f * pscale) + 1.0f;
  // coordinates far outside the image cannot come out of the refinement; guard the int conversion anyway
  const bool sane = xpos > -64.0f && ypos > -64.0f && xpos < (float)(w + 64) && ypos < (float)(h + 64);
  g.fits = sane && reach <= max_reach;               // (max_reach <= PATCH_REACH: PyramidInfo.patch_reach)
  g.x0 = (int)floorf(sane ? xpos : 0.0f) - 19;
  g.y0 = (int)floorf(sane ? ypos : 0.0f) - 19;
  return g;
}

// Texel t = lane + 64*k of the window is (row t / 40, column t % 40); 320 = 8 rows, so k = 5m + j is row r_j + 8m,
// column c_j with five lane constants (r_j, c_j).  Clamp-to-edge is applied here, once per texel, instead of in every
// bilinear fetch.  A load instruction covers 64 consecutive texels = 1.6 rows: 4..6 cache lines.
#ifndef PATCH_LANE_MUL24
#define PATCH_LANE_MUL24 1
#endif
struct PatchLane { int r[5], c[5]; };
__device__ __forceinline__ PatchLane patch_lane(int lane)
{
  PatchLane pl;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int t = lane + 64 * j;
#if PATCH_LANE_MUL24
    // t / 40 for t < 1600 as a full-rate 24-bit multiply and a shift (1639 / 65536 = 1/40 + 9.2e-6: exact below 2 730); the
    // compiler's own division by a constant is v_mul_hi_i32 + v_mad_u64_u32, both quarter-rate (r06: 10 per keypoint)
    static_assert(PW == 40, "1639 / 65536 stands for 1 / 40");
    pl.r[j] = (int)(__umul24((unsigned)t, 1639u) >> 16);
    pl.c[j] = __mul24(pl.r[j], -PW) + t;                           // v_mad_i32_i24
#else
    pl.r[j] = t / PW;
    pl.c[j] = t - pl.r[j] * PW;
#endif
  }
  return pl;
}
__device__ __forceinline__ void patch_fetch(const float *img, int w, int h, int pitch, const PatchGeom &g,
                                            int lane, float (&R)[PATCH_LOADS])
{
  const PatchLane pl = patch_lane(lane);      // recomputed per keypoint (20 VALU): ten registers less across the main loop
  if (g.x0 >= 0 && g.y0 >= 0 && g.x0 + PW <= w && g.y0 + PW <= h) {
    // the whole window lies inside the image (the usual case): no clamps; the five lane offsets r_j*pitch + c_j are
    // all the vector arithmetic there is — the window origin and the 8-row steps are scalar and ride in the base
    unsigned lo[5];
#pragma unroll
    for (int j = 0; j < 5; j++) lo[j] = (__umul24((unsigned)pl.r[j], (unsigned)pitch) + (unsigned)pl.c[j]) * 4u;
#pragma unroll
    for (int m = 0; m < 5; m++) {
      const char *base = reinterpret_cast<const char *>(img + (size_t)(g.y0 + 8 * m) * pitch + g.x0);   // wave-uniform
#pragma unroll
      for (int j = 0; j < 5; j++) R[5 * m + j] = *reinterpret_cast<const float *>(base + lo[j]);
    }
    return;
  }
  unsigned col[5];
#pragma unroll
  for (int j = 0; j < 5; j++) col[j] = (unsigned)clampi(g.x0 + pl.c[j], 0, w - 1);
#pragma unroll
  for (int m = 0; m < 5; m++)
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const unsigned row = (unsigned)clampi(g.y0 + pl.r[j] + 8 * m, 0, h - 1);
      R[5 * m + j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(img) +
                                                       (__umul24(row, (unsigned)pitch) + col[j]) * 4u);   // saddr + 32-bit voffset
    }
}
__device__ __forceinline__ void patch_store(float *tile, int lane, const float (&R)[PATCH_LOADS])
{
#pragma unroll
  for (int k = 0; k < PATCH_LOADS; k++) tile[lane + 64 * k] = R[k];     // row-major, row stride PW
}

// tex2d() on a staged tile of row stride TW whose element [0][0] is texel (x0, y0): same operation sequence, the
// clamp-to-edge addressing is already in the tile's contents.
#ifndef TILE_FRACT
#define TILE_FRACT 1
#endif
#ifndef TILE_MAD24
#define TILE_MAD24 1
#endif
// `porg` = tile - (y0 * TW + x0): the address of texel (0, 0) of the level, so that a fetch at integer (ifx, ify) is
// porg[ify * TW + ifx] — one v_mad_i32_i24 and one v_lshl_add_u32 per fetch instead of two subtractions, a shift, a
// multiply and a three-operand add (r03: -3 of 27 instructions per bilinear fetch, 1 024 fetches per descriptor).
template <int TW>
__device__ __forceinline__ const float *tile_origin(const float *tile, int x0, int y0) { return tile - (__mul24(y0, TW) + x0); }
template <int TW>
__device__ __forceinline__ float tex2d_tile(const float *porg, float x, float y, bool frac8)
{
  const float xb = x - 0.5f, yb = y - 0.5f;
#if TILE_FRACT
  // v_fract_f32 = xb - floor(xb) exactly (it differs from the subtraction only for xb in (-3e-8, 0), where the
  // subtraction rounds up to 1.0: a coordinate no sample of a window inside the image can have), and the floor then
  // feeds nothing but the integer conversion (v_cvt_flr_i32_f32): two instructions per coordinate instead of three
  float a = __builtin_amdgcn_fractf(xb), b = __builtin_amdgcn_fractf(yb);
  int ifx, ify;                                   // (int)floorf(.) in one instruction; the compiler does not form it
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ifx) : "v"(xb));
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ify) : "v"(yb));
#else
  const float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  const int ifx = (int)fx, ify = (int)fy;
#endif
  if (frac8) {                                    // 256 x the weights, ties to even: see tex2d() in common.hpp
    a = __builtin_fmaf(a, 256.0f, 12582912.0f) - 12582912.0f;
    b = __builtin_fmaf(b, 256.0f, 12582912.0f) - 12582912.0f;
  }
#if TILE_MAD24
  int tidx;                    // left to itself the compiler forms ify * (4 TW) + (ifx << 2) + porg: three instructions
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(tidx) : "v"(ify), "n"(TW), "v"(ifx));
  const float *p = porg + tidx;
#else
  const float *p = porg + (__mul24(ify, TW) + ifx);
#endif
  const float t00 = p[0], t10 = p[1], t01 = p[TW], t11 = p[TW + 1];
  const float one = frac8 ? 256.0f : 1.0f;
  const float ia = one - a, ib = one - b;
  float v = (ia * ib) * t00;
  v = __builtin_fmaf(a * ib, t10, v);
  v = __builtin_fmaf(ia * b, t01, v);
  v = __builtin_fmaf(a * b, t11, v);
  return frac8 ? v * (1.0f / 65536.0f) : v;
}

// ---------------------------------------------------- written-out elementary functions
// atan2f / expf as explicit fmaf chains, IDENTICAL to oracle det_atan2() / det_exp() (sift_oracle.c): the orientation
// histogram takes hard decisions on them (bin of a sample, which bins are peaks), and two libm's that agree to an ulp
// still flip such a decision once in a few hundred keypoints.  ~1-2 ulp like the CUDA libm the reference calls.
__device__ __forceinline__ float det_atan2(float y, float x)
{
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = mx == 0.0f ? 0.0f : mn / mx;
  const bool red = a > 0.414213562f;
  const float base = red ? 0.785398163f : 0.0f;
  a = red ? (a - 1.0f) / (a + 1.0f) : a;
  const float z = a * a;
  float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = __builtin_fmaf(p, z, 1.99777106478e-1f);
  p = __builtin_fmaf(p, z, -3.33329491539e-1f);
  float r = base + __builtin_fmaf(p * z, a, a);
  r = ay > ax ? 1.57079637f - r : r;
  r = x < 0.0f ? 3.14159274f - r : r;
  return y < 0.0f ? -r : r;
}
__device__ __forceinline__ float det_exp(float x)
{
  const bool tiny = x < -87.0f;
  x = fminf(x, 88.0f);
  x = tiny ? 0.0f : x;
  const float n = rintf(x * 1.44269504f);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = __builtin_fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  const float e = __builtin_fmaf(p * r, r, r) + 1.0f;
  const float sc = __builtin_bit_cast(float, ((int)n + 127) << 23);
  return tiny ? 0.0f : e * sc;
}

// ------------------------------------------------------------- orientation
struct OrientResult { float ori1, ori2; bool has2; };      // meaningful in lane 0 only

// Second half of the orientation (reference cudaSiftD.cu:1011-1037): histogram of the 121 (bin, weight) samples in
// smp[], smoothing, the two best local maxima, parabolic peaks.  Result meaningful in lane 0.
__device__ __forceinline__ OrientResult orient_finish(float *hist, const float2 *smp, int lane)
{
  wave_sync();
  // privatized histogram (no LDS atomics): lane (b, half) sums the samples of its half that fall in bin b
  {
    const float fb = (float)(lane & 31);
    const float2 *sp = smp + (lane >> 5) * 64;
    float acc = 0.0f;
    // acc += (e.x == fb) ? e.y : 0 as v_cmpx (the comparison IS the execution mask) + a masked v_add + one scalar move
    // that restores the mask: 2 vector + 1 scalar instruction per sample instead of compare, select, add (r03: 64 of
    // the ~480 vector instructions of an orientation).  Adding under the mask = adding 0 (weights are >= 0, acc starts +0).
    const unsigned long long all_lanes = __builtin_amdgcn_read_exec();
#pragma unroll 1
    for (int jj = 0; jj < 64; jj += 16) {
      float4 e[8];                                   // 16 samples per trip, loaded before the (unschedulable) asm blocks
#pragma unroll
      for (int k = 0; k < 8; k++) e[k] = reinterpret_cast<const float4 *>(sp + jj)[k];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        asm volatile("v_cmpx_eq_f32_e32 vcc, %1, %2\n\tv_add_f32_e32 %0, %0, %3\n\ts_mov_b64 exec, %4"
                     : "+v"(acc) : "v"(e[k].x), "v"(fb), "v"(e[k].y), "s"(all_lanes) : "vcc");
        asm volatile("v_cmpx_eq_f32_e32 vcc, %1, %2\n\tv_add_f32_e32 %0, %0, %3\n\ts_mov_b64 exec, %4"
                     : "+v"(acc) : "v"(e[k].z), "v"(fb), "v"(e[k].w), "s"(all_lanes) : "vcc");
      }
    }
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 32) hist[lane] = acc;
  }
  wave_sync();
  const int t = lane & 31;
  const int x1m = (t >= 1 ? t - 1 : t + 31), x1p = (t <= 30 ? t + 1 : t - 31);
  const int x2m = (t >= 2 ? t - 2 : t + 30), x2p = (t <= 29 ? t + 2 : t - 30);
  if (lane < 32) hist[t + 32] = 6.0f * hist[t] + 4.0f * (hist[x1m] + hist[x1p]) + (hist[x2m] + hist[x2p]);
  wave_sync();
  // non-maximum suppression, then the two largest peaks by wave reductions (the reference scans the 32 bins
  // serially, cudaSiftD.cu:1020-1033: first index of the maximum, first index of the runner-up)
  float pk = 0.0f;
  if (lane < 32) {
    const float v = hist[32 + t];
    pk = (v > hist[32 + x1m] && v >= hist[32 + x1p] ? v : 0.0f);
  }
  const float maxval1 = wave_max32_nonneg(pk);            // pk >= 0 everywhere, 0 in lanes 32..63
  const unsigned long long b1 = __ballot(lane < 32 && pk == maxval1 && maxval1 > 0.0f);
  const int i1 = b1 ? __ffsll((long long)b1) - 1 : -1;
  const float pk2 = lane == i1 ? 0.0f : pk;
  const float maxval2 = wave_max32_nonneg(pk2);
  const unsigned long long b2 = __ballot(lane < 32 && pk2 == maxval2 && maxval2 > 0.0f);
  const int i2 = b2 ? __ffsll((long long)b2) - 1 : -1;
  OrientResult r;
  r.ori1 = 0.0f; r.ori2 = 0.0f; r.has2 = false;
  if (lane == 0) {
    if (i1 >= 0) {                                      // empty histogram -> orientation 0
      const float val1 = hist[32 + ((i1 + 1) & 31)];
      const float val2 = hist[32 + ((i1 + 31) & 31)];
      const float peak = i1 + 0.5f * (val1 - val2) / (2.0f * maxval1 - val1 - val2);
      r.ori1 = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
      if (maxval2 > 0.8f * maxval1) {
        const float v1 = hist[32 + ((i2 + 1) & 31)];
        const float v2 = hist[32 + ((i2 + 31) & 31)];
        const float peak2 = i2 + 0.5f * (v1 - v2) / (2.0f * maxval2 - v1 - v2);
        r.ori2 = 11.25f * (peak2 < 0.0f ? peak2 + 32.0f : peak2);
        r.has2 = true;
      }
    }
  }
  wave_sync();
  return r;
}

// Orientation of one keypoint by one wavefront (reference cudaSiftD.cu:984-1037).  hist[64], gauss[16] and
// smp[128], tgrid[169] are wave-private LDS slices; smp[121..127] must hold bin -1 (never matches).
__device__ __forceinline__ OrientResult orient_core(const float *img, int w, int h, int pitch, bool q8, float xpos,
                                                    float ypos, float scale, float *hist, float *gauss,
                                                    float2 *smp, float *tgrid, int lane)
{
  const float i2sigma2 = -1.0f / (2.0f * 1.5f * 1.5f * scale * scale);
  if (lane < 11) gauss[lane] = det_exp(i2sigma2 * (lane - 5) * (lane - 5));
  wave_sync();
  const float xp = xpos - 4.5f;
  const float yp = ypos - 4.5f;
  // The 121 samples sit on an integer grid and take central differences of bilinear fetches one pixel to
  // either side (cudaSiftD.cu:1003-1010): all 484 fetches are values of ONE 13x13 grid T[gy][gx] =
  // tex(xp + gx, yp + gy), gx, gy = -1..11.  Evaluate the 169 grid values once (3 fetches per lane instead of
  // 8) and difference them out of LDS.  The reference computes the coordinates as (xp + xd) +- 1, which
  // differs from xp + (xd +- 1) in the last bit when a sum crosses a binade; that is checked here (20 + 20
  // float comparisons) and such keypoints take the literal per-sample path, so results stay bit-identical.
  bool same = true;
  if (lane < 40) {
    const float base = lane < 20 ? xp : yp;
    const int k = lane < 20 ? lane : lane - 20;
    if (k < 10) same = ((base + (float)k) + 1.0f) == (base + (float)(k + 1));          // xd = 0..9, "+1"
    else same = ((base + (float)(k - 9)) - 1.0f) == (base + (float)(k - 10));           // xd = 1..10, "-1"
  }
  if (__all(same)) {
    // grid spans xpos-5.5 .. xpos+6.5: fetches need no clamping when that (plus the bilinear footprint) is inside
    const bool inside = xpos - 7.0f >= 1.0f && xpos + 8.0f <= (float)(w - 2) && ypos - 7.0f >= 1.0f &&
                        ypos + 8.0f <= (float)(h - 2);
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
      const int id = lane + 64 * rep;
      if (id < 169) {
        const int gy = id / 13, gx = id - gy * 13;          // grid index + 1
        // the outermost ring is reached only as (xp + 10) + 1 resp. (xp + 0) - 1
        const float xf = gx == 0 ? (xp + 0.0f) - 1.0f : (gx == 12 ? (xp + 10.0f) + 1.0f : xp + (float)(gx - 1));
        const float yf = gy == 0 ? (yp + 0.0f) - 1.0f : (gy == 12 ? (yp + 10.0f) + 1.0f : yp + (float)(gy - 1));
        tgrid[id] = inside ? tex2d<true>(img, w, h, pitch, xf, yf, q8) : tex2d<false>(img, w, h, pitch, xf, yf, q8);
      }
    }
    wave_sync();
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
      const int tx = lane + 64 * rep;
      if (tx < 121) {
        const int yd = tx / 11;
        const int xd = tx - yd * 11;
        const float *t = tgrid + (yd + 1) * 13 + (xd + 1);
        const float dx = t[1] - t[-1];
        const float dy = t[13] - t[-13];
        int bin = (int)(16.0f * det_atan2(dy, dx) / 3.1416f + 16.5f);
        if (bin > 31) bin = 0;
        const float grad = sqrtf(dx * dx + dy * dy);
        smp[tx] = make_float2((float)bin, grad * gauss[xd] * gauss[yd]);
      }
    }
  } else {
#pragma unroll 1
    for (int rep = 0; rep < 2; rep++) {
      const int tx = lane + 64 * rep;
      if (tx < 121) {
        const int yd = tx / 11;
        const int xd = tx - yd * 11;
        const float xf = xp + xd;
        const float yf = yp + yd;
        const float dx = tex2d(img, w, h, pitch, xf + 1.0f, yf, q8) - tex2d(img, w, h, pitch, xf - 1.0f, yf, q8);
        const float dy = tex2d(img, w, h, pitch, xf, yf + 1.0f, q8) - tex2d(img, w, h, pitch, xf, yf - 1.0f, q8);
        int bin = (int)(16.0f * det_atan2(dy, dx) / 3.1416f + 16.5f);
        if (bin > 31) bin = 0;
        const float grad = sqrtf(dx * dx + dy * dy);
        smp[tx] = make_float2((float)bin, grad * gauss[xd] * gauss[yd]);
      }
    }
  }
  return orient_finish(hist, smp, lane);
}

// Orientation from a staged 16x16 window (tile[0][0] = texel (x0, y0), clamp-to-edge applied when it was loaded):
// the 169 grid values are bilinear fetches from LDS.  x0 = floor(xpos) - 7: the grid spans tex(x) with
// x - 0.5 in [xpos - 6, xpos + 6] (up to a rounding of the float sums, which can only move a coordinate ONTO the next
// integer, never across it), so the texel columns lie in [floor(xpos) - 6, floor(xpos) + 8] — inside
// [x0, x0 + 15] with a column to spare on the left.  Keypoints whose coordinate sums are not binade-safe take the
// literal per-sample path from global memory (orient_core's second branch) — bit-identical either way.
__device__ __forceinline__ OrientResult orient_core_tile(const float *img, int w, int h, int pitch, bool q8, float xpos,
                                                         float ypos, float scale, const float *tile, int x0, int y0,
                                                         float *hist, float *gauss, float2 *smp, float *tgrid, int lane)
{
  const float i2sigma2 = -1.0f / (2.0f * 1.5f * 1.5f * scale * scale);
  if (lane < 11) gauss[lane] = det_exp(i2sigma2 * (lane - 5) * (lane - 5));
  const float xp = xpos - 4.5f;
  const float yp = ypos - 4.5f;
  bool same = true;
  if (lane < 40) {
    const float base = lane < 20 ? xp : yp;
    const int k = lane < 20 ? lane : lane - 20;
    if (k < 10) same = ((base + (float)k) + 1.0f) == (base + (float)(k + 1));          // xd = 0..9, "+1"
    else same = ((base + (float)(k - 9)) - 1.0f) == (base + (float)(k - 10));           // xd = 1..10, "-1"
  }
  if (!__all(same))
    return orient_core(img, w, h, pitch, q8, xpos, ypos, scale, hist, gauss, smp, tgrid, lane);
#pragma unroll
  for (int rep = 0; rep < 3; rep++) {
    const int id = lane + 64 * rep;
    if (id < 169) {
      const int gy = id / 13, gx = id - gy * 13;          // grid index + 1
      const float xf = gx == 0 ? (xp + 0.0f) - 1.0f : (gx == 12 ? (xp + 10.0f) + 1.0f : xp + (float)(gx - 1));
      const float yf = gy == 0 ? (yp + 0.0f) - 1.0f : (gy == 12 ? (yp + 10.0f) + 1.0f : yp + (float)(gy - 1));
      tgrid[id] = tex2d_tile<OW>(tile_origin<OW>(tile, x0, y0), xf, yf, q8);
    }
  }
  wave_sync();
#pragma unroll
  for (int rep = 0; rep < 2; rep++) {
    const int tx = lane + 64 * rep;
    if (tx < 121) {
      const int yd = tx / 11;
      const int xd = tx - yd * 11;
      const float *t = tgrid + (yd + 1) * 13 + (xd + 1);
      const float dx = t[1] - t[-1];
      const float dy = t[13] - t[-13];
      int bin = (int)(16.0f * det_atan2(dy, dx) / 3.1416f + 16.5f);
      if (bin > 31) bin = 0;
      const float grad = sqrtf(dx * dx + dy * dy);
      smp[tx] = make_float2((float)bin, grad * gauss[xd] * gauss[yd]);
    }
  }
  return orient_finish(hist, smp, lane);
}

template <bool Q8>
__global__ __launch_bounds__(256) void orient_kernel(const float *__restrict__ base, long long base_frame_stride,
                                                     int w, int h, int pitch, int octave,
                                                     unsigned *__restrict__ counters, SiftPointD *__restrict__ pts,
                                                     int max_pts, int frac8)
{
  __shared__ float s_hist[WAVES_PER_BLOCK][64];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  __shared__ float2 s_smp[WAVES_PER_BLOCK][128];     // (bin, weight) of the 121 samples
  __shared__ float s_tgrid[WAVES_PER_BLOCK][176];   // 13x13 grid of bilinear fetches
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  const float *img = base + (long long)frame * base_frame_stride;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  SiftPointD *sift = pts + (size_t)frame * max_pts;
  const bool q8 = Q8;          // compile-time: no per-fetch branch on the weight quantisation

  const int fstPts = (int)min(cnt[2 * octave - 1], (unsigned)max_pts);
  const int totPts = (int)min(cnt[2 * octave + 0], (unsigned)max_pts);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&cnt[2 * octave + 1], cnt[2 * octave + 0]);
  if (lane >= 57) s_smp[wave][64 + lane] = make_float2(-1.0f, 0.0f);     // slots 121..127 never match a bin

  for (int bx = fstPts + blockIdx.x * WAVES_PER_BLOCK + wave; bx < totPts; bx += gridDim.x * WAVES_PER_BLOCK) {
    const OrientResult r = orient_core(img, w, h, pitch, q8, sift[bx].xpos, sift[bx].ypos, sift[bx].scale,
                                       s_hist[wave], s_gauss[wave], s_smp[wave], s_tgrid[wave], lane);
    if (lane == 0) {
      sift[bx].orientation = r.ori1;
      if (r.has2) {                                     // duplicate with the second orientation (cudaSiftD.cu:1038-1052)
        atomicMax(&cnt[2 * octave + 1], cnt[2 * octave + 0]);
        const unsigned idx = atomicAdd(&cnt[2 * octave + 1], 1u);
        if (idx < (unsigned)max_pts) {
          sift[idx].xpos = sift[bx].xpos;
          sift[idx].ypos = sift[bx].ypos;
          sift[idx].scale = sift[bx].scale;
          sift[idx].sharpness = sift[bx].sharpness;
          sift[idx].edgeness = sift[bx].edgeness;
          sift[idx].orientation = r.ori2;
          sift[idx].subsampling = sift[bx].subsampling;
        } else {
          atomicAdd(&cnt[CNT_PTOVF], 1u);
        }
      }
    }
  }
}

// -------------------------------------------------------------- descriptors
// sin / cos of the descriptor rotation, theta in [0, 2 pi]: the SAME written-out expression as oracle det_sincos()
// (Cody-Waite reduction by pi/2, cephes minimax kernels as fmaf chains; 1 ulp) — so the sample coordinates, and with
// them every 8-bit texture weight, are bit-identical to the oracle's; ~25 VALU instead of the ~150 of sinf + cosf.
__device__ __forceinline__ void det_sincos(float x, float &sn, float &cs)
{
  const float kf = rintf(x * 0.636619747f);
  float r = __builtin_fmaf(kf, -1.57079625f, x);
  r = __builtin_fmaf(kf, -7.54978942e-08f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  const float s = __builtin_fmaf(ps * z, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  const float c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const int q = (int)kf & 3;
  const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  sn = (q & 2) ? -s1 : s1;
  cs = ((q + 1) & 2) ? -c1 : c1;
}

__device__ __forceinline__ float fast_atan2(float y, float x)
{
  const float absx = fabsf(x), absy = fabsf(y);
  const float mx = fmaxf(absx, absy), mn = fminf(absx, absy);
  const float a = mx == 0.0f ? 0.0f : mn / mx;         // (0,0) -> angle 0, as a select: no divergent branch
  const float s = a * a;
  float r = ((-0.0464964749f * s + 0.15931422f) * s - 0.327622764f) * s * a + a;
  r = (absy > absx ? 1.57079637f - r : r);
  r = (x < 0 ? 3.14159274f - r : r);
  r = (y < 0 ? -r : r);
  return r;
}

// Descriptor accumulation without LDS atomics and without searching.
// Phase 1: every lane evaluates 4 of the 256 rotated samples; a sample votes iangf*grad into angle bin
// angi and angf*grad into bin angi+1 (mod 8) of the (up to) 2x2 cells around it (cudaSiftD.cu:346-386).
// Phase 2 is output-centric: lane = (cell c = lane>>2, angle bins a and a+4).  The votes are laid out in
// LDS as per-bin planes [bin][20][20] over the 16x16 sample grid with a 2-sample zero border, so a lane
// reads the 8x8 footprint of its cell in ITS bin's plane with 16 b128 loads and sums it with 64 FMAs
// whose spatial weights are compile-time literals — no compares, no selects (the previous version
// searched (angi, vote) records: 770 VALU instead of 130 per keypoint-lane; rocprof showed the kernel
// VALU-bound at 4 waves/SIMD).  Only four planes exist: bins 0..3 are accumulated first, then the same
// four planes are re-used for bins 4..7; every slot a lane writes it zeroes again afterwards, so the
// table stays all-zero between passes and keypoints.  Same votes and the same summation order as before.
// A fifth plane takes the rare angi == 8 votes (dy == +0, dx < 0).
#define SMP_W 20
#define SMP_PLANE (SMP_W * SMP_W)        // 400 floats: plane stride / 4 = 100 = 4 (mod 16) -> conflict-free b128 reads
#define DESCR_TBL (5 * SMP_PLANE)
__device__ __forceinline__ constexpr float spatial_w(int m)      // horf/verf for m<4, 1-horf/1-verf for m>=4
{
  return m < 4 ? (m + 0.5f) * 0.25f : (7.5f - m) * 0.25f;
}

__device__ __forceinline__ void descr_init(float *tbl, float *gauss, int lane)
{
  if (lane < 16) gauss[lane] = det_exp(-(lane - 7.5f) * (lane - 7.5f) / 128.0f);
  for (int i = lane; i < DESCR_TBL; i += 64) tbl[i] = 0.0f;
}

// 8x8 footprint of a cell in one plane: rows are 20 floats apart, the two b128 loads of a row are adjacent
__device__ __forceinline__ float footprint_sum(const float *base, float acc)
{
#pragma unroll
  for (int my = 0; my < 8; my++) {
    const float4 lo = *reinterpret_cast<const float4 *>(base + my * SMP_W);
    const float4 hi = *reinterpret_cast<const float4 *>(base + my * SMP_W + 4);
    const float wy = spatial_w(my);
    acc = __builtin_fmaf(wy * spatial_w(0), lo.x, acc);
    acc = __builtin_fmaf(wy * spatial_w(1), lo.y, acc);
    acc = __builtin_fmaf(wy * spatial_w(2), lo.z, acc);
    acc = __builtin_fmaf(wy * spatial_w(3), lo.w, acc);
    acc = __builtin_fmaf(wy * spatial_w(4), hi.x, acc);
    acc = __builtin_fmaf(wy * spatial_w(5), hi.y, acc);
    acc = __builtin_fmaf(wy * spatial_w(6), hi.z, acc);
    acc = __builtin_fmaf(wy * spatial_w(7), hi.w, acc);
  }
  return acc;
}

// Phase 1 of the descriptor: this lane's 4 of the 256 rotated samples -> votes and their table slots.
template <bool INTERIOR>
__device__ __forceinline__ void descr_samples(const float *img, int w, int h, int pitch, bool q8, float px, float py,
                                              float sina, float cosa, float ssina, float scosa, const float *gauss,
                                              int lane, float (&vx)[4], float (&vy)[4], int (&slotx)[4],
                                              int (&sloty)[4], bool &has8)
{
  // two samples per trip of a rolled loop: 16 gathers in flight per lane instead of 32 keeps the kernel at
  // 128 VGPRs (4 waves/SIMD); the results are moved into the named slots of the trip
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    float tvx[2], tvy[2];
    int tsx[2], tsy[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int id = lane + 64 * (2 * half + j);
      const int tx = id & 15, y = id >> 4;
      const float xpos = px + (tx - 7.5f) * scosa - (y - 7.5f) * ssina + 0.5f;
      const float ypos = py + (tx - 7.5f) * ssina + (y - 7.5f) * scosa + 0.5f;
      const float dx = tex2d<INTERIOR>(img, w, h, pitch, xpos + cosa, ypos + sina, q8) -
                       tex2d<INTERIOR>(img, w, h, pitch, xpos - cosa, ypos - sina, q8);
      const float dy = tex2d<INTERIOR>(img, w, h, pitch, xpos - sina, ypos + cosa, q8) -
                       tex2d<INTERIOR>(img, w, h, pitch, xpos + sina, ypos - cosa, q8);
      const float grad = gauss[y] * gauss[tx] * sqrtf(dx * dx + dy * dy);
      float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
      const int angi = (int)angf;
      angf -= angi;
      const float iangf = 1.0f - angf;
      has8 |= angi >= 8;
      tvx[j] = iangf * grad;
      tvy[j] = angf * grad;
      const int pos = (y + 2) * SMP_W + tx + 2;
      tsx[j] = angi * SMP_PLANE + pos;                       // angi == 8: the special fifth plane
      tsy[j] = (angi >= 7 ? 0 : angi + 1) * SMP_PLANE + pos; // angi+1 wraps to bin 0 (7 and 8 alike)
    }
    if (half == 0) {
      vx[0] = tvx[0]; vy[0] = tvy[0]; slotx[0] = tsx[0]; sloty[0] = tsy[0];
      vx[1] = tvx[1]; vy[1] = tvy[1]; slotx[1] = tsx[1]; sloty[1] = tsy[1];
    } else {
      vx[2] = tvx[0]; vy[2] = tvy[0]; slotx[2] = tsx[0]; sloty[2] = tsy[0];
      vx[3] = tvx[1]; vy[3] = tvy[1]; slotx[3] = tsx[1]; sloty[3] = tsy[1];
    }
  }
}

// Normalised descriptor bins (8*cell + (lane&3)) and (+4) of one keypoint, cell = lane >> 2.
__device__ __forceinline__ void descr_core(const float *img, int w, int h, int pitch, bool q8, float px, float py,
                                           float pscale, float orientation, float *tbl, const float *gauss,
                                           int lane, float &out0, float &out1)
{
  const int cell = lane >> 2, cx = cell & 3, cy = cell >> 2;
  // this lane's plane (bin lane&3, then bin (lane&3)+4) and the top-left slot of its cell's footprint:
  // table slot of sample (tx,y) is (y+2)*20 + tx+2, the footprint of cell (cx,cy) starts at sample (4cx-2, 4cy-2)
  const float *mine = tbl + (lane & 3) * SMP_PLANE + (4 * cy) * SMP_W + 4 * cx;
  const float theta = 2.0f * 3.1415f / 360.0f * orientation;
  float sina, cosa;
  det_sincos(theta, sina, cosa);
  const float scale = 12.0f / 16.0f * pscale;
  const float ssina = scale * sina;
  const float scosa = scale * cosa;
  float vx[4], vy[4];
  int slotx[4], sloty[4];         // plane-relative slot (bin * SMP_PLANE + position) of the two votes, bins 0..7 (8: special)
  bool has8 = false;
  // all 1024 texels of the patch inside the image (the usual case): fetches without clamping or edge selects
  const float reach = 10.6067f * scale + 2.5f;
  const bool interior = px - reach >= 1.0f && px + reach <= (float)(w - 2) && py - reach >= 1.0f &&
                        py + reach <= (float)(h - 2);
  if (interior) descr_samples<true>(img, w, h, pitch, q8, px, py, sina, cosa, ssina, scosa, gauss, lane, vx, vy, slotx, sloty, has8);
  else descr_samples<false>(img, w, h, pitch, q8, px, py, sina, cosa, ssina, scosa, gauss, lane, vx, vy, slotx, sloty, has8);
  // ---- bins 0..3
#pragma unroll
  for (int rep = 0; rep < 4; rep++) {
    if (slotx[rep] < 4 * SMP_PLANE) tbl[slotx[rep]] = vx[rep];
    if (sloty[rep] < 4 * SMP_PLANE) tbl[sloty[rep]] = vy[rep];
  }
  wave_sync();
  float acc0 = footprint_sum(mine, 0.0f);
  wave_sync();
  // ---- bins 4..7 re-use planes 0..3 (a lane's pass-A and pass-B slots never coincide: different plane or
  // position).  The rare angi == 8 votes land in the fifth plane here (slot - 4 planes = plane 4).
#pragma unroll
  for (int rep = 0; rep < 4; rep++) {
    if (slotx[rep] < 4 * SMP_PLANE) tbl[slotx[rep]] = 0.0f;
    if (sloty[rep] < 4 * SMP_PLANE) tbl[sloty[rep]] = 0.0f;
    if (slotx[rep] >= 4 * SMP_PLANE) tbl[slotx[rep] - 4 * SMP_PLANE] = vx[rep];
    if (sloty[rep] >= 4 * SMP_PLANE) tbl[sloty[rep] - 4 * SMP_PLANE] = vy[rep];
  }
  wave_sync();
  float acc1 = footprint_sum(mine, 0.0f);
  if (__any(has8)) {
    // rare (dy == +0 and dx < 0): angi == 8 makes the iangf vote land in bin 0 of the
    // NEXT cell of the flattened 4x4 grid, with the spatial weights of the cell it was computed for; cell 16
    // does not exist (dropped).  Sum plane 4 over the PREVIOUS cell's footprint.
    if ((lane & 3) == 0 && cell >= 1) {
      const int pc = cell - 1, pcx = pc & 3, pcy = pc >> 2;
      acc0 = footprint_sum(tbl + 4 * SMP_PLANE + (4 * pcy) * SMP_W + 4 * pcx, acc0);
    }
  }
  wave_sync();
#pragma unroll
  for (int rep = 0; rep < 4; rep++) {
    if (slotx[rep] >= 4 * SMP_PLANE) tbl[slotx[rep] - 4 * SMP_PLANE] = 0.0f;
    if (sloty[rep] >= 4 * SMP_PLANE) tbl[sloty[rep] - 4 * SMP_PLANE] = 0.0f;
  }
  // normalise, clamp at 0.2, normalise again (reference cudaSiftD.cu:390-409)
  const float tsum1 = wave_sum(acc0 * acc0 + acc1 * acc1);
  const float rs1 = 1.0f / sqrtf(tsum1);
  const float c0 = fminf(acc0 * rs1, 0.2f), c1 = fminf(acc1 * rs1, 0.2f);
  const float tsum2 = wave_sum(c0 * c0 + c1 * c1);
  const float rs2 = 1.0f / sqrtf(tsum2);
  out0 = c0 * rs2;
  out1 = c1 * rs2;
}

template <bool Q8>
__global__ __launch_bounds__(256) void descr_kernel(const float *__restrict__ base, long long base_frame_stride,
                                                    int w, int h, int pitch, float subsampling, int octave,
                                                    const unsigned *__restrict__ counters,
                                                    SiftPointD *__restrict__ pts, int max_pts, int frac8)
{
  __shared__ __attribute__((aligned(16))) float s_smp[WAVES_PER_BLOCK][DESCR_TBL];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  const float *img = base + (long long)frame * base_frame_stride;
  const unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  SiftPointD *sift = pts + (size_t)frame * max_pts;
  const bool q8 = Q8;          // compile-time: no per-fetch branch on the weight quantisation
  descr_init(s_smp[wave], s_gauss[wave], lane);
  const int cell = lane >> 2;

  const int fstPts = (int)min(cnt[2 * octave - 1], (unsigned)max_pts);
  const int totPts = (int)min(cnt[2 * octave + 1], (unsigned)max_pts);
  for (int bx = fstPts + blockIdx.x * WAVES_PER_BLOCK + wave; bx < totPts; bx += gridDim.x * WAVES_PER_BLOCK) {
    const float px = sift[bx].xpos, py = sift[bx].ypos, pscale = sift[bx].scale;
    float o0, o1;
    descr_core(img, w, h, pitch, q8, px, py, pscale, sift[bx].orientation, s_smp[wave], s_gauss[wave], lane, o0, o1);
    sift[bx].data[8 * cell + (lane & 3)] = o0;
    sift[bx].data[8 * cell + (lane & 3) + 4] = o1;
    if (lane == 0) {
      sift[bx].xpos = px * subsampling;
      sift[bx].ypos = py * subsampling;
      sift[bx].scale = pscale * subsampling;
    }
  }
}

// ------------------------------------------------- merged-octave variants
// One launch each over ALL octaves of ALL frames (instead of one per octave): detections wait in a
// per-octave staging area (Detection records written by refine_all_kernel); orient_all_kernel adds the
// orientation(s) and hands out duplicate slots; descr_all_kernel lays the final SiftPoint array out in
// the reference's segment order  [oct 1 detections | oct 1 duplicates | oct 2 detections | ...]
// (cudaSiftD.cu:1297-1300, :1038-1044) and writes the reference's 17 counters.

// The per-octave detection counts of a frame are final when these kernels run: read them ONCE per wavefront
// into (scalar) registers instead of chasing 5-10 dependent global loads per keypoint, and keep the keypoint
// index wave-uniform (readfirstlane) so that the level lookup in the kernel arguments is a scalar load.
// ---- which frame a workgroup of a per-keypoint launch works on
// Unbalanced (default): a (sub-blocks, frames) grid, every frame the same number of workgroups — a frame with three
// times the keypoints of its neighbours finishes three times later and the batch waits for it.  Balanced (BAL): a 1-D
// grid and a table block -> (frame, sub-block, sub-blocks of that frame) written by frame_shares_kernel, which deals
// the workgroups out in proportion to the frames' keypoint counts.  Within a frame the keypoints are strided over its
// sub-blocks exactly as before, so the records do not depend on the split.
struct FrameShare { int frame, sub, nsub; };
template <bool BAL> __device__ __forceinline__ FrameShare frame_share(const int4 *__restrict__ block_map)
{
  if (BAL) {
    const int4 e = block_map[blockIdx.x];            // uniform address: one scalar load
    return FrameShare{e.x, e.y, e.z};                // frame < 0: a spare workgroup of the launch
  }
  return FrameShare{(int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x};
}

struct FrameCounts {
  int ndet[MISIFT_MAX_OCTAVES + 1];           // min(count, max_pts)
  unsigned bdet[MISIFT_MAX_OCTAVES + 1];      // segment base of the octave's detections in the reference layout
  unsigned bdup[MISIFT_MAX_OCTAVES + 1];      // ... and of its duplicates
};
__device__ __forceinline__ FrameCounts load_frame_counts(const unsigned *cnt, int noct, int max_pts, bool with_dups)
{
  FrameCounts c;
  unsigned b = 0;
#pragma unroll
  for (int k = 1; k <= MISIFT_MAX_OCTAVES; k++) {
    const unsigned nd = k <= noct ? __builtin_amdgcn_readfirstlane(cnt[CNT_DET + k]) : 0u;
    const unsigned nu = (with_dups && k <= noct) ? __builtin_amdgcn_readfirstlane(cnt[CNT_DUP + k]) : 0u;
    c.ndet[k] = (int)min(nd, (unsigned)max_pts);
    c.bdet[k] = b;
    c.bdup[k] = b + nd;
    b += nd + nu;
  }
  c.ndet[0] = 0; c.bdet[0] = 0; c.bdup[0] = 0;
  return c;
}
__device__ __forceinline__ bool flat_to_octave(const FrameCounts &c, int noct, int idx, int &o, int &i)
{
  bool found = false;
  o = 0; i = idx;
#pragma unroll
  for (int k = 1; k <= MISIFT_MAX_OCTAVES; k++) {
    if (!found && k <= noct) {
      if (i < c.ndet[k]) { o = k; found = true; }
      else i -= c.ndet[k];
    }
  }
  return found;
}

// share(f) = 1 + floor((T - nframes) * n_f / sum n): every frame keeps a workgroup (descr_all's first sub-block of a
// frame publishes its counters), the shares add up to at most T, the rest of the table says "spare".
__host__ __device__ static inline unsigned keypoint_share(unsigned long long spare, unsigned long long n,
                                                          unsigned long long total)
{
  return 1u + (total ? (unsigned)(spare * n / total) : 0u);
}
// host-only test hook (no device needed): the shares frame_shares_kernel gives `nframes` frames with points[f] keypoints
// out of `nblocks` workgroups
extern "C" int misift_test_frame_shares(int nblocks, int nframes, const unsigned *points, int *shares)
{
  if (!points || !shares || nframes < 1 || nblocks < nframes) return MISIFT_EINVAL;
  unsigned long long total = 0;
  for (int f = 0; f < nframes; f++) total += points[f];
  for (int f = 0; f < nframes; f++)
    shares[f] = (int)keypoint_share((unsigned long long)(nblocks - nframes), points[f], total);
  return MISIFT_OK;
}
// The block tables of a balanced batch's orient_all / descr_all launches (see build_block_maps): ONE workgroup of 256
// threads — a kernel of its own (frame_shares_kernel), or the extra workgroup of the bin_detections launch (r05: one
// dependent dispatch less per batch; it needs nothing the binning produces, only refine_all's per-octave counts).
struct FrameSharesArgs { int t_a, t_b; int4 *map_a, *map_b; };
__device__ __forceinline__ void frame_shares_body(const unsigned *__restrict__ counters, int nframes, int noct, int max_pts,
                                                  int t_a, int4 *__restrict__ map_a, int t_b, int4 *__restrict__ map_b,
                                                  unsigned *s_scan, unsigned long long &s_total)
{
  const int tid = threadIdx.x;
  auto frame_points = [&](int f) -> unsigned {
    unsigned n = 0;
    for (int o = 1; o <= noct; o++) n += min(counters[(size_t)f * CNT_STRIDE + CNT_DET + o], (unsigned)max_pts);
    return n;
  };
  if (tid == 0) s_total = 0ull;
  __syncthreads();
  unsigned long long mine = 0;
  for (int f = tid; f < nframes; f += 256) mine += frame_points(f);
  if (mine) atomicAdd(&s_total, mine);
  __syncthreads();
  const unsigned long long total = s_total;
  for (int which = 0; which < 2; which++) {
    const int T = which ? t_b : t_a;
    int4 *map = which ? map_b : map_a;
    if (T < nframes || !map) continue;                         // (the host sizes T >= 8 * nframes)
    const unsigned long long spare = (unsigned long long)(T - nframes);
    unsigned carry = 0;
    for (int base = 0; base < nframes; base += 256) {
      const int f = base + tid;
      const unsigned share = f < nframes ? keypoint_share(spare, frame_points(f), total) : 0u;
      s_scan[tid] = share;
      __syncthreads();
      for (int d = 1; d < 256; d <<= 1) {                       // inclusive scan of the 256 shares
        const unsigned v = tid >= d ? s_scan[tid - d] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
      }
      const unsigned start = carry + s_scan[tid] - share;
      for (unsigned k = 0; k < share; k++) map[start + k] = make_int4(f, (int)k, (int)share, 0);
      carry += s_scan[255];
      __syncthreads();
    }
    for (int b = (int)carry + tid; b < T; b += 256) map[b] = make_int4(-1, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void frame_shares_kernel(const unsigned *__restrict__ counters, int nframes, int noct,
                                                           int max_pts, int t_a, int4 *__restrict__ map_a, int t_b,
                                                           int4 *__restrict__ map_b)
{
  __shared__ unsigned s_scan[256];
  __shared__ unsigned long long s_total;
  frame_shares_body(counters, nframes, noct, max_pts, t_a, map_a, t_b, map_b, s_scan, s_total);
}

// ------------------------------------------------- spatial binning of the staged detections
// refine_all appends detections in atomic order, i.e. spatially random: consecutive wavefronts of orient_all /
// descr_all then share no cache lines and every keypoint's window comes from HBM (r01 PMC: 6 KB fetched per
// keypoint, as much as the whole DoG scan).  One workgroup per (octave, frame) counting-sorts the octave's
// Detection records by 32x32-pixel tile (row-major tile order; larger tiles when a level has more than 4096 of
// them): wavefronts that run at the same time now work on neighbouring keypoints.  The order inside a tile stays
// arbitrary, like the reference's atomic append order (cudaSiftD.cu:1420).
#define BIN_MAX_TILES 4096
// total order of two detections of one tile (deterministic mode): y, then x, then scale, then sharpness — by bit pattern
__device__ __forceinline__ bool det_less(const Detection &a, const Detection &b)
{
  if (a.ypos != b.ypos) return a.ypos < b.ypos;
  if (a.xpos != b.xpos) return a.xpos < b.xpos;
  if (a.scale != b.scale) return a.scale < b.scale;
  return a.sharpness < b.sharpness;
}

__global__ __launch_bounds__(256) void bin_detections_kernel(PyramidInfo P, const unsigned *__restrict__ counters,
                                                             const Detection *__restrict__ in,
                                                             Detection *__restrict__ out, int max_pts, int total_order,
                                                             FrameSharesArgs sh)
{
  __shared__ unsigned s_hist[BIN_MAX_TILES];
  __shared__ unsigned s_part[256];
  if ((int)blockIdx.x == P.noct) {            // the extra column of workgroups of a balanced batch: one of them builds the
    if (blockIdx.y == 0) {                    // block tables, the others have nothing to do
      __shared__ unsigned long long s_total;
      frame_shares_body(counters, P.nframes, P.noct, max_pts, sh.t_a, sh.map_a, sh.t_b, sh.map_b, s_part, s_total);
    }
    return;
  }
  const int o = blockIdx.x + 1, frame = blockIdx.y, tid = threadIdx.x;
  const unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const int n = (int)min(cnt[CNT_DET + o], (unsigned)max_pts);
  if (n == 0) return;
  const Detection *src = in + ((size_t)frame * MISIFT_MAX_OCTAVES + (o - 1)) * max_pts;
  Detection *dst = out + ((size_t)frame * MISIFT_MAX_OCTAVES + (o - 1)) * max_pts;
  int shift = 5;
  int tx = (P.o[o].w >> shift) + 1, ty = (P.o[o].h >> shift) + 1;
  while (tx * ty > BIN_MAX_TILES) { shift++; tx = (P.o[o].w >> shift) + 1; ty = (P.o[o].h >> shift) + 1; }
  const int ntiles = tx * ty;
  for (int t = tid; t < ntiles; t += 256) s_hist[t] = 0;
  __syncthreads();
  auto key = [&](const Detection &d) -> int {
    const int kx = clampi((int)d.xpos >> shift, 0, tx - 1), ky = clampi((int)d.ypos >> shift, 0, ty - 1);
    return ky * tx + kx;
  };
  for (int i = tid; i < n; i += 256) atomicAdd(&s_hist[key(src[i])], 1u);
  __syncthreads();
  // exclusive prefix sum over the tiles: 16 consecutive tiles per thread, then a scan of the 256 partial sums
  const int per = (ntiles + 255) / 256;
  unsigned local = 0;
  for (int t = tid * per; t < min((tid + 1) * per, ntiles); t++) local += s_hist[t];
  s_part[tid] = local;
  __syncthreads();
  for (int ofs = 1; ofs < 256; ofs <<= 1) {
    const unsigned v = tid >= ofs ? s_part[tid - ofs] : 0u;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  unsigned run = s_part[tid] - local;                   // exclusive prefix of this thread's first tile
  for (int t = tid * per; t < min((tid + 1) * per, ntiles); t++) {
    const unsigned c = s_hist[t];
    s_hist[t] = run;
    run += c;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const Detection d = src[i];
    dst[atomicAdd(&s_hist[key(d)], 1u)] = d;
  }
  if (!total_order) return;
  // deterministic mode: the scatter above left every tile's records contiguous (s_hist[t] is now the END of tile t) but
  // in atomic order; one thread per tile insertion-sorts its handful of records into the total order of det_less
  __syncthreads();
  __threadfence_block();
  for (int t = tid; t < ntiles; t += 256) {
    const int end = (int)s_hist[t], beg = t ? (int)s_hist[t - 1] : 0;
    for (int a = beg + 1; a < end; a++) {
      const Detection v = dst[a];
      int b = a - 1;
      while (b >= beg && det_less(v, dst[b])) { dst[b + 1] = dst[b]; b--; }
      dst[b + 1] = v;
    }
  }
}

// deterministic mode: orient_all handed out the second-orientation slots with atomicAdd, i.e. in completion order;
// renumber them in keypoint order (one workgroup per (octave, frame), blocked prefix sum over the has-duplicate flags)
__global__ __launch_bounds__(256) void renumber_dups_kernel(PyramidInfo P, const unsigned *__restrict__ counters,
                                                            Detection *__restrict__ det, int max_pts)
{
  __shared__ int s_part[256];
  const int o = blockIdx.x + 1, frame = blockIdx.y, tid = threadIdx.x;
  const unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const int n = (int)min(cnt[CNT_DET + o], (unsigned)max_pts);
  Detection *d = det + ((size_t)frame * MISIFT_MAX_OCTAVES + (o - 1)) * max_pts;
  const int per = (n + 255) / 256, beg = min(tid * per, n), end = min(beg + per, n);
  int local = 0;
  for (int i = beg; i < end; i++) local += d[i].dupslot >= 0 ? 1 : 0;
  s_part[tid] = local;
  __syncthreads();
  for (int ofs = 1; ofs < 256; ofs <<= 1) {
    const int v = tid >= ofs ? s_part[tid - ofs] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - local;
  for (int i = beg; i < end; i++)
    if (d[i].dupslot >= 0) d[i].dupslot = run++;
}

#ifndef ORIENT_FETCH_CONST
#define ORIENT_FETCH_CONST 0
#endif
template <bool Q8>
__global__ __launch_bounds__(256) void orient_all_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                         unsigned *__restrict__ counters,
                                                         Detection *__restrict__ det, int max_pts, int frac8)
{
  __shared__ float s_hist[WAVES_PER_BLOCK][64];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  __shared__ float2 s_smp[WAVES_PER_BLOCK][128];
  __shared__ float s_tgrid[WAVES_PER_BLOCK][176];
  __shared__ float s_tile[WAVES_PER_BLOCK][OW * OW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * max_pts;
  const bool q8 = Q8;          // compile-time: no per-fetch branch on the weight quantisation
  if (lane >= 57) s_smp[wave][64 + lane] = make_float2(-1.0f, 0.0f);
  const FrameCounts fc = load_frame_counts(cnt, P.noct, max_pts, false);
  const int stride = gridDim.x * WAVES_PER_BLOCK;
  // window texel t = lane + 64*k, k = 0..3, is (row 4k + lane/16, column lane%16): one load instruction covers 4 rows
  const int tr = lane >> 4, tc = lane & 15;
  auto fetch = [&](const float4 &k, const OctaveInfo &L, float (&T)[4]) {
    const float *img = scratch + (long long)frame * P.frame_stride + L.img_off;
    const bool sane = k.x > -64.0f && k.y > -64.0f && k.x < (float)(L.w + 64) && k.y < (float)(L.h + 64);
#if ORIENT_FETCH_CONST      // timing experiment only (wrong results): every window is the image's corner, i.e. L2-resident —
    const int x0 = (sane ? 0 : 1) - 7, y0 = -7;      // the upper bound of what sharing descr_all's window could save here
#else
    const int x0 = (int)floorf(sane ? k.x : 0.0f) - 7, y0 = (int)floorf(sane ? k.y : 0.0f) - 7;
#endif
    const unsigned col = (unsigned)clampi(x0 + tc, 0, L.w - 1);
#pragma unroll
    for (int q = 0; q < 4; q++)
      T[q] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(img) +
                                              (__umul24((unsigned)clampi(y0 + tr + 4 * q, 0, L.h - 1), (unsigned)L.p) + col) * 4u);
  };
  int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + wave);
  int o, i, o1 = 0, i1 = 0, o2 = 0, i2 = 0;
  bool more = flat_to_octave(fc, P.noct, idx, o, i);
  if (!more) return;
  // three-deep software pipeline: record of keypoint n+2 | window of keypoint n+1 (registers) | keypoint n (LDS)
  float4 cur = *reinterpret_cast<const float4 *>(&fdet[(size_t)(o - 1) * max_pts + i]), nxt = cur, nxt2 = cur;
  bool more1 = flat_to_octave(fc, P.noct, idx + stride, o1, i1);
  if (more1) nxt = *reinterpret_cast<const float4 *>(&fdet[(size_t)(o1 - 1) * max_pts + i1]);
  float T[4];
  fetch(cur, P.o[o], T);
  while (more) {
    const OctaveInfo &L = P.o[o];
    const float *img = scratch + (long long)frame * P.frame_stride + L.img_off;
    Detection *d = &fdet[(size_t)(o - 1) * max_pts + i];
    const int co = o;
    const bool sane = cur.x > -64.0f && cur.y > -64.0f && cur.x < (float)(L.w + 64) && cur.y < (float)(L.h + 64);
    const int x0 = (int)floorf(sane ? cur.x : 0.0f) - 7, y0 = (int)floorf(sane ? cur.y : 0.0f) - 7;
#pragma unroll
    for (int q = 0; q < 4; q++) s_tile[wave][lane + 64 * q] = T[q];
    wave_sync();
    if (more1) fetch(nxt, P.o[o1], T);
    const bool more2 = more1 && flat_to_octave(fc, P.noct, idx + 2 * stride, o2, i2);
    if (more2) nxt2 = *reinterpret_cast<const float4 *>(&fdet[(size_t)(o2 - 1) * max_pts + i2]);
    const OrientResult r = sane ? orient_core_tile(img, L.w, L.h, L.p, q8, cur.x, cur.y, cur.z, s_tile[wave], x0, y0,
                                                   s_hist[wave], s_gauss[wave], s_smp[wave], s_tgrid[wave], lane)
                                : orient_core(img, L.w, L.h, L.p, q8, cur.x, cur.y, cur.z, s_hist[wave], s_gauss[wave],
                                              s_smp[wave], s_tgrid[wave], lane);
    if (lane == 0) {
      d->ori1 = r.ori1;
      d->ori2 = r.ori2;
      d->dupslot = r.has2 ? (int)atomicAdd(&cnt[CNT_DUP + co], 1u) : -1;
    }
    idx += stride;
    more = more1; o = o1; i = i1; cur = nxt;
    more1 = more2; o1 = o2; i1 = i2; nxt = nxt2;
  }
}

// Phase 1 of the descriptor from the staged tile: this lane's 4 of the 256 rotated samples -> the two votes of each
// (iangf*grad into angle bin angi, angf*grad into bin angi+1) and angi itself.  ONE sample per trip of a rolled loop:
// four bilinear fetches (8 ds_read2_b32) in flight are what fits beside the prefetched window at 4 waves/SIMD.
template <bool UNROLL>
__device__ __forceinline__ void descr_samples_tile(const float *tile, int x0, int y0, bool q8, float px, float py,
                                                   float sina, float cosa, float ssina, float scosa,
                                                   const float *gauss, int lane, float (&vx)[4], float (&vy)[4],
                                                   int (&ang)[4])
{
  const int tx = lane & 15;
  const float fx = tx - 7.5f, gx = gauss[tx];
  const float *porg = tile_origin<PW>(tile, x0, y0);
#pragma unroll
  for (int j = 0; j < 4; j++) { vx[j] = 0.0f; vy[j] = 0.0f; ang[j] = 0; }     // defined before the selects below read them
  if (UNROLL) {                                    // registers to spare (3 wavefronts per SIMD): no selects, no loop
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int y = (lane >> 4) + 4 * j;
      const float fy = y - 7.5f;
      const float xpos = px + fx * scosa - fy * ssina + 0.5f;
      const float ypos = py + fx * ssina + fy * scosa + 0.5f;
      const float dx = tex2d_tile<PW>(porg, xpos + cosa, ypos + sina, q8) -
                       tex2d_tile<PW>(porg, xpos - cosa, ypos - sina, q8);
      const float dy = tex2d_tile<PW>(porg, xpos - sina, ypos + cosa, q8) -
                       tex2d_tile<PW>(porg, xpos + sina, ypos - cosa, q8);
      const float grad = gauss[y] * gx * sqrtf(dx * dx + dy * dy);
      float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
      const int angi = (int)angf;
      angf -= angi;
      vx[j] = (1.0f - angf) * grad; vy[j] = angf * grad; ang[j] = angi;
    }
    return;
  }
#pragma unroll 1
  for (int j = 0; j < 4; j++) {
    const int y = (lane >> 4) + 4 * j;
    const float fy = y - 7.5f;
    const float xpos = px + fx * scosa - fy * ssina + 0.5f;
    const float ypos = py + fx * ssina + fy * scosa + 0.5f;
    const float dx = tex2d_tile<PW>(porg, xpos + cosa, ypos + sina, q8) -
                     tex2d_tile<PW>(porg, xpos - cosa, ypos - sina, q8);
    const float dy = tex2d_tile<PW>(porg, xpos - sina, ypos + cosa, q8) -
                     tex2d_tile<PW>(porg, xpos + sina, ypos - cosa, q8);
    const float grad = gauss[y] * gx * sqrtf(dx * dx + dy * dy);
    float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
    const int angi = (int)angf;
    angf -= angi;
    const float tvx = (1.0f - angf) * grad, tvy = angf * grad;
    // the rolled loop cannot index registers dynamically: the trip's results go to their named slot by selects
    vx[0] = j == 0 ? tvx : vx[0]; vy[0] = j == 0 ? tvy : vy[0]; ang[0] = j == 0 ? angi : ang[0];
    vx[1] = j == 1 ? tvx : vx[1]; vy[1] = j == 1 ? tvy : vy[1]; ang[1] = j == 1 ? angi : ang[1];
    vx[2] = j == 2 ? tvx : vx[2]; vy[2] = j == 2 ? tvy : vy[2]; ang[2] = j == 2 ? angi : ang[2];
    vx[3] = j == 3 ? tvx : vx[3]; vy[3] = j == 3 ? tvy : vy[3]; ang[3] = j == 3 ? angi : ang[3];
  }
}

// 8x8 footprint of a cell, four rows at a time (8 b128 loads in flight: the fully unrolled form holds 64 registers and
// makes the compiler spill the prefetched window).  The 64 weights wy(row) * spatial_w(column) come from a per-wavefront
// LDS table (footprint_weights_init; wave-uniform ds_read_b128, no VALU): the rolled loop used to select wy at run time
// and multiply the 32 products out again in every trip — 16 v_mul + 4 v_cndmask per trip, 80 of a descriptor's ~970
// instructions (r03).  Same products (one IEEE multiplication each), same FMA order.
__device__ __forceinline__ void footprint_weights_init(float *wt, int lane)
{
  const int my = lane >> 3, mx = lane & 7;
  const float wy = my < 4 ? (my + 0.5f) * 0.25f : (3.5f - (my - 4)) * 0.25f;       // = the old wy of row 4h + r
  const float wx = mx < 4 ? (mx + 0.5f) * 0.25f : (7.5f - mx) * 0.25f;             // = spatial_w(mx)
  wt[lane] = wy * wx;
}
__device__ __forceinline__ float footprint_sum2(const float *base, const float *wt, float acc)
{
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int my = 4 * h + r;
      const float4 lo = *reinterpret_cast<const float4 *>(base + my * SMP_W);
      const float4 hi = *reinterpret_cast<const float4 *>(base + my * SMP_W + 4);
      const float4 wl = *reinterpret_cast<const float4 *>(wt + 8 * my);
      const float4 wh = *reinterpret_cast<const float4 *>(wt + 8 * my + 4);
      acc = __builtin_fmaf(wl.x, lo.x, acc);
      acc = __builtin_fmaf(wl.y, lo.y, acc);
      acc = __builtin_fmaf(wl.z, lo.z, acc);
      acc = __builtin_fmaf(wl.w, lo.w, acc);
      acc = __builtin_fmaf(wh.x, hi.x, acc);
      acc = __builtin_fmaf(wh.y, hi.y, acc);
      acc = __builtin_fmaf(wh.z, hi.z, acc);
      acc = __builtin_fmaf(wh.w, hi.w, acc);
    }
  }
  return acc;
}

// r06, the default (DESCR_FOOT_LITERAL=1): the same sum with the 64 weights as instruction literals — they are compile-time
// constants: no weight reads from LDS at all (16 wave-uniform ds_read_b128 per pass; an LDS instruction holds the issuing SIMD for 5-9 cycles, r06
// counters).  Same products, same FMA order: bit-identical records.  descr_all 0.321 -> 0.302 ms per 64 x 1080p step,
// +1.5...2.7 % frames/s (profiles/r06_footlit_ab.txt).
__device__ __forceinline__ float footprint_sum3(const float *base, float acc)
{
  // the literal is part of the instruction (an integer literal in an f32 operand is its bit pattern): written through the
  // compiler's own fma the 64 constants are hoisted into 64 VGPRs across the orientation loop and the kernel spills.
  // One asm statement per half row (the compiler pads every statement with an s_nop); row r + 1 is loaded before row r's
  // FMAs, pinned by the scheduling barrier — left alone the compiler re-uses the row's eight registers: no load in flight.
#define FOOT_W(M) "n"(__builtin_bit_cast(unsigned, spatial_w(my) * spatial_w(M)))
  float4 lo = *reinterpret_cast<const float4 *>(base), hi = *reinterpret_cast<const float4 *>(base + 4);
#pragma unroll
  for (int my = 0; my < 8; my++) {
    float4 nlo = lo, nhi = hi;
    if (my < 7) {
      nlo = *reinterpret_cast<const float4 *>(base + (my + 1) * SMP_W);
      nhi = *reinterpret_cast<const float4 *>(base + (my + 1) * SMP_W + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm("v_fmac_f32 %0, %5, %1\n\tv_fmac_f32 %0, %6, %2\n\tv_fmac_f32 %0, %7, %3\n\tv_fmac_f32 %0, %8, %4"
        : "+v"(acc) : "v"(lo.x), "v"(lo.y), "v"(lo.z), "v"(lo.w), FOOT_W(0), FOOT_W(1), FOOT_W(2), FOOT_W(3));
    asm("v_fmac_f32 %0, %5, %1\n\tv_fmac_f32 %0, %6, %2\n\tv_fmac_f32 %0, %7, %3\n\tv_fmac_f32 %0, %8, %4"
        : "+v"(acc) : "v"(hi.x), "v"(hi.y), "v"(hi.z), "v"(hi.w), FOOT_W(4), FOOT_W(5), FOOT_W(6), FOOT_W(7));
    __builtin_amdgcn_sched_barrier(0);
    lo = nlo; hi = nhi;
  }
#undef FOOT_W
  return acc;
}
#ifndef DESCR_FOOT_LITERAL
#define DESCR_FOOT_LITERAL 1
#endif
#if DESCR_FOOT_LITERAL
#define FOOTPRINT_SUM(base, wt, acc) footprint_sum3((base), (acc))
#else
#define FOOTPRINT_SUM(base, wt, acc) footprint_sum2((base), (wt), (acc))
#endif

// Votes -> normalised descriptor bins (8*cell + (lane&3)) and (+4).  tbl: FOUR per-bin planes [bin][20][20] over the
// 16x16 sample grid (2-sample zero border), all zero on entry and on exit; bins 0..3 first, then the same planes are
// re-used for bins 4..7 (see the comment above SMP_W).  Sample j of this lane sits at plane position pos_j.
__device__ __forceinline__ void descr_accumulate(float *tbl, const float *wt, int lane, const float (&vx)[4], const float (&vy)[4],
                                                 const int (&ang)[4], float &out0, float &out1)
{
  const int cell = lane >> 2, cx = cell & 3, cy = cell >> 2;
  const float *mine = tbl + (lane & 3) * SMP_PLANE + (4 * cy) * SMP_W + 4 * cx;
  int pos[4], bx[4], by[4];
  bool has8 = false;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int id = lane + 64 * j;
    pos[j] = ((id >> 4) + 2) * SMP_W + (id & 15) + 2;
    bx[j] = ang[j];                                   // bin of the iangf vote (8: the special case below)
    by[j] = ang[j] >= 7 ? 0 : ang[j] + 1;             // bin of the angf vote: wraps to 0 (7 and 8 alike)
    has8 |= ang[j] >= 8;
  }
  // ---- bins 0..3
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (bx[j] < 4) tbl[bx[j] * SMP_PLANE + pos[j]] = vx[j];
    if (by[j] < 4) tbl[by[j] * SMP_PLANE + pos[j]] = vy[j];
  }
  wave_sync();
  float acc0 = FOOTPRINT_SUM(mine, wt, 0.0f);
  wave_sync();
  // ---- bins 4..7 re-use planes 0..3 (a lane's pass-A and pass-B slots never coincide: different plane or position)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (bx[j] < 4) tbl[bx[j] * SMP_PLANE + pos[j]] = 0.0f;
    if (by[j] < 4) tbl[by[j] * SMP_PLANE + pos[j]] = 0.0f;
    if (bx[j] >= 4 && bx[j] < 8) tbl[(bx[j] - 4) * SMP_PLANE + pos[j]] = vx[j];
    if (by[j] >= 4) tbl[(by[j] - 4) * SMP_PLANE + pos[j]] = vy[j];
  }
  wave_sync();
  float acc1 = FOOTPRINT_SUM(mine, wt, 0.0f);
  wave_sync();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (bx[j] >= 4 && bx[j] < 8) tbl[(bx[j] - 4) * SMP_PLANE + pos[j]] = 0.0f;
    if (by[j] >= 4) tbl[(by[j] - 4) * SMP_PLANE + pos[j]] = 0.0f;
  }
  if (__any(has8)) {
    // rare (dy == +0 and dx < 0): angi == 8 makes the iangf vote land in bin 0 of the NEXT cell
    // of the flattened 4x4 grid, with the spatial weights of the cell it was computed for; cell 16 does not exist
    // (dropped).  Third pass: those votes go to plane 0 and every cell >= 1 adds the PREVIOUS cell's footprint.
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (bx[j] >= 8) tbl[pos[j]] = vx[j];
    wave_sync();
    if ((lane & 3) == 0 && cell >= 1) {
      const int pc = cell - 1, pcx = pc & 3, pcy = pc >> 2;
      acc0 = FOOTPRINT_SUM(tbl + (4 * pcy) * SMP_W + 4 * pcx, wt, acc0);
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (bx[j] >= 8) tbl[pos[j]] = 0.0f;
  }
  // normalise, clamp at 0.2, normalise again (reference cudaSiftD.cu:390-409)
  const float tsum1 = wave_sum(acc0 * acc0 + acc1 * acc1);
  const float rs1 = 1.0f / sqrtf(tsum1);
  const float c0 = fminf(acc0 * rs1, 0.2f), c1 = fminf(acc1 * rs1, 0.2f);
  const float tsum2 = wave_sum(c0 * c0 + c1 * c1);
  const float rs2 = 1.0f / sqrtf(tsum2);
  out0 = c0 * rs2;
  out1 = c1 * rs2;
}

// r06 (DESCR_SCATTER_SELECT): the same accumulation with every vote owning ONE table slot for both passes — plane
// (bin & 3), its sample's position — written unconditionally three times: the vote or 0 (bins 0..3), 0 or the vote (bins
// 4..7), 0.  The conditional form above costs, per store, a compare into an SGPR pair, s_and_saveexec, a v_mad_u64_u32 for
// bin * 400 + position (a quarter-rate instruction: the compiler has no full-rate 32-bit multiply-add), the store and the
// exec restore — 32 of them per descriptor; here: 8 v_mad_u32_u24 once, 24 stores, 16 selects.  The slots of different
// votes never coincide (different sample = different position; a sample's two votes = adjacent bins = different planes)
// except for the angi == 8 vote (plane 0 like its partner's bin 0), which goes to a per-lane dump slot instead.
// Same table contents in every pass, same footprint sums: bit-identical records.
__device__ __forceinline__ void descr_accumulate_sel(float *tbl, const int dump, int lane, const float (&vx)[4],
                                                     const float (&vy)[4], const int (&ang)[4], float &out0, float &out1)
{
  const int cell = lane >> 2, cx = cell & 3, cy = cell >> 2;
  const float *mine = tbl + (lane & 3) * SMP_PLANE + (4 * cy) * SMP_W + 4 * cx;
  const unsigned pos0 = (unsigned)(((lane >> 4) + 2) * SMP_W + (lane & 15) + 2);       // sample j: + 4 * SMP_W * j
  unsigned sx[4], sy[4];
  bool has8 = false;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned bx = (unsigned)ang[j];
    const unsigned by = ang[j] >= 7 ? 0u : bx + 1u;
    const unsigned pos = pos0 + 4 * SMP_W * j;
    sx[j] = ang[j] >= 8 ? (unsigned)dump : __umul24(bx & 3u, SMP_PLANE) + pos;
    sy[j] = __umul24(by & 3u, SMP_PLANE) + pos;
    has8 |= ang[j] >= 8;
  }
  // ---- bins 0..3
#pragma unroll
  for (int j = 0; j < 4; j++) {
    tbl[sx[j]] = ang[j] < 4 ? vx[j] : 0.0f;
    tbl[sy[j]] = (ang[j] < 3 || ang[j] >= 7) ? vy[j] : 0.0f;          // by < 4
  }
  wave_sync();
  float acc0 = FOOTPRINT_SUM(mine, nullptr, 0.0f);
  wave_sync();
  // ---- bins 4..7 in the same planes
#pragma unroll
  for (int j = 0; j < 4; j++) {
    tbl[sx[j]] = ang[j] < 4 ? 0.0f : vx[j];                           // (angi == 8: the dump slot)
    tbl[sy[j]] = (ang[j] < 3 || ang[j] >= 7) ? 0.0f : vy[j];
  }
  wave_sync();
  float acc1 = FOOTPRINT_SUM(mine, nullptr, 0.0f);
  wave_sync();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    tbl[sx[j]] = 0.0f;
    tbl[sy[j]] = 0.0f;
  }
  if (__any(has8)) {                                                  // rare: see descr_accumulate
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (ang[j] >= 8) tbl[pos0 + 4 * SMP_W * j] = vx[j];
    wave_sync();
    if ((lane & 3) == 0 && cell >= 1) {
      const int pc = cell - 1, pcx = pc & 3, pcy = pc >> 2;
      acc0 = FOOTPRINT_SUM(tbl + (4 * pcy) * SMP_W + 4 * pcx, nullptr, acc0);
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (ang[j] >= 8) tbl[pos0 + 4 * SMP_W * j] = 0.0f;
  }
  const float tsum1 = wave_sum(acc0 * acc0 + acc1 * acc1);
  const float rs1 = 1.0f / sqrtf(tsum1);
  const float c0 = fminf(acc0 * rs1, 0.2f), c1 = fminf(acc1 * rs1, 0.2f);
  const float tsum2 = wave_sum(c0 * c0 + c1 * c1);
  const float rs2 = 1.0f / sqrtf(tsum2);
  out0 = c0 * rs2;
  out1 = c1 * rs2;
}
#ifndef DESCR_SCATTER_SELECT
#define DESCR_SCATTER_SELECT 1      // descr_all 0.300 -> 0.285 ms per 64 x 1080p step (profiles/r06_scatter_ab.txt)
#endif

// Write one finished record (both targets: the per-frame array and / or the packed array).
__device__ __forceinline__ void descr_write(SiftPointD *sift, SiftPointD *pack_dst, int pack_off, unsigned pack_cnt,
                                            unsigned dst, int lane, float o0, float o1v, const Detection &d,
                                            float orientation, float subsampling, float out_scale)
{
  // out_scale is 1, or 0.5 with scaleUp: the reference's RescalePositions (cudaSiftH.cu:130) folded into the write.  It
  // covers numPts records, so the callers pass 1 for the finest octave's second orientations, which lie PAST numPts
  // (cudaSiftH.cu:115) and keep their unscaled coordinates in the reference's device array.
  const int cell = lane >> 2;
#pragma unroll
  for (int tgt = 0; tgt < 2; tgt++) {
    SiftPointD *p = tgt == 0 ? (sift ? &sift[dst] : nullptr)
                             : (pack_dst && dst < pack_cnt ? pack_dst + pack_off + dst : nullptr);
    if (!p) continue;
    p->data[8 * cell + (lane & 3)] = o0;
    p->data[8 * cell + (lane & 3) + 4] = o1v;
    if (lane == 0) {
      p->xpos = d.xpos * subsampling * out_scale;       // out_scale is 1 or 0.5: exact, = a later RescalePositions
      p->ypos = d.ypos * subsampling * out_scale;
      p->scale = d.scale * subsampling * out_scale;
      p->sharpness = d.sharpness;
      p->edgeness = d.edgeness;
      p->orientation = orientation;
      p->subsampling = subsampling;
      if (tgt == 1) {                       // a packed record is complete: the match fields start out cleared
        p->score = 0.0f; p->ambiguity = 0.0f; p->match = 0; p->match_xpos = 0.0f; p->match_ypos = 0.0f;
        p->match_error = 0.0f; p->empty[0] = 0.0f; p->empty[1] = 0.0f; p->empty[2] = 0.0f;
      }
    }
  }
}

// descr_all_kernel — the default descriptor kernel.  One wavefront per keypoint; ONE 6.4 KB LDS buffer per wavefront
// that is first the keypoint's 40x40 image window and then, once all samples are taken, the 4-plane vote table
// (1600 floats either way) — so LDS never limits the occupancy.  The kernel is bound by VALU instruction issue
// (tools/valu_rates: 2.85 cycles per plain fp32 op, 4.4-4.7 for everything else, per SIMD from 2-3 wavefronts on); the
// fourth wavefront per SIMD only covers the LDS round trips between the phases.
//   per keypoint:  window (registers, prefetched) -> LDS | samples of the 1st and, if there is one, the 2nd orientation
//                  (votes stay in registers) | clear | votes -> table -> footprints -> normalise -> record, once per
//                  orientation.
// Keypoints whose window would not fit (scale > 2.1: only through the refinement's unclamped fallback step) are
// appended to a per-frame list and done by descr_big_kernel from global memory afterwards.
#ifndef DESCR_OCC
#define DESCR_OCC 4
#endif
#ifndef DESCR_UNROLL_SAMPLES
#define DESCR_UNROLL_SAMPLES 1
#endif
// one LDS record per wavefront, so that every access is one lane-offset register plus an immediate offset
struct alignas(16) DescrWaveLds {
  float buf[PATCH_FLOATS];      // window, then vote table
  float park[12 * 64];          // votes of a first orientation while the second is sampled
  float gauss[16];
  float wtab[64];               // footprint weights wy(row) * wx(column) (DESCR_FOOT_LITERAL=0 only)
};
// FUSE (orient_descr_fused_kernel): the orientations were computed by THIS launch — the duplicate counts of the coarser
// octaves come from `fdup` (LDS, filled behind the launch's in-kernel wait), a keypoint's own orientation fields are read
// back with vector loads (the scalar cache may hold the record's line from before they were written), and the reference's
// counters are published by the launch's last workgroup instead of its first.
template <bool Q8, bool BAL, bool FUSE = false>
__device__ __forceinline__ void descr_all_body(const float *__restrict__ scratch, const PyramidInfo &P,
                                               unsigned *__restrict__ counters, const Detection *__restrict__ det,
                                               SiftPointD *__restrict__ pts, int max_pts,
                                               const int *__restrict__ pack_offsets, SiftPointD *__restrict__ pack_dst,
                                               unsigned *__restrict__ big_list, unsigned big_stride, DescrWaveLds *s_w,
                                               const int4 *__restrict__ block_map, const unsigned *fdup = nullptr,
                                               int fuse_items = 0)
{
  static_assert(PATCH_FLOATS == 4 * SMP_PLANE, "the window and the four vote planes share one buffer");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const FrameShare fs = frame_share<BAL>(block_map);
  if (BAL && fs.frame < 0) return;
  const int frame = fs.frame;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * max_pts;
  SiftPointD *sift = pts ? pts + (size_t)frame * max_pts : nullptr;
  // packed output: this frame's first numPts records go to pack_dst[pack_offsets[frame] ...] (what a gather ships)
  const int pack_off = pack_dst ? __builtin_amdgcn_readfirstlane(pack_offsets[frame]) : 0;
  const unsigned pack_cnt = pack_dst ? (unsigned)(__builtin_amdgcn_readfirstlane(pack_offsets[frame + 1]) - pack_off) : 0u;
  const bool q8 = Q8;          // compile-time: no per-fetch branch on the weight quantisation
  float *buf = s_w[wave].buf;
  const float *gauss = s_w[wave].gauss;
  if (lane < 16) s_w[wave].gauss[lane] = det_exp(-(lane - 7.5f) * (lane - 7.5f) / 128.0f);
  if (!DESCR_FOOT_LITERAL) footprint_weights_init(s_w[wave].wtab, lane);
  // segment layout of the reference: detections of octave o start where octave o-1 (incl. its duplicates) ended
  if (!FUSE && fs.sub == 0 && threadIdx.x == 0) {       // publish the reference's counters (cudaSiftD.cu:14)
    unsigned b = 0;                                      // (write-through: a workgroup on another XCD may export them)
    for (int k = 1; k <= P.noct; k++) {
      __hip_atomic_store(&cnt[2 * k - 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b += cnt[CNT_DET + k];
      __hip_atomic_store(&cnt[2 * k], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b += cnt[CNT_DUP + k];
      __hip_atomic_store(&cnt[2 * k + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // FUSE: a wavefront takes `fuse_items` CONSECUTIVE keypoints instead (see orient_descr_fused_kernel: what a workgroup
  // waits for then belongs to workgroups with lower indices)
  const int stride = FUSE ? 1 : fs.nsub * WAVES_PER_BLOCK;
  // Keypoints of a frame are numbered octave after octave (coarsest first); a wavefront takes every stride-th one.
  // (octave, index) advance incrementally — the per-octave counts are re-read only when an octave is exhausted.
  auto ndet = [&](int k) -> int {
    return (int)min(__builtin_amdgcn_readfirstlane(cnt[CNT_DET + k]), (unsigned)max_pts);
  };
  // `lim` = number of keypoints of octave o, carried along with (o, i): the counters are read from memory only when
  // an octave is exhausted.  (Reading them on every call put a global load + s_waitcnt vmcnt(0) right behind the
  // window prefetch of the next keypoint — vmcnt retires in order, so every keypoint waited for its successor's
  // window: r02 SQ counters, a third of the wavefronts' lifetime in s_waitcnt.)
  auto advance = [&](int &o, int &i, int &lim, int by) -> bool {      // false: past the last keypoint
    i += by;
    while (i >= lim) {
      i -= lim;
      o++;
      if (o > P.noct) return false;
      lim = ndet(o);
    }
    return true;
  };
  // three-deep software pipeline over this wavefront's keypoints:
  //   Detection record of keypoint n+2 | window of keypoint n+1 (global -> registers) | keypoint n (LDS)
  int o = 1, i = 0, lim = ndet(1), o1, i1, lim1, o2, i2, lim2;
  bool more = advance(o, i, lim, __builtin_amdgcn_readfirstlane(fs.sub * WAVES_PER_BLOCK + wave) * (FUSE ? fuse_items : 1));
  if (!more) return;
  int taken = 0;                                          // FUSE: keypoints of this wavefront's range behind the current one
  o1 = o; i1 = i; lim1 = lim;
  bool more1 = (!FUSE || 1 < fuse_items) && advance(o1, i1, lim1, stride);
  o2 = o1; i2 = i1; lim2 = lim1;
  // of the keypoints ahead only the first 16 bytes of the record (xpos, ypos, scale: what the window needs) are held;
  // the full record is (re)loaded when the keypoint becomes current — scalar loads, the line is in the scalar cache
  auto head = [&](int oo, int ii) -> float4 {
    const Detection &r = fdet[(size_t)(oo - 1) * max_pts + ii];      // uniform address: scalar loads
    return make_float4(r.xpos, r.ypos, r.scale, 0.0f);
  };
  float4 h0 = head(o, i), h1 = h0, h2 = h0;
  if (more1) h1 = head(o1, i1);
  PatchGeom g = patch_geom(h0.x, h0.y, h0.z, P.o[o].w, P.o[o].h, P.patch_reach), g1 = g;
  float R[PATCH_LOADS];
  if (g.fits) patch_fetch(scratch + (long long)frame * P.frame_stride + P.o[o].img_off, P.o[o].w, P.o[o].h, P.o[o].p, g, lane, R);
  int cur_o = 0;
  unsigned bdet = 0, bdup = 0;                            // segment bases of octave cur_o in the reference layout
  while (more) {
    const float subsampling = P.o[o].subsampling;
    Detection d = fdet[(size_t)(o - 1) * max_pts + i];
    if (FUSE) {                                           // written by this wavefront a moment ago: not through the scalar cache
      const unsigned *rec = reinterpret_cast<const unsigned *>(&fdet[(size_t)(o - 1) * max_pts + i]);
      static_assert(offsetof(Detection, ori1) == 20 && offsetof(Detection, ori2) == 24 && offsetof(Detection, dupslot) == 28, "");
      unsigned v = 0u;                                    // ONE load: lanes 0..2 fetch ori1, ori2, dupslot
      if (lane < 3) v = __hip_atomic_load(rec + 5 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d.ori1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)v, 0));
      d.ori2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)v, 1));
      d.dupslot = __builtin_amdgcn_readlane((int)v, 2);
    }
    if (g.fits) patch_store(buf, lane, R);
    wave_sync();
    // ---- keypoint n+1: window loads into the registers just drained; keypoint n+2: the head of its record
    // (lane_i: the lane id behind an opaque barrier, so that the dozen lane-only address terms of the window fetch and
    //  of the clear below are recomputed here — a few VALU — instead of being hoisted out of the loop and SPILLED: a
    //  scratch reload is a VMEM operation, and waiting for it means waiting for the whole prefetch in front of it)
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i));
    if (more1) {
      const OctaveInfo &L1 = P.o[o1];
      g1 = patch_geom(h1.x, h1.y, h1.z, L1.w, L1.h, P.patch_reach);
      if (g1.fits) patch_fetch(scratch + (long long)frame * P.frame_stride + L1.img_off, L1.w, L1.h, L1.p, g1, lane_i, R);
    }
    const bool more2 = more1 && (!FUSE || taken + 2 < fuse_items) && advance(o2, i2, lim2, stride);
    taken++;
    if (more2) h2 = head(o2, i2);
    // ---- keypoint n
    if (o != cur_o) {                                     // octave changed: segment bases (cudaSiftD.cu:1297-1300)
      unsigned b = 0;
      for (int k = 1; k < o; k++)
        b += __builtin_amdgcn_readfirstlane(cnt[CNT_DET + k]) +
             __builtin_amdgcn_readfirstlane(FUSE ? fdup[k] : cnt[CNT_DUP + k]);
      bdet = b;
      bdup = b + __builtin_amdgcn_readfirstlane(cnt[CNT_DET + o]);
      cur_o = o;
    }
    const unsigned dstA = bdet + (unsigned)i;
    const bool dup = d.dupslot >= 0;
    const unsigned dstB = bdup + (unsigned)(dup ? d.dupslot : 0);
    const bool doA = dstA < (unsigned)max_pts, doB = dup && dstB < (unsigned)max_pts;   // capacity: dropped, still counted
    if (!g.fits) {
      if (lane == 0 && (doA || doB)) {                    // too large for the window: descr_big_kernel takes it
        const unsigned slot = atomicAdd(&cnt[CNT_BIG], 1u);
        if (slot < big_stride)
          big_list[(size_t)frame * big_stride + slot] = ((unsigned)o << 24) | (unsigned)i;
      }
    } else if (doA || doB) {
      // one set of vote registers: when a keypoint has both orientations, the first one's votes wait in LDS (s_park)
      // while the second one is sampled — two live sets cost 12 registers the kernel does not have at 4 waves/SIMD
      float vx[4], vy[4];
      int ang[4];
      const float scale = 12.0f / 16.0f * d.scale;
      const int nori = (doA ? 1 : 0) + (doB ? 1 : 0);
      float *park = s_w[wave].park + lane;
      // sampling: the orientation(s) to do, first then second (ONE copy of the sampling code: rolled loop)
#pragma unroll 1
      for (int k = 0; k < nori; k++) {
        if (k == 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            park[64 * j] = vx[j];
            park[64 * (4 + j)] = vy[j];
            park[64 * (8 + j)] = __builtin_bit_cast(float, ang[j]);
          }
        }
        const float theta = 2.0f * 3.1415f / 360.0f * ((k == 0 && doA) ? d.ori1 : d.ori2);
        float sina, cosa;
        det_sincos(theta, sina, cosa);
        descr_samples_tile<DESCR_UNROLL_SAMPLES>(buf, g.x0, g.y0, q8, d.xpos, d.ypos, sina, cosa, scale * sina, scale * cosa, gauss, lane, vx, vy, ang);
      }
      wave_sync();                                        // every lane is done with the window
      {                                                   // the same 1600 floats become the (all-zero) vote table:
        float *z = buf + 4 * lane_i;                      // 6 x b128 (1536 floats) + 1 x b32 (64), immediate offsets
#pragma unroll
        for (int k = 0; k < 6; k++) *reinterpret_cast<float4 *>(z + 256 * k) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int lane_k = lane;                                // its own opaque copy: derived from z the address is z - 12 * lane,
        asm volatile("" : "+v"(lane_k));                  // which the compiler forms with a quarter-rate v_mad_u64_u32
        buf[1536 + lane_k] = 0.0f;
      }
      wave_sync();
      // accumulation: the orientation sampled last first (its votes are in registers), then the parked one
#pragma unroll 1
      for (int k = nori - 1; k >= 0; k--) {
        if (k == 0 && nori == 2) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            vx[j] = park[64 * j];
            vy[j] = park[64 * (4 + j)];
            ang[j] = __builtin_bit_cast(int, park[64 * (8 + j)]);
          }
        }
        const bool first = k == 0 && doA;
        float o0, o1v;
#if DESCR_SCATTER_SELECT && DESCR_FOOT_LITERAL
        descr_accumulate_sel(buf, (int)((offsetof(DescrWaveLds, wtab) - offsetof(DescrWaveLds, buf)) / sizeof(float)) + lane, lane,
                             vx, vy, ang, o0, o1v);                   // wtab: unused with literal weights, the dump slots
#else
        descr_accumulate(buf, s_w[wave].wtab, lane, vx, vy, ang, o0, o1v);
#endif
        descr_write(sift, pack_dst, pack_off, pack_cnt, first ? dstA : dstB, lane, o0, o1v, d, first ? d.ori1 : d.ori2,
                    subsampling, (!first && o == P.noct && !P.fix_numpts) ? 1.0f : P.out_scale);
      }
    }
    wave_sync();                                          // the buffer is free (and all zero or about to be overwritten)
    more = more1; o = o1; i = i1; g = g1;
    more1 = more2; o1 = o2; i1 = i2; h1 = h2;
    (void)lim; (void)lim1;
  }
}

template <bool Q8>
__device__ __forceinline__ void descr_big_frame(const float *__restrict__ scratch, const PyramidInfo &P,
                                                const unsigned *__restrict__ cnt, const Detection *__restrict__ det,
                                                SiftPointD *__restrict__ pts, int max_pts,
                                                const int *__restrict__ pack_offsets, SiftPointD *__restrict__ pack_dst,
                                                const unsigned *__restrict__ big_list, unsigned big_stride, unsigned nbig,
                                                float *s_smp_w, float *s_gauss_w, int lane, int frame, unsigned first,
                                                unsigned step)
{
  const Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * max_pts;
  SiftPointD *sift = pts ? pts + (size_t)frame * max_pts : nullptr;
  const int pack_off = pack_dst ? pack_offsets[frame] : 0;
  const unsigned pack_cnt = pack_dst ? (unsigned)(pack_offsets[frame + 1] - pack_off) : 0u;
  descr_init(s_smp_w, s_gauss_w, lane);
  for (unsigned t = first; t < nbig; t += step) {
    const unsigned code = big_list[(size_t)frame * big_stride + t];
    const int o = (int)(code >> 24), i = (int)(code & 0xffffffu);
    const OctaveInfo &L = P.o[o];
    const float *img = scratch + (long long)frame * P.frame_stride + L.img_off;
    const Detection d = fdet[(size_t)(o - 1) * max_pts + i];
    unsigned b = 0;
    for (int k = 1; k < o; k++) b += cnt[CNT_DET + k] + cnt[CNT_DUP + k];
    const unsigned bdet = b, bdup = b + cnt[CNT_DET + o];
    for (int which = 0; which < 2; which++) {
      if (which == 1 && d.dupslot < 0) break;
      const unsigned dst = which == 0 ? bdet + (unsigned)i : bdup + (unsigned)d.dupslot;
      if (dst >= (unsigned)max_pts) continue;
      float o0, o1v;
      descr_core(img, L.w, L.h, L.p, Q8, d.xpos, d.ypos, d.scale, which == 0 ? d.ori1 : d.ori2, s_smp_w, s_gauss_w,
                 lane, o0, o1v);
      descr_write(sift, pack_dst, pack_off, pack_cnt, dst, lane, o0, o1v, d, which == 0 ? d.ori1 : d.ori2, L.subsampling,
                  (which == 1 && o == P.noct && !P.fix_numpts) ? 1.0f : P.out_scale);
    }
  }
}

// Hand the counter blocks of all frames to the host (misift_extract_sync): called by ONE workgroup once every other
// workgroup of the call's last kernel has finished.  The counters go into pinned host memory, the call's sequence
// number behind them — the host polls that word instead of queueing a blocking copy behind the kernel (r04 single-call
// budget: the copy was a blit kernel of its own plus a stream synchronisation).
__device__ __forceinline__ void export_counters_host(const unsigned *counters, unsigned nframes, unsigned *host_out,
                                                     unsigned host_seq)
{
  const unsigned nwords = nframes * CNT_STRIDE;
  // (agent-scope atomic loads: the counters were written by other workgroups — atomics, or the write-through stores of
  //  publish_counters — and must not be served from this XCD's L2)
  // (system scope since r06: the fused kernel's workgroups poll words of these blocks while other workgroups of the SAME
  //  launch still add to them, so a line of them may sit in this XCD's L2 from before the last update)
  for (unsigned w = threadIdx.x; w < nwords; w += blockDim.x)
    host_out[w] = __hip_atomic_load(counters + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_out + nwords, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// true in exactly one workgroup of the launch: the one that draws the last ticket.  NO device-scope fence per workgroup:
// on this multi-XCD part a release at agent scope writes the whole L2 back, and a thousand workgroups doing that cost a
// single frame 80 us (r04, profiles/r04_single_call_sweep_step3.txt).  What the last workgroup reads from the others
// must therefore have been written with agent-scope atomics (write-through).
__device__ __forceinline__ bool last_workgroup(unsigned *counters)
{
  __shared__ unsigned s_last;
  // The ordering below RESTS on this wait: a workgroup-scope release fence compiles to s_waitcnt lgkmcnt(0) only, and the
  // ticket must not be drawn before every store of this wavefront has been ACKNOWLEDGED (each wavefront waits for its own;
  // no L2 write-back) — else the host could be told "done" with records still in flight (advisor r04).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (inline asm: invisible to the pass that drops waits)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // this workgroup's stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(counters + CNT_TICKET, 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
  __syncthreads();
  return s_last != 0;
}

// The same for a launch of a thousand workgroups of ONE frame (single-call path, r05): tickets in two levels — 16 words in
// cache lines of their own, then one more for the workgroups that drew the last ticket of theirs — because same-address
// atomics are served one at a time (10-50 ns each: r04 measured a single-level ticket of descr_all at more than the
// dispatch it saved).  The words live in the spare counter blocks behind the last frame's (cleared by the prefilter; the
// scan's embedded chain uses words 0, 32 and 64 + 32 k of the same blocks, these are 48 and 80 + 32 k).
__device__ __forceinline__ bool last_workgroup_2level(unsigned *spare)
{
  __shared__ unsigned s_last2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // records and published counters acknowledged BEFORE the ticket
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nwg = gridDim.x * gridDim.y, lb = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned sub = lb & 15u, members = (nwg - sub + 15u) / 16u;
    unsigned last = 0u;
    if (atomicAdd(&spare[80 + 32 * sub], 1u) == members - 1u) {
      const unsigned nsub = nwg < 16u ? nwg : 16u;
      if (atomicAdd(&spare[48], 1u) == nsub - 1u) last = 1u;
    }
    s_last2 = last;
  }
  __syncthreads();
  return s_last2 != 0;
}

// tail_host_out != NULL (single-call path, r05): this launch is the call's LAST one — the workgroup that finishes last hands
// the counter blocks to the host, and descr_big_kernel is launched only if the host then finds a keypoint deferred to it
// (CNT_BIG != 0: a descriptor window larger than the LDS tile, i.e. next to never) — five dependent dispatches per call
// instead of six.
template <bool Q8, int OCC, bool BAL>
__global__ __launch_bounds__(256, OCC) void descr_all_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                        unsigned *__restrict__ counters,
                                                        const Detection *__restrict__ det,
                                                        SiftPointD *__restrict__ pts, int max_pts, int frac8,
                                                        const int *__restrict__ pack_offsets,
                                                        SiftPointD *__restrict__ pack_dst,
                                                        unsigned *__restrict__ big_list, unsigned big_stride,
                                                        const int4 *__restrict__ block_map,
                                                        unsigned *__restrict__ tail_host_out, unsigned tail_seq)
{
  __shared__ DescrWaveLds s_w[WAVES_PER_BLOCK];
  descr_all_body<Q8, BAL>(scratch, P, counters, det, pts, max_pts, pack_offsets, pack_dst, big_list, big_stride, s_w,
                          block_map);
  if (!BAL && tail_host_out) {                      // (wave-uniform kernel argument; batches never take this branch)
    if (last_workgroup_2level(counters + (size_t)P.nframes * CNT_STRIDE))
      export_counters_host(counters, (unsigned)P.nframes, tail_host_out, tail_seq);
  }
}

// The few keypoints descr_all_kernel deferred (window larger than 40x40 texels): bilinear fetches from global memory.
// Last kernel of a batch's extraction: on request the workgroup that finishes last also exports the counters.
template <bool Q8>
__global__ __launch_bounds__(256) void descr_big_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                        const unsigned *__restrict__ counters,
                                                        const Detection *__restrict__ det,
                                                        SiftPointD *__restrict__ pts, int max_pts,
                                                        const int *__restrict__ pack_offsets,
                                                        SiftPointD *__restrict__ pack_dst,
                                                        const unsigned *__restrict__ big_list, unsigned big_stride,
                                                        unsigned *__restrict__ host_out, unsigned host_seq)
{
  __shared__ __attribute__((aligned(16))) float s_smp[WAVES_PER_BLOCK][DESCR_TBL];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  const unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const unsigned nbig = min(cnt[CNT_BIG], big_stride);
  if (nbig != 0)
    descr_big_frame<Q8>(scratch, P, cnt, det, pts, max_pts, pack_offsets, pack_dst, big_list, big_stride, nbig,
                        s_smp[wave], s_gauss[wave], lane, frame, blockIdx.x * WAVES_PER_BLOCK + wave,
                        gridDim.x * WAVES_PER_BLOCK);
  if (!host_out) return;
  if (!last_workgroup(const_cast<unsigned *>(counters))) return;
  export_counters_host(counters, gridDim.y, host_out, host_seq);
}

// ---- the r01 forms of the two merged-octave kernels: bilinear fetches straight from global memory (no staged
// tile), 4 waves/SIMD.  Selected with MISIFT_TILE=0 (A/B measurements, docs/LOG.md section 9).
// 80 VGPRs: 6 waves/SIMD (the kernel is bound by the dependent LDS / shuffle chain of one keypoint per wavefront, so
// every extra resident wavefront helps; 5 until r06)
// (s_done: orient_descr_fused_kernel only — the workgroup's finished orientations per octave, for the octaves some other
//  wavefront will wait for, i.e. all but the finest)
template <bool Q8, bool BAL, bool FUSE = false>
__device__ __forceinline__ void orient_gather_body(const float *__restrict__ scratch, const PyramidInfo &P,
                                                   unsigned *__restrict__ counters, Detection *__restrict__ det, int max_pts,
                                                   const int4 *__restrict__ block_map, float *hist, float *gauss, float2 *smp,
                                                   float *tgrid, unsigned *s_done = nullptr, int fuse_items = 0)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const FrameShare fs = frame_share<BAL>(block_map);
  if (BAL && fs.frame < 0) return;
  const int frame = fs.frame;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * max_pts;
  const bool q8 = Q8;
  if (lane >= 57) smp[64 + lane] = make_float2(-1.0f, 0.0f);
  const FrameCounts fc = load_frame_counts(cnt, P.noct, max_pts, false);
  const int stride = FUSE ? 1 : fs.nsub * WAVES_PER_BLOCK;      // FUSE: `fuse_items` consecutive keypoints per wavefront
  int idx = __builtin_amdgcn_readfirstlane(fs.sub * WAVES_PER_BLOCK + wave) * (FUSE ? fuse_items : 1);
  const int end = idx + fuse_items;
  int o, i;
  bool more = flat_to_octave(fc, P.noct, idx, o, i);
  // the next keypoint's record is fetched while the current one is processed (its ~1 us load latency was exposed
  // at the top of every iteration)
  float4 nxt = more ? *reinterpret_cast<const float4 *>(&fdet[(size_t)(o - 1) * max_pts + i]) : make_float4(0, 0, 0, 0);
  while (more) {
    const float4 cur = nxt;
    const OctaveInfo &L = P.o[o];
    const float *img = scratch + (long long)frame * P.frame_stride + L.img_off;
    Detection *d = &fdet[(size_t)(o - 1) * max_pts + i];
    const int co = o;
    idx += stride;
    more = (!FUSE || idx < end) && flat_to_octave(fc, P.noct, idx, o, i);
    if (more) nxt = *reinterpret_cast<const float4 *>(&fdet[(size_t)(o - 1) * max_pts + i]);
    const OrientResult r = orient_core(img, L.w, L.h, L.p, q8, cur.x, cur.y, cur.z, hist, gauss, smp, tgrid, lane);
    if (lane == 0) {
      d->ori1 = r.ori1;
      d->ori2 = r.ori2;
#ifdef FUSE_NO_DUP_ATOMIC          // timing experiment only (wrong duplicate slots)
      d->dupslot = r.has2 ? 0 : -1;
#else
      d->dupslot = r.has2 ? (int)atomicAdd(&cnt[CNT_DUP + co], 1u) : -1;     // (returns: performed before anything below)
#endif
      if (FUSE && co < P.noct) atomicAdd(&s_done[co], 1u);                     // LDS
    }
  }
}

#ifndef ORIENT_OCC
#define ORIENT_OCC 6             // r06: 80 VGPRs fit six wavefronts per SIMD; with orient_blocks_per_cu = 6: orient_all 0.133 -> 0.123 ms
#endif
template <bool Q8, bool BAL>
__global__ __launch_bounds__(256, ORIENT_OCC) void orient_all_gather_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                         unsigned *__restrict__ counters,
                                                         Detection *__restrict__ det, int max_pts, int frac8,
                                                         const int4 *__restrict__ block_map)
{
  __shared__ float s_hist[WAVES_PER_BLOCK][64];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  __shared__ float2 s_smp[WAVES_PER_BLOCK][128];
  __shared__ float s_tgrid[WAVES_PER_BLOCK][176];
  const int wave = threadIdx.x >> 6;
  orient_gather_body<Q8, BAL>(scratch, P, counters, det, max_pts, block_map, s_hist[wave], s_gauss[wave], s_smp[wave],
                              s_tgrid[wave]);
}

// orient_descr_fused_kernel — ONE launch for the orientations and the descriptors of a single call (a frame or two; r05
// review: "persistent orientation + descriptor kernel").  A record's slot in the reference's layout is
// sum_{k<o}(det_k + dup_k) + i: a descriptor of octave o can be written once every orientation of the COARSER octaves is
// known.  So every wavefront first computes the orientations of ITS keypoints (orient_gather_body, nothing to wait for),
// the workgroup reports how many it finished per octave (one agent-scope atomic per octave and workgroup — same-address
// atomics are served one at a time, 10-50 ns each — and none for the finest octave, which nobody waits for), waits until
// the coarser octaves are complete, and goes on with the descriptors of the same keypoints (descr_all_body).
// FORWARD PROGRESS needs no co-residency, only dispatch in index order: a wavefront takes a range of CONSECUTIVE keypoints
// (coarsest octave first), a workgroup waits — after all its own orientations — for the octaves coarser than its LAST
// keypoint's, and every keypoint of those lies in the range of a workgroup with a lower index, whose orientation pass waits
// for nothing.  (r06's first version waited for all octaves but the finest: under HSA_CU_MASK=0:0-7 the 32 resident
// workgroups waited for orientations of workgroups that could not start.)  The wait is bounded all
// the same (`wait_ticks` of the 100 MHz clock): on expiry the workgroup skips its descriptors and raises CNT_FUSETMO; the
// host re-runs the call with the two separate launches and keeps them on that context (misift_ctx_fuse_fallbacks).
// Polling reads with SYSTEM scope: an agent-scope load may be served from this XCD's L2, which does not see the other
// XCDs' atomics (dog_scan_all_kernel's chain wait learnt that); 32 lanes fetch the block's words 32..63 in one instruction.
// What it saves is a dependent dispatch and the second kernel's ramp: r06 measurements in DESIGN.md section 4.
#ifndef FUSE_POLL_SLEEP
#define FUSE_POLL_SLEEP 32
#endif
#ifndef FUSE_STAMPS
#define FUSE_STAMPS 0            // developer build (tools/variants.sh -DFUSE_STAMPS=1): 100 MHz time stamps, tools/fuse_stamps.py
#endif
#if FUSE_STAMPS
__device__ unsigned g_fuse_stamp[16];
#define FUSE_STAMP_MAX(slot) do { if (threadIdx.x == 0) atomicMax(&g_fuse_stamp[slot], (unsigned)wall_clock64()); } while (0)
#define FUSE_STAMP_MIN(slot) do { if (threadIdx.x == 0) atomicMin(&g_fuse_stamp[slot], (unsigned)wall_clock64()); } while (0)
extern "C" int misift_debug_fuse_stamps(unsigned *out16)
{
  unsigned init[16];
  for (int i = 0; i < 16; i++) init[i] = (i == 0 || i == 8 || i == 9) ? 0xffffffffu : 0u;      // minima
  HIP_TRY(hipDeviceSynchronize());
  if (out16) HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fuse_stamp), sizeof(init)));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_stamp), init, sizeof(init)));
  return MISIFT_OK;
}
#else
#define FUSE_STAMP_MAX(slot) do { } while (0)
#define FUSE_STAMP_MIN(slot) do { } while (0)
#endif
template <bool Q8, int OCC>
__global__ __launch_bounds__(256, OCC) void orient_descr_fused_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                                 unsigned *__restrict__ counters, Detection *__restrict__ det,
                                                                 SiftPointD *__restrict__ pts, int max_pts,
                                                                 unsigned *__restrict__ big_list, unsigned big_stride,
                                                                 unsigned wait_ticks, unsigned *__restrict__ tail_host_out,
                                                                 unsigned tail_seq)
{
  __shared__ DescrWaveLds s_w[WAVES_PER_BLOCK];
  __shared__ unsigned s_done[MISIFT_MAX_OCTAVES + 1], s_fdup[MISIFT_MAX_OCTAVES + 1], s_ok;
  static_assert(64 + 16 + 2 * 128 + 176 <= 12 * 64, "the orientation's LDS arrays borrow the descriptor's parking area");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  FUSE_STAMP_MIN(0); FUSE_STAMP_MAX(1);
#if FUSE_STAMPS
  const unsigned long long t_wg = wall_clock64();
#endif
  if (threadIdx.x <= MISIFT_MAX_OCTAVES) { s_done[threadIdx.x] = 0u; s_fdup[threadIdx.x] = 0u; }
  if (threadIdx.x == 0) s_ok = 1u;
  __syncthreads();
  // keypoints per wavefront: consecutive ranges, coarsest octave first — whatever a workgroup will wait for (octaves
  // coarser than its own last keypoint's) lies in the ranges of workgroups with LOWER indices
  int per_wave, omax = 0;                      // omax: octave of the workgroup's last keypoint (0: it has none)
  {
    const FrameCounts fc = load_frame_counts(cnt, P.noct, max_pts, false);
    int total = 0;
#pragma unroll
    for (int k = 1; k <= MISIFT_MAX_OCTAVES; k++) total += fc.ndet[k];       // (0 beyond P.noct)
    const int nwaves = (int)gridDim.x * WAVES_PER_BLOCK;
    per_wave = max(1, (total + nwaves - 1) / nwaves);
    const int first = (int)blockIdx.x * WAVES_PER_BLOCK * per_wave;
    const int last = min(first + WAVES_PER_BLOCK * per_wave, total) - 1;
    int oi;
    if (last >= first) flat_to_octave(fc, P.noct, last, omax, oi);
  }
  {
    float *park = s_w[wave].park;            // hist[64] | gauss[16] | smp[128] (float2: 8-byte aligned at float 80) | tgrid[176]
    orient_gather_body<Q8, false, true>(scratch, P, counters, det, max_pts, nullptr, park, park + 64,
                                        reinterpret_cast<float2 *>(park + 80), park + 80 + 256, s_done, per_wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's orientation fields are on their way to the L2
#if FUSE_STAMPS
  if (lane == 0 && omax >= 1) atomicMax(&g_fuse_stamp[7], (unsigned)(wall_clock64() - t_wg));    // longest orientation pass of a wavefront
#endif
  // ---- report, then wait for the octaves coarser than the workgroup's last keypoint's.  (NOT with the first descriptor
  // window already in flight: the window fetches of the waiting workgroups stream through the 16 KB vector caches that the
  // orientations of the others gather from — the launch went from 40 to 59 us that way, r06_fused_orient_descr.txt.)
  const auto wait_coarser = [&]() -> bool {
    __syncthreads();
    if (omax >= 1) { FUSE_STAMP_MAX(2); FUSE_STAMP_MIN(8); if (omax < P.noct) FUSE_STAMP_MAX(6); }   // orientations of a workgroup done
    if (wave == 0) {
      if (lane >= 1 && lane < P.noct && s_done[lane] != 0u)
        __hip_atomic_fetch_add(&cnt[CNT_ORIDONE + lane], s_done[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef FUSE_NOWAIT
#define FUSE_NOWAIT 0              // timing experiment only (wrong record positions): no wait at all
#endif
      if (omax > 1 && !FUSE_NOWAIT) {              // (workgroup-uniform: the octaves 1 .. omax-1 must be complete)
        const unsigned long long t0 = wall_clock64();
        unsigned ok = wait_ticks != 0u ? 1u : 0u;           // (a bound of 0 expires before the first poll: the test of the fallback)
        // (the polled words 48..63 share no cache line with CNT_DUP, which the orientation passes of the other workgroups
        //  still add to: polling THAT line stretched their atomics — and the launch's orientation phase — from 13 to 58 us)
        const unsigned det = lane < 16 ? cnt[32 + lane] : 0u;                // CNT_DET: final since refine_all
        while (ok) {
          unsigned v = 0u;
          if (lane < 16) v = __hip_atomic_load(&cnt[48 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          bool all = true;
          for (int k = 1; k < omax; k++) {
            const unsigned need = min((unsigned)__builtin_amdgcn_readlane((int)det, CNT_DET - 32 + k), (unsigned)max_pts);
            all = all && (unsigned)__builtin_amdgcn_readlane((int)v, CNT_ORIDONE - 48 + k) >= need;
          }
          if (all) break;
          if (wall_clock64() - t0 > (unsigned long long)wait_ticks) { ok = 0u; break; }
          __builtin_amdgcn_s_sleep(FUSE_POLL_SLEEP);          // x 64 cycles
        }
        if (ok) {
          // the duplicate counts, read AFTER the completion counts were seen (another cache line: one load could see them
          // in either order)
          unsigned u = 0u;
          if (lane < 16) u = __hip_atomic_load(&cnt[32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (lane >= CNT_DUP - 32 + 1 && lane < CNT_DUP - 32 + omax) s_fdup[lane - (CNT_DUP - 32)] = u;
        } else if (lane == 0) {
          atomicAdd(&counters[CNT_FUSETMO], 1u);
          s_ok = 0u;
        }
      }
    }
    __syncthreads();
    if (omax > 1) { FUSE_STAMP_MAX(3); FUSE_STAMP_MIN(9); }                     // through the wait
#ifdef FUSE_SKIP_DESCR             // timing experiment only: the orientation pass alone in this kernel
    return false;
#endif
    return s_ok != 0u;
  };
  if (wait_coarser())
    descr_all_body<Q8, false, true>(scratch, P, counters, det, pts, max_pts, nullptr, nullptr, big_list, big_stride, s_w, nullptr,
                                    s_fdup, per_wave);
  if (omax >= 1) FUSE_STAMP_MAX(4);                                            // descriptors of a workgroup written
  // ---- the launch's last workgroup publishes the reference's counters (cudaSiftD.cu:14) and, on request, hands the blocks
  // to the host
  if (last_workgroup_2level(counters + (size_t)P.nframes * CNT_STRIDE)) {
    // (wavefront f does frame f — a single call has at most WAVES_PER_BLOCK... see the loop —: one load for all the counts,
    //  the prefix sums by lane, one store for all 2 * noct + 1 slots; system scope: these words were updated by atomics of
    //  workgroups on other XCDs during this launch)
    for (int f = wave; f < P.nframes; f += WAVES_PER_BLOCK) {
      unsigned *fc = counters + (size_t)f * CNT_STRIDE;
      unsigned v = 0u;
      if (lane < 16) v = __hip_atomic_load(&fc[32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // slot s = 2k - 1 (+0: start of octave k, +1: behind its detections), s = 2 noct + 1: behind everything
      unsigned b = 0u, mine = 0u;
      for (int k = 1; k <= P.noct; k++) {
        if (lane == 2 * k - 1) mine = b;
        b += (unsigned)__builtin_amdgcn_readlane((int)v, CNT_DET - 32 + k);
        if (lane == 2 * k) mine = b;
        b += (unsigned)__builtin_amdgcn_readlane((int)v, CNT_DUP - 32 + k);
        if (lane == 2 * k + 1) mine = b;
      }
      if (lane >= 1 && lane <= 2 * P.noct + 1) __hip_atomic_store(&fc[lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tail_host_out) export_counters_host(counters, (unsigned)P.nframes, tail_host_out, tail_seq);
    FUSE_STAMP_MAX(5);
  }
}


template <bool Q8>
__global__ __launch_bounds__(256, 4) void descr_all_gather_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                        unsigned *__restrict__ counters,
                                                        const Detection *__restrict__ det,
                                                        SiftPointD *__restrict__ pts, int max_pts, int frac8,
                                                        const int *__restrict__ pack_offsets,
                                                        SiftPointD *__restrict__ pack_dst)
{
  __shared__ __attribute__((aligned(16))) float s_smp[WAVES_PER_BLOCK][DESCR_TBL];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * max_pts;
  SiftPointD *sift = pts ? pts + (size_t)frame * max_pts : nullptr;
  // packed output: this frame's first numPts records go to pack_dst[pack_offsets[frame] ...] (what a gather ships)
  const int pack_off = pack_dst ? __builtin_amdgcn_readfirstlane(pack_offsets[frame]) : 0;
  const unsigned pack_cnt = pack_dst ? (unsigned)(__builtin_amdgcn_readfirstlane(pack_offsets[frame + 1]) - pack_off) : 0u;
  const bool q8 = Q8;
  descr_init(s_smp[wave], s_gauss[wave], lane);
  const int cell = lane >> 2;
  // segment layout of the reference: detections of octave o start where octave o-1 (incl. its duplicates) ended
  if (blockIdx.x == 0 && threadIdx.x == 0) {             // publish the reference's counters (cudaSiftD.cu:14)
    unsigned b = 0;
    for (int k = 1; k <= P.noct; k++) {
      cnt[2 * k - 1] = b;
      b += cnt[CNT_DET + k];
      cnt[2 * k] = b;
      b += cnt[CNT_DUP + k];
      cnt[2 * k + 1] = b;
    }
  }
  const FrameCounts fc = load_frame_counts(cnt, P.noct, max_pts, true);
  for (int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + wave);;
       idx += gridDim.x * WAVES_PER_BLOCK) {
    int o, i;
    if (!flat_to_octave(fc, P.noct, idx, o, i)) break;
    const OctaveInfo &L = P.o[o];
    const float *img = scratch + (long long)frame * P.frame_stride + L.img_off;
    const Detection d = fdet[(size_t)(o - 1) * max_pts + i];
    unsigned bdet = 0, bdup = 0;                          // segment bases of octave o
#pragma unroll
    for (int k = 1; k <= MISIFT_MAX_OCTAVES; k++)
      if (k == o) { bdet = fc.bdet[k]; bdup = fc.bdup[k]; }
    const unsigned ci = (unsigned)i;
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
      if (which == 1 && d.dupslot < 0) break;
      const unsigned dst = which == 0 ? bdet + ci : bdup + (unsigned)d.dupslot;
      if (dst >= (unsigned)max_pts) continue;             // capacity: dropped, still counted
      float o0, o1;
      descr_core(img, L.w, L.h, L.p, q8, d.xpos, d.ypos, d.scale, which == 0 ? d.ori1 : d.ori2, s_smp[wave],
                 s_gauss[wave], lane, o0, o1);
#pragma unroll
      for (int tgt = 0; tgt < 2; tgt++) {
        SiftPointD *p = tgt == 0 ? (sift ? &sift[dst] : nullptr)
                                 : (pack_dst && dst < pack_cnt ? pack_dst + pack_off + dst : nullptr);
        if (!p) continue;
        p->data[8 * cell + (lane & 3)] = o0;
        p->data[8 * cell + (lane & 3) + 4] = o1;
        if (lane == 0) {
          const float os = (which == 1 && o == P.noct && !P.fix_numpts) ? 1.0f : P.out_scale;   // records past numPts are not rescaled
          p->xpos = d.xpos * L.subsampling * os;       // out_scale is 1 or 0.5: exact, = a later RescalePositions
          p->ypos = d.ypos * L.subsampling * os;
          p->scale = d.scale * L.subsampling * os;
          p->sharpness = d.sharpness;
          p->edgeness = d.edgeness;
          p->orientation = which == 0 ? d.ori1 : d.ori2;
          p->subsampling = L.subsampling;
          if (tgt == 1) {                       // a packed record is complete: the match fields start out cleared
            p->score = 0.0f; p->ambiguity = 0.0f; p->match = 0; p->match_xpos = 0.0f; p->match_ypos = 0.0f;
            p->match_error = 0.0f; p->empty[0] = 0.0f; p->empty[1] = 0.0f; p->empty[2] = 0.0f;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ rescale
__global__ void rescale_kernel(SiftPointD *pts, int npts, float scale)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npts) {
    pts[i].xpos *= scale;
    pts[i].ypos *= scale;
    pts[i].scale *= scale;
  }
}

// RescalePositions over a batch (unfused path with scaleUp): frame f's first numPts records, numPts from the counters
__global__ void rescale_batch_kernel(SiftPointD *pts, int max_pts, const unsigned *__restrict__ counters, int slot, float scale)
{
  SiftPointD *p = pts + (size_t)blockIdx.y * max_pts;
  const int n = (int)min(counters[(size_t)blockIdx.y * CNT_STRIDE + slot], (unsigned)max_pts);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    p[i].xpos *= scale;
    p[i].ypos *= scale;
    p[i].scale *= scale;
  }
}

// ---- deterministic record order on the DENSE path (options.deterministic with fused = 0, or the exact re-run after a
// candidate-list overflow): every segment of the reference layout — detections of octave o, then its second
// orientations — is sorted by (ypos, xpos, scale, orientation).  Rank by counting (a segment holds a few thousand
// records at most and this path is rare), keys staged through LDS, records scattered into a scratch copy.
__device__ __forceinline__ bool sort_key_less(const float4 a, const float4 b)
{
  if (a.x != b.x) return a.x < b.x;
  if (a.y != b.y) return a.y < b.y;
  if (a.z != b.z) return a.z < b.z;
  return a.w < b.w;
}
__global__ __launch_bounds__(256) void sort_segments_kernel(const SiftPointD *__restrict__ pts, SiftPointD *__restrict__ tmp,
                                                            const unsigned *__restrict__ counters, int max_pts)
{
  __shared__ float4 s_key[256];
  const int frame = blockIdx.z, seg = blockIdx.y, o = seg / 2 + 1;
  const unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const int lo = (int)min(cnt[2 * o - 1 + (seg & 1)], (unsigned)max_pts), hi = (int)min(cnt[2 * o + (seg & 1)], (unsigned)max_pts);
  const SiftPointD *P = pts + (size_t)frame * max_pts;
  SiftPointD *T = tmp + (size_t)frame * max_pts;
  for (int base = lo + blockIdx.x * 256; base < hi; base += gridDim.x * 256) {     // block-uniform trip count
    const int i = base + threadIdx.x;
    const bool live = i < hi;
    float4 ki = make_float4(0, 0, 0, 0);
    if (live) ki = make_float4(P[i].ypos, P[i].xpos, P[i].scale, P[i].orientation);
    int rank = 0;
    for (int t0 = lo; t0 < hi; t0 += 256) {
      const int j = t0 + threadIdx.x;
      __syncthreads();
      if (j < hi) s_key[threadIdx.x] = make_float4(P[j].ypos, P[j].xpos, P[j].scale, P[j].orientation);
      __syncthreads();
      const int m = min(256, hi - t0);
      for (int k = 0; k < m; k++) {
        const float4 kj = s_key[k];
        const bool eq = kj.x == ki.x && kj.y == ki.y && kj.z == ki.z && kj.w == ki.w;
        rank += (sort_key_less(kj, ki) || (eq && t0 + k < i)) ? 1 : 0;
      }
    }
    if (live) {
      const float4 *src = reinterpret_cast<const float4 *>(&P[i]);
      float4 *dst = reinterpret_cast<float4 *>(&T[lo + rank]);
#pragma unroll 4
      for (int q = 0; q < (int)(sizeof(SiftPointD) / 16); q++) dst[q] = src[q];
    }
  }
}
__global__ __launch_bounds__(256) void sort_copy_back_kernel(SiftPointD *__restrict__ pts, const SiftPointD *__restrict__ tmp,
                                                             const unsigned *__restrict__ counters, int max_pts, int last_slot)
{
  const int frame = blockIdx.y;
  const unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const size_t n = (size_t)min(cnt[last_slot], (unsigned)max_pts) * (sizeof(SiftPointD) / 16);
  float4 *dst = reinterpret_cast<float4 *>(pts + (size_t)frame * max_pts);
  const float4 *src = reinterpret_cast<const float4 *>(tmp + (size_t)frame * max_pts);
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) dst[q] = src[q];
}

// ------------------------------------------------------------- host wrappers
static inline int points_grid_x(misift_ctx *ctx, int nframes, int blocks_per_cu = 0)
{
  if (blocks_per_cu <= 0) blocks_per_cu = ctx->point_blocks_per_cu;
  // enough wavefronts to cover a few thousand keypoints per frame while keeping
  // the total near a few waves per SIMD when many frames are batched
  int per_frame = (ctx->num_cus * blocks_per_cu + nframes - 1) / nframes;
  if (per_frame < 8) per_frame = 8;
  // (batches: 512 workgroups per frame at most; a frame or two: 1024, i.e. one wavefront per keypoint up to 4096 — a
  //  second keypoint per wavefront doubles the latency of the whole launch)
  const int cap = nframes <= ctx->small_frames ? 1024 : 512;
  if (per_frame > cap) per_frame = cap;
  return per_frame;
}

// the texture-weight quantisation (texfrac_bits 8 / 23) is a template parameter of the kernels
#define LAUNCH_Q8(kernel, grid, block, ...)                                                       \
  do {                                                                                            \
    if (ctx->opt.texfrac_bits == 8) hipLaunchKernelGGL(kernel<true>, grid, block, 0, ctx->stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL(kernel<false>, grid, block, 0, ctx->stream, __VA_ARGS__);             \
  } while (0)

int launch_orient(misift_ctx *ctx, const float *base, long long base_frame_stride, int w, int h, int pitch,
                  int nframes, int octave, SiftPointD *pts, int max_pts)
{
  LaunchScope ls(ctx, "orient");
  LAUNCH_Q8(orient_kernel, dim3(points_grid_x(ctx, nframes), nframes), dim3(256), base, base_frame_stride, w, h, pitch,
            octave, ctx->d_counters, pts, max_pts, 0);
  return ls.finish();
}

int launch_descr(misift_ctx *ctx, const float *base, long long base_frame_stride, int w, int h, int pitch,
                 int nframes, float subsampling, int octave, SiftPointD *pts, int max_pts)
{
  LaunchScope ls(ctx, "descr");
  LAUNCH_Q8(descr_kernel, dim3(points_grid_x(ctx, nframes), nframes), dim3(256), base, base_frame_stride, w, h, pitch,
            subsampling, octave, ctx->d_counters, pts, max_pts, 0);
  return ls.finish();
}

static int prepare_block_maps(misift_ctx *ctx, const PyramidInfo &P, bool *wanted);

int launch_bin_detections(misift_ctx *ctx, const PyramidInfo &P, int max_pts)
{
  // a balanced batch: the block tables of orient_all / descr_all are built by one extra workgroup of this launch
  bool balanced = false;
  int rc = prepare_block_maps(ctx, P, &balanced);
  if (rc) return rc;
  FrameSharesArgs sh = {0, 0, nullptr, nullptr};
  if (balanced) {
    sh.t_a = ctx->map_t_orient; sh.t_b = ctx->map_t_descr;
    sh.map_a = ctx->d_block_map; sh.map_b = ctx->d_block_map + ctx->map_t_orient;
  }
  LaunchScope ls(ctx, "bin_detections");
  hipLaunchKernelGGL(bin_detections_kernel, dim3(P.noct + (balanced ? 1 : 0), P.nframes), dim3(256), 0, ctx->stream, P,
                     ctx->d_counters, ctx->d_det, ctx->d_det_sorted, max_pts, ctx->opt.deterministic ? 1 : 0, sh);
  rc = ls.finish();
  if (rc == MISIFT_OK && balanced) ctx->cur_balanced = 1;
  return rc;
}

int launch_renumber_dups(misift_ctx *ctx, const PyramidInfo &P, int max_pts)
{
  LaunchScope ls(ctx, "renumber_dups");
  hipLaunchKernelGGL(renumber_dups_kernel, dim3(P.noct, P.nframes), dim3(256), 0, ctx->stream, P, ctx->d_counters,
                     ctx->d_det_sorted, max_pts);
  return ls.finish();
}

// Balanced batches (MISIFT_BALANCE=1): the block tables of this call's orient_all and descr_all launches, written by one
// small kernel behind refine_all (the per-octave detection counts are final there; second orientations are done by the
// wavefront of their keypoint, so the same counts weigh both launches).
static int prepare_block_maps(misift_ctx *ctx, const PyramidInfo &P, bool *wanted)
{
  *wanted = false;
  if (!ctx->balance_frames || P.nframes <= ctx->small_frames || ctx->in_capture || ctx->tile_orient || !ctx->tile_descr)
    return MISIFT_OK;
  const int t_orient = points_grid_x(ctx, P.nframes, ctx->orient_blocks_per_cu) * P.nframes;
  const int t_descr = points_grid_x(ctx, P.nframes) * P.nframes;
  if (ctx->block_map_cap < t_orient + t_descr) {
    if (ctx->d_block_map) HIP_TRY(misift_dev_free(ctx->d_block_map));
    ctx->d_block_map = nullptr; ctx->block_map_cap = 0;
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_block_map, sizeof(int4) * (size_t)(t_orient + t_descr), "block_maps"));
    ctx->block_map_cap = t_orient + t_descr;
    ctx->alloc_gen++;
  }
  ctx->map_t_orient = t_orient; ctx->map_t_descr = t_descr;
  *wanted = true;
  return MISIFT_OK;
}

// ... or, when the batch is not binned (MISIFT_BIN=0), by a small kernel of their own in front of orient_all
static int build_block_maps(misift_ctx *ctx, const PyramidInfo &P, int max_pts)
{
  if (ctx->cur_balanced) return MISIFT_OK;            // launch_bin_detections has built them
  bool wanted = false;
  int rc = prepare_block_maps(ctx, P, &wanted);
  if (rc || !wanted) return rc;
  LaunchScope ls(ctx, "frame_shares");
  hipLaunchKernelGGL(frame_shares_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_counters, P.nframes, P.noct, max_pts,
                     ctx->map_t_orient, ctx->d_block_map, ctx->map_t_descr, ctx->d_block_map + ctx->map_t_orient);
  rc = ls.finish();
  if (rc == MISIFT_OK) ctx->cur_balanced = 1;
  return rc;
}

int launch_orient_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts)
{
  (void)pts;
  {
    int rc = build_block_maps(ctx, P, max_pts);
    if (rc) return rc;
  }
  LaunchScope ls(ctx, "orient_all");
  Detection *det = ctx->cur_binned ? ctx->d_det_sorted : ctx->d_det;
  const dim3 grid(points_grid_x(ctx, P.nframes, ctx->orient_blocks_per_cu), P.nframes);
  const bool q8 = ctx->opt.texfrac_bits == 8;
  if (ctx->tile_orient) LAUNCH_Q8(orient_all_kernel, grid, dim3(256), scratch, P, ctx->d_counters, det, max_pts, 0);
  else if (ctx->cur_balanced) {
    const dim3 flat(ctx->map_t_orient);
    if (q8) hipLaunchKernelGGL((orient_all_gather_kernel<true, true>), flat, dim3(256), (size_t)ctx->lds_pad_orient, ctx->stream, scratch, P,
                               ctx->d_counters, det, max_pts, 0, ctx->d_block_map);
    else hipLaunchKernelGGL((orient_all_gather_kernel<false, true>), flat, dim3(256), (size_t)ctx->lds_pad_orient, ctx->stream, scratch, P,
                            ctx->d_counters, det, max_pts, 0, ctx->d_block_map);
  } else {
    if (q8) hipLaunchKernelGGL((orient_all_gather_kernel<true, false>), grid, dim3(256), (size_t)ctx->lds_pad_orient, ctx->stream, scratch, P,
                               ctx->d_counters, det, max_pts, 0, (const int4 *)nullptr);
    else hipLaunchKernelGGL((orient_all_gather_kernel<false, false>), grid, dim3(256), (size_t)ctx->lds_pad_orient, ctx->stream, scratch, P,
                            ctx->d_counters, det, max_pts, 0, (const int4 *)nullptr);
  }
  return ls.finish();
}

int launch_descr_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts,
                     const int *pack_offsets, SiftPointD *pack_dst)
{
  LaunchScope ls(ctx, "descr_all");
  const Detection *det = ctx->cur_binned ? ctx->d_det_sorted : ctx->d_det;
  const dim3 grid(points_grid_x(ctx, P.nframes), P.nframes);
  if (ctx->tile_descr) {
    // keypoints too large for the LDS window go to a per-frame list in the (by now idle) candidate buffer
    unsigned big_stride = 0;
    for (int o = 1; o <= P.noct; o++) big_stride += P.o[o].cand_cap;
    // the last kernel of the call: on request it also hands the counter blocks to the host (export_counters_host)
    unsigned *host_out = nullptr;
    if (ctx->want_export && !ctx->in_capture) {
      host_out = ctx->h_counters;
      ctx->export_seq++;
      ctx->exported = 1;
    }
    const bool bal = ctx->cur_balanced != 0;
    // single-call path: descr_all itself is the last kernel (its last workgroup exports); descr_big follows only if the
    // host finds a deferred keypoint in the exported counters (launch_descr_big_pending, called by read_counts)
    const bool fold = host_out && !bal && ctx->fold_descr_tail && P.nframes <= ctx->small_frames;
    const dim3 dgrid = bal ? dim3(ctx->map_t_descr) : grid;
    const int4 *bmap = bal ? ctx->d_block_map + ctx->map_t_orient : nullptr;
    unsigned *tail_out = fold ? host_out : nullptr;
#define DESCR_LAUNCH(Q, O, B) hipLaunchKernelGGL((descr_all_kernel<Q, O, B>), dgrid, dim3(256), (size_t)ctx->lds_pad_descr, ctx->stream, scratch, P, \
                                                 ctx->d_counters, det, pts, max_pts, 0, pack_offsets, pack_dst, ctx->d_cand, \
                                                 big_stride, bmap, tail_out, ctx->export_seq)
#define DESCR_LAUNCH_B(Q, O) do { if (bal) DESCR_LAUNCH(Q, O, true); else DESCR_LAUNCH(Q, O, false); } while (0)
    const bool q8 = ctx->opt.texfrac_bits == 8;
    if (ctx->descr_occ >= 4) { if (q8) DESCR_LAUNCH_B(true, 4); else DESCR_LAUNCH_B(false, 4); }
    else { if (q8) DESCR_LAUNCH_B(true, 3); else DESCR_LAUNCH_B(false, 3); }
#undef DESCR_LAUNCH_B
#undef DESCR_LAUNCH
    PendingBig &pb = ctx->pending_big;
    pb.valid = 0;
    if (fold) {
      pb.valid = 1; pb.scratch = scratch; pb.P = P; pb.det = det; pb.pts = pts; pb.max_pts = max_pts;
      pb.pack_offsets = pack_offsets; pb.pack_dst = pack_dst; pb.big_stride = big_stride;
    } else {
      LAUNCH_Q8(descr_big_kernel, dim3(2, P.nframes), dim3(256), scratch, P, ctx->d_counters, det, pts, max_pts,
                pack_offsets, pack_dst, ctx->d_cand, big_stride, host_out, ctx->export_seq);
    }
  } else
    LAUNCH_Q8(descr_all_gather_kernel, grid, dim3(256), scratch, P, ctx->d_counters, det, pts, max_pts, 0, pack_offsets,
              pack_dst);
  return ls.finish();
}

// A single call's orientations + descriptors as ONE launch (orient_descr_fused_kernel).  Always the call's last kernel in the
// sense of launch_descr_all's `fold`: its last workgroup publishes the counters and, on request, exports them; descr_big
// follows at once when nobody will look at the exported counters first.
int launch_orient_descr_fused(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts)
{
  LaunchScope ls(ctx, "orient_descr");
  Detection *det = ctx->d_det;
  const dim3 grid(points_grid_x(ctx, P.nframes), P.nframes);
  unsigned big_stride = 0;
  for (int o = 1; o <= P.noct; o++) big_stride += P.o[o].cand_cap;
  unsigned *host_out = nullptr;
  if (ctx->want_export && !ctx->in_capture) {
    host_out = ctx->h_counters;
    ctx->export_seq++;
    ctx->exported = 1;
  }
  const bool fold = host_out && ctx->fold_descr_tail;
  unsigned *tail_out = fold ? host_out : nullptr;
  const bool q8 = ctx->opt.texfrac_bits == 8;
#define FUSED_LAUNCH(Q, O) hipLaunchKernelGGL((orient_descr_fused_kernel<Q, O>), grid, dim3(256), (size_t)ctx->lds_pad_descr, ctx->stream, \
                                              scratch, P, ctx->d_counters, det, pts, max_pts, ctx->d_cand, big_stride,                \
                                              ctx->fuse_wait_ticks, tail_out, ctx->export_seq)
  if (!q8) FUSED_LAUNCH(false, 3);               // (full-precision weights at 4 workgroups per CU would spill 17 registers)
  else if (ctx->descr_occ >= 4) FUSED_LAUNCH(true, 4);
  else FUSED_LAUNCH(true, 3);
#undef FUSED_LAUNCH
  PendingBig &pb = ctx->pending_big;
  pb.valid = 0;
  if (fold) {
    pb.valid = 1; pb.scratch = scratch; pb.P = P; pb.det = det; pb.pts = pts; pb.max_pts = max_pts;
    pb.pack_offsets = nullptr; pb.pack_dst = nullptr; pb.big_stride = big_stride;
  } else {
    LAUNCH_Q8(descr_big_kernel, dim3(2, P.nframes), dim3(256), scratch, P, ctx->d_counters, det, pts, max_pts,
              (const int *)nullptr, (SiftPointD *)nullptr, ctx->d_cand, big_stride, host_out, ctx->export_seq);
  }
  return ls.finish();
}

// The rare second half of a folded single call: the host found keypoints deferred to descr_big in the counters descr_all's
// last workgroup exported.  Runs descr_big_kernel on the launch recorded by launch_descr_all; it exports the counters again
// under a new sequence number (the caller waits for that one).
int launch_descr_big_pending(misift_ctx *ctx)
{
  PendingBig &pb = ctx->pending_big;
  if (!pb.valid) return MISIFT_OK;
  pb.valid = 0;
  ctx->export_seq++;
  ctx->descr_big_fallbacks++;
  const float *scratch = pb.scratch;
  const PyramidInfo &P = pb.P;
  LaunchScope ls(ctx, "descr_big");
  LAUNCH_Q8(descr_big_kernel, dim3(2, P.nframes), dim3(256), scratch, P, ctx->d_counters, pb.det, pb.pts, pb.max_pts,
            pb.pack_offsets, pb.pack_dst, ctx->d_cand, pb.big_stride, ctx->h_counters, ctx->export_seq);
  return ls.finish();
}

int launch_rescale_batch(misift_ctx *ctx, SiftPointD *pts, int max_pts, int nframes, int num_octaves, float scale)
{
  LaunchScope ls(ctx, "rescale");
  const int slot = 2 * num_octaves + (ctx->opt.fix_numpts ? 1 : 0);
  hipLaunchKernelGGL(rescale_batch_kernel, dim3(16, nframes), dim3(256), 0, ctx->stream, pts, max_pts, ctx->d_counters,
                     slot, scale);
  return ls.finish();
}

// options.deterministic on the dense path: see sort_segments_kernel
int launch_sort_segments(misift_ctx *ctx, SiftPointD *pts, int max_pts, int nframes, int num_octaves)
{
  int rc = misift_ensure_tmp(ctx, sizeof(SiftPointD) * (size_t)nframes * max_pts);
  if (rc) return rc;
  LaunchScope ls(ctx, "sort_segments");
  SiftPointD *tmp = reinterpret_cast<SiftPointD *>(ctx->d_match_tmp);
  hipLaunchKernelGGL(sort_segments_kernel, dim3(8, 2 * num_octaves, nframes), dim3(256), 0, ctx->stream, pts, tmp,
                     ctx->d_counters, max_pts);
  hipLaunchKernelGGL(sort_copy_back_kernel, dim3(64, nframes), dim3(256), 0, ctx->stream, pts, tmp, ctx->d_counters, max_pts,
                     2 * num_octaves + 1);
  return ls.finish();
}

int launch_rescale(misift_ctx *ctx, SiftPointD *pts, int npts, float scale)
{
  if (npts <= 0) return MISIFT_OK;
  LaunchScope ls(ctx, "rescale");
  hipLaunchKernelGGL(rescale_kernel, dim3((npts + 255) / 256), dim3(256), 0, ctx->stream, pts, npts, scale);
  return ls.finish();
}

// ---- test-only: the device det_atan2 / det_exp / det_sincos on caller-supplied inputs (misift_test_elementary)
__global__ void test_points_fn_kernel(int fn, const float *__restrict__ x, const float *__restrict__ y,
                                      float *__restrict__ out, float *__restrict__ out2, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (fn == 1) out[i] = det_atan2(y[i], x[i]);
  else if (fn == 2) out[i] = det_exp(x[i]);
  else {
    float sn, cs;
    det_sincos(x[i], sn, cs);
    out[i] = sn;
    out2[i] = cs;
  }
}
int launch_test_points_fn(misift_ctx *ctx, int fn, const float *x, const float *y, float *out, float *out2, int n)
{
  hipLaunchKernelGGL(test_points_fn_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fn, x, y, out, out2, n);
  HIP_TRY(hipGetLastError());
  return MISIFT_OK;
}

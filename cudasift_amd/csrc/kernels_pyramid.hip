// kernels_pyramid.hip — Gaussian pyramid kernels for gfx950 (wave64).
//
//   lowpass_kernel    replaces LowPassBlock   (reference cudaSiftD.cu:1986-2037, host cudaSiftH.cu:406-435)
//   scaledown_kernel  replaces ScaleDown      (reference cudaSiftD.cu:84-168,    host cudaSiftH.cu:308-338)
//   scaleup_kernel    replaces ScaleUp        (reference cudaSiftD.cu:170-190,   host cudaSiftH.cu:340-351)
//
// Design (not a translation of the reference's 32-lane shuffle / 16-row ring
// tilings): every wavefront streams down a 256-pixel-wide strip.  A lane owns a
// quad (float4) per row, so each image row is one coalesced 1 KiB dwordx4 read
// per wavefront; the radius-4 horizontal neighbourhood is exactly the two
// adjacent lanes' quads, fetched with DPP wave shifts (no LDS, no re-read); the
// vertical filter window lives in VGPRs and slides one row per step, so every
// input row is read once per segment and every output row written once.
// HBM-bound: algorithmic bytes are 8 B/px (lowpass) and 5 B/px of input (scaledown).
//
// Arithmetic is the explicit fmaf chain of oracle/sift_oracle.c (bit-identical).  Since r03 the chains of LowPass and
// ScaleDown follow what a contracting compiler makes of the reference's single-expression sums (left product fused,
// right product rounded: conv9_expr) — pinned bit for bit by the reference's own kernels compiled that way and run on
// the CPU SIMT emulator (oracle/_ref/libcudasift_refemul_fast.so, tests/test_refemul_cpu.py).
#include <string.h>
#include "common.hpp"
#include "chain.hpp"

#define WAVES_PER_BLOCK 4
#define OUT_LANES 62

struct ItemCoord { int frame, strip, seg; bool valid; };

__device__ __forceinline__ ItemCoord decode_item(const StripGeom &g)
{
  const unsigned lb = (g.noremap & 1) ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  // the wave index is wave-uniform: keep it (and everything derived from it — frame, strip, segment,
  // row bounds, row pointers) in SGPRs.  Besides cheaper scalar loop control this keeps the loop bounds
  // out of reach of VGPR live-range splitting around divergent regions.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long item = (long long)lb * WAVES_PER_BLOCK + wave;
  const long long nitems = (long long)g.nframes * g.nstrips * g.nsegs;
  ItemCoord c;
  c.valid = item < nitems;
  if (g.noremap & 2) {             // strip-fastest: the 4 waves of a workgroup read 4 KB contiguous per row
    c.strip = (int)(item % g.nstrips);
    const long long r = item / g.nstrips;
    c.seg = (int)(r % g.nsegs);
    c.frame = (int)(r / g.nsegs);
  } else {
    c.seg = (int)(item % g.nsegs);
    const long long r = item / g.nsegs;
    c.strip = (int)(r % g.nstrips);
    c.frame = (int)(r / g.nstrips);
  }
  return c;
}

// The prefilter is the first kernel of an extraction: the wavefront that owns a frame's first item also clears the
// frame's counter block (CNT_STRIDE = 64 words, one per lane), which the scan — a later kernel of the same stream — is
// the first to touch.  Replaces a hipMemsetAsync: one dependent dispatch less per call (r04 single-call budget).
// (frame 0's wavefront also clears the spare blocks behind the last frame's: flags of the call, see dog_scan_all_kernel)
__device__ __forceinline__ void zero_frame_counters(unsigned *zero_cnt, const ItemCoord &it, int lane, int nframes)
{
  if (zero_cnt && it.strip == 0 && it.seg == 0) {
    zero_cnt[(size_t)it.frame * CNT_STRIDE + lane] = 0u;
    if (it.frame == 0)
      for (int b = 0; b < CNT_SPARE_BLOCKS; b++) zero_cnt[(size_t)(nframes + b) * CNT_STRIDE + lane] = 0u;
  }
}

__device__ __forceinline__ void store_quad(float *row, int q, int width, bool aligned, float4 v)
{
  const int x = 4 * q;
  if (aligned && x + 3 < width) {
    *reinterpret_cast<float4 *>(row + x) = v;
  } else {
    if (x < width) row[x] = v.x;
    if (x + 1 < width) row[x + 1] = v.y;
    if (x + 2 < width) row[x + 2] = v.z;
    if (x + 3 < width) row[x + 3] = v.w;
  }
}

// ------------------------------------------------------------------ LowPass
// out = G9^T (vertical) applied to G9 (horizontal) applied to in, clamp-to-edge.
template <bool FAST, typename SRC>
__global__ __launch_bounds__(256, 4) void lowpass_kernel(const SRC *__restrict__ src, StripGeom g,
                                                      float *__restrict__ dst, int dpitch,
                                                      long long dst_frame_stride, Taps5 t, int src_aligned,
                                                      int dst_aligned, unsigned *__restrict__ zero_cnt)
{
  const ItemCoord it = decode_item(g);
  if (!it.valid) return;
  const int lane = threadIdx.x & 63;
  zero_frame_counters(zero_cnt, it, lane, g.nframes);
  const int q = it.strip * OUT_LANES + lane - 1;
  const SRC *img = src + (long long)it.frame * g.frame_stride;
  float *out = dst + (long long)it.frame * dst_frame_stride;
  const int y0 = it.seg * g.seg_rows;
  const int y1 = min(y0 + g.seg_rows, g.height);
  const float k0 = t.k[0], k1 = t.k[1], k2 = t.k[2], k3 = t.k[3], k4 = t.k[4];
  const bool sal = src_aligned != 0, dal = dst_aligned != 0;
  const QuadCol qc = make_quadcol(q, g.width);
  auto ldraw = [&](int y) -> float4 {
    return load_quad_t<FAST>(img + (size_t)clampi(y, 0, g.height - 1) * g.pitch, q, g.width, sal, qc);
  };

  auto hfilt = [&](const float4 c) -> float4 {
    const float4 l = quad_from_left(c);
    const float4 r = quad_from_right(c);
    float4 h;
    h.x = conv9_expr(k0, k1, k2, k3, k4, c.x, c.y + l.w, c.z + l.z, c.w + l.y, r.x + l.x);
    h.y = conv9_expr(k0, k1, k2, k3, k4, c.y, c.z + c.x, c.w + l.w, r.x + l.z, r.y + l.y);
    h.z = conv9_expr(k0, k1, k2, k3, k4, c.z, c.w + c.y, r.x + c.x, r.y + l.w, r.z + l.z);
    h.w = conv9_expr(k0, k1, k2, k3, k4, c.w, r.x + c.z, r.y + c.y, r.z + c.x, r.w + l.w);
    return h;
  };
  auto hrow = [&](int y) -> float4 { return hfilt(ldraw(y)); };

  float4 w0 = hrow(y0 - 4), w1 = hrow(y0 - 3), w2 = hrow(y0 - 2), w3 = hrow(y0 - 1), w4 = hrow(y0);
  float4 w5 = hrow(y0 + 1), w6 = hrow(y0 + 2), w7 = hrow(y0 + 3), w8;
  // two rows of loads stay in flight per wavefront: with ~4 resident waves per SIMD one row ahead only
  // keeps ~4 MB in flight chip-wide, which caps the stream near 4 TB/s at ~1 us of HBM latency
  float4 raw = ldraw(y0 + 4), raw1 = ldraw(y0 + 5);
  const bool writer = lane >= 1 && lane <= OUT_LANES && 4 * q < g.width;
  for (int y = y0; y < y1; y++) {
    const float4 rawnext = ldraw(y + 6);
    w8 = hfilt(raw);
    raw = raw1;
    raw1 = rawnext;
    float4 o;
    o.x = conv9_expr(k0, k1, k2, k3, k4, w4.x, w3.x + w5.x, w2.x + w6.x, w1.x + w7.x, w0.x + w8.x);
    o.y = conv9_expr(k0, k1, k2, k3, k4, w4.y, w3.y + w5.y, w2.y + w6.y, w1.y + w7.y, w0.y + w8.y);
    o.z = conv9_expr(k0, k1, k2, k3, k4, w4.z, w3.z + w5.z, w2.z + w6.z, w1.z + w7.z, w0.z + w8.z);
    o.w = conv9_expr(k0, k1, k2, k3, k4, w4.w, w3.w + w5.w, w2.w + w6.w, w1.w + w7.w, w0.w + w8.w);
    if (writer) {
      if (FAST) *reinterpret_cast<float4 *>(out + (size_t)y * dpitch + 4 * q) = o;
      else store_quad(out + (size_t)y * dpitch, q, g.width, dal, o);
    }
    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7; w7 = w8;
  }
}

// ------------------------------------------------- LowPass + first ScaleDown, fused
// The prefiltered image is consumed twice downstream: by the finest octave's DoG scan and by the
// first ScaleDown.  This kernel emits BOTH the prefiltered image and its 2x decimation while the
// prefiltered rows are still in registers, so the finest pyramid level is never re-read for the
// decimation (-8.3 MB of HBM reads per 1080p frame, and one launch less).  Same arithmetic as
// lowpass_kernel followed by scaledown_kernel (bit-identical): each lane turns its prefiltered quad
// (px 4q..4q+3) into the two horizontally decimated values X = 2q, 2q+1 (neighbour pixels by DPP), a
// 5-row ring of those feeds the vertical 5-tap of every second row.  Fast path only (width % 4 == 0,
// aligned rows); other shapes run the two separate kernels.
// Lanes 1..62 hold valid prefiltered quads, lanes 2..61 store (the decimation needs both neighbours),
// so a strip advances by 60 quads.
// The prefiltered rows (8.3 MB per 1080p frame, read back by the scan from HBM anyway: the L2 cannot hold them) leave as
// non-temporal stores: r04 A/B over six alternating pairs on two boxes, default bench: 58.5 k +- 0.1 against 57.8 k +- 0.8
// frames/s (tools/ab_bench.sh; LPD_NT = 2 adds the decimated rows: 57.4 k, noisier; r01's rejection of non-temporal stores
// predates batches in flight).
#ifndef LPD_NT
#define LPD_NT 1
#endif
#define FUSED_OUT_LANES 60
// MODE 1: width % 4 == 0; MODE 2: any width (ragged last quad, see clamp_quad in common.hpp).
template <typename SRC, int MODE>
__device__ __forceinline__ void lowpass_down_item(const SRC *__restrict__ src, const StripGeom &g,
                                                  float *__restrict__ dst, int dpitch,
                                                  long long dst_frame_stride, const Taps5 &t,
                                                  float *__restrict__ dst2, int dpitch2,
                                                  long long dst2_frame_stride, const Taps5 &t5,
                                                  unsigned *__restrict__ zero_cnt, const ItemCoord &it)
{
  const int lane = threadIdx.x & 63;
  zero_frame_counters(zero_cnt, it, lane, g.nframes);
  const int q = it.strip * FUSED_OUT_LANES + lane - 2;
  const SRC *img = src + (long long)it.frame * g.frame_stride;
  float *out = dst + (long long)it.frame * dst_frame_stride;
  float *out2 = dst2 + (long long)it.frame * dst2_frame_stride;
  const int y0 = it.seg * g.seg_rows;                      // seg_rows is even
  const int y1 = min(y0 + g.seg_rows, g.height);
  const int h2 = g.height / 2;
  const float k0 = t.k[0], k1 = t.k[1], k2 = t.k[2], k3 = t.k[3], k4 = t.k[4];
  const float d0 = t5.k[0], d1 = t5.k[1], d2 = t5.k[2];   // t5.k[2] = centre tap (reference order)
  const QuadCol qc = make_quadcol(q, g.width);
  auto ldraw = [&](int y) -> float4 {
    return load_quad_t<MODE>(img + (size_t)clampi(y, 0, g.height - 1) * g.pitch, q, g.width, true, qc);
  };
  auto hfilt = [&](const float4 c) -> float4 {
    const float4 l = quad_from_left(c);
    const float4 r = quad_from_right(c);
    float4 h;
    h.x = conv9_expr(k0, k1, k2, k3, k4, c.x, c.y + l.w, c.z + l.z, c.w + l.y, r.x + l.x);
    h.y = conv9_expr(k0, k1, k2, k3, k4, c.y, c.z + c.x, c.w + l.w, r.x + l.z, r.y + l.y);
    h.z = conv9_expr(k0, k1, k2, k3, k4, c.z, c.w + c.y, r.x + c.x, r.y + l.w, r.z + l.z);
    h.w = conv9_expr(k0, k1, k2, k3, k4, c.w, r.x + c.z, r.y + c.y, r.z + c.x, r.w + l.w);
    return h;
  };
  auto hrow = [&](int y) -> float4 { return hfilt(ldraw(y)); };
  const bool first_quad = q == 0, last_quad = 4 * q + 4 >= g.width;
  // horizontal 5-tap decimation of a prefiltered row (scaledown_kernel's hfilt, clamp-to-edge in image space)
  auto hdec = [&](const float4 o) -> float2 {
    float lz = lane_from_left(o.z), lw = lane_from_left(o.w), rx = lane_from_right(o.x);
    if (first_quad) { lz = o.x; lw = o.x; }
    if (last_quad) rx = o.w;
    float2 h;
    float s;
    s = __builtin_fmaf(d0, lz + o.z, d1 * (lw + o.y));  h.x = __builtin_fmaf(d2, o.x, s);
    s = __builtin_fmaf(d0, o.x + rx, d1 * (o.y + o.w)); h.y = __builtin_fmaf(d2, o.z, s);
    return h;
  };
  auto vcomb = [&](float a0, float a1, float a2, float a3, float a4) -> float {
    float s = __builtin_fmaf(d2, a2, d0 * (a0 + a4));
    s = __builtin_fmaf(d1, a1 + a3, s);
    return s;
  };
  const bool writer = lane >= 2 && lane <= FUSED_OUT_LANES + 1 && 4 * q < g.width;
  auto emit = [&](int Y, float2 a0, float2 a1, float2 a2, float2 a3, float2 a4) {
    if (writer && Y < h2) {
      float2 o;
      o.x = vcomb(a0.x, a1.x, a2.x, a3.x, a4.x);
      o.y = vcomb(a0.y, a1.y, a2.y, a3.y, a4.y);
      float *d2 = out2 + (size_t)Y * dpitch2 + 2 * q;
      if (MODE == 2) {               // ragged width: the lane of the last quad may own only one (or no) decimated column,
        const int w2o = g.width >> 1;     // and the next one may already be the neighbouring row (pitch == width/2)
        if (2 * q + 1 < w2o) *reinterpret_cast<float2 *>(d2) = o;
        else if (2 * q < w2o) d2[0] = o.x;
      } else {
#if LPD_NT >= 2
        typedef float lpd_v2f __attribute__((ext_vector_type(2)));
        lpd_v2f o2; o2.x = o.x; o2.y = o.y;
        __builtin_nontemporal_store(o2, reinterpret_cast<lpd_v2f *>(d2));
#else
        *reinterpret_cast<float2 *>(d2) = o;
#endif
      }
    }
  };

  const int rs = max(y0 - 2, 0);                           // prefiltered rows rs .. re feed this segment's decimated rows
  const int re = min(y1, g.height - 1);
  // The 9-row window of horizontally filtered rows and the three raw rows in flight rotate by NAME: the row loop is
  // unrolled nine times (window slot of row y+k = (j+k) % 9, compile-time), so no register is ever moved.  r02's loop
  // rotated them with 48 v_mov per row — 37 % of its 130 VALU instructions (r03: ISA count of the loop body).
  float4 w[9], rw[3];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = hrow(rs - 4 + k);
  rw[0] = ldraw(rs + 4);
  rw[1] = ldraw(rs + 5);
  float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0, a4 = a0;
  // one row: window rows y-4 .. y+3 in W0..W7, W8 receives row y+4 (= hfilt of RCUR); RNEXT receives the raw row y+6
  auto row = [&](const int y, const float4 &W0, const float4 &W1, const float4 &W2, const float4 &W3, const float4 &W4,
                 const float4 &W5, const float4 &W6, const float4 &W7, float4 &W8, const float4 &RCUR,
                 float4 &RNEXT) __attribute__((always_inline)) {
    RNEXT = ldraw(y + 6);
    W8 = hfilt(RCUR);
    float4 o;
    o.x = conv9_expr(k0, k1, k2, k3, k4, W4.x, W3.x + W5.x, W2.x + W6.x, W1.x + W7.x, W0.x + W8.x);
    o.y = conv9_expr(k0, k1, k2, k3, k4, W4.y, W3.y + W5.y, W2.y + W6.y, W1.y + W7.y, W0.y + W8.y);
    o.z = conv9_expr(k0, k1, k2, k3, k4, W4.z, W3.z + W5.z, W2.z + W6.z, W1.z + W7.z, W0.z + W8.z);
    o.w = conv9_expr(k0, k1, k2, k3, k4, W4.w, W3.w + W5.w, W2.w + W6.w, W1.w + W7.w, W0.w + W8.w);
#if LPD_NT
    if (writer && y >= y0 && y < y1) {
      typedef float lpd_v4f __attribute__((ext_vector_type(4)));
      lpd_v4f o4; o4.x = o.x; o4.y = o.y; o4.z = o.z; o4.w = o.w;
      __builtin_nontemporal_store(o4, reinterpret_cast<lpd_v4f *>(out + (size_t)y * dpitch + 4 * q));
    }
#else
    if (writer && y >= y0 && y < y1) *reinterpret_cast<float4 *>(out + (size_t)y * dpitch + 4 * q) = o;
#endif
    float4 oc = o;                                         // the prefiltered row as ScaleDown's clamped reads see it:
    if (MODE == 2 && qc.edge == 2) {                       // columns past width-1 take the value of column width-1
      const float e = qc.rem == 1 ? o.x : (qc.rem == 2 ? o.y : o.z);
      if (qc.rem < 2) oc.y = e;
      if (qc.rem < 3) oc.z = e;
      oc.w = e;
    }
    const float2 hd = hdec(oc);
    a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = hd;
    if (y == 0) { a2 = hd; a3 = hd; }                     // rows -2, -1 clamp to row 0
    if ((y & 1) == 0 && y >= y0 + 2) emit((y - 2) >> 1, a0, a1, a2, a3, a4);
  };
  int y = rs;
  for (; y + 8 <= re; y += 9) {
#pragma unroll
    for (int j = 0; j < 9; j++)
      row(y + j, w[j % 9], w[(j + 1) % 9], w[(j + 2) % 9], w[(j + 3) % 9], w[(j + 4) % 9], w[(j + 5) % 9],
          w[(j + 6) % 9], w[(j + 7) % 9], w[(j + 8) % 9], rw[j % 3], rw[(j + 2) % 3]);
  }
  for (; y <= re; y++) {                                   // fewer than nine rows left: rotate the registers
    row(y, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], rw[0], rw[2]);
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = w[k + 1];
    rw[0] = rw[1];
    rw[1] = rw[2];
  }
  // last segment of an even-height image: row `height` clamps to row height-1
  if (y1 == g.height && (g.height & 1) == 0) emit((g.height - 2) >> 1, a1, a2, a3, a4, a4);
}

template <typename SRC, int MODE>
__global__ __launch_bounds__(256, 4) void lowpass_down_kernel(const SRC *__restrict__ src, StripGeom g,
                                                              float *__restrict__ dst, int dpitch,
                                                              long long dst_frame_stride, Taps5 t,
                                                              float *__restrict__ dst2, int dpitch2,
                                                              long long dst2_frame_stride, Taps5 t5,
                                                              unsigned *__restrict__ zero_cnt)
{
  const ItemCoord it = decode_item(g);
  if (!it.valid) return;
  lowpass_down_item<SRC, MODE>(src, g, dst, dpitch, dst_frame_stride, t, dst2, dpitch2, dst2_frame_stride, t5, zero_cnt, it);
}

// ------------------------------------- LowPass + first ScaleDown, tiled (small batches, r04)
// The streamed kernel above is built for throughput: a wavefront walks a 240-pixel strip row by row, 8 rows of prologue
// before its first output row, ~19 dependent rows for an 8-row segment.  For ONE frame that is a latency chain at one
// wavefront per SIMD (17 us of a call's 83, r04 single-call budget).  Here a workgroup owns a 64 x 32 tile of the
// prefiltered image and does both separable passes — and the decimation — from LDS with every lane busy in parallel:
//   S  (TH+12) x (TW+12)  source region, clamp-to-edge          H  (TH+12) x (TW+4)   horizontal 9-tap
//   O  (TH+4)  x (TW+4)   vertical 9-tap = prefiltered tile + 2 px halo (an out-of-image entry holds the value of the
//                         clamped pixel, as the decimation's clamped reads see it); the inner TH x TW goes to `dst`
//   DH (TH+4)  x (TW/2)   horizontal 5-tap decimation            then the vertical 5-tap -> TH/2 x TW/2 pixels of `dst2`
// Same expressions on the same operands as lowpass_kernel + scaledown_kernel (conv9_expr; fmaf chains of hdec / vcomb):
// bit-identical.  Any width, any alignment, fp32 or 8-bit source (scalar loads); ~1.5x the arithmetic of the streamed
// kernel (halo), which is why batches keep that one.
// (stores: LPT_NT = 1 streams them out non-temporally while the kernel runs instead of leaving 10 MB of dirty lines for
//  the end-of-kernel L2 write-back, which is part of a dependent kernel's duration)
#ifndef LPT_NT
#define LPT_NT 1
#endif
#if LPT_NT
#define LPT_STORE1(p, v) __builtin_nontemporal_store((v), (p))
typedef float lpt_v2f __attribute__((ext_vector_type(2)));
#define LPT_STORE2(p, v) do { const float2 v2_ = (v); lpt_v2f w2_; w2_.x = v2_.x; w2_.y = v2_.y; __builtin_nontemporal_store(w2_, reinterpret_cast<lpt_v2f *>(p)); } while (0)
#else
#define LPT_STORE1(p, v) (*(p) = (v))
#define LPT_STORE2(p, v) (*reinterpret_cast<float2 *>(p) = (v))
#endif
#define LPT_TW 64
#ifndef LPT_TH
#define LPT_TH 32
#endif
#define LPT_SW (LPT_TW + 12)
#define LPT_SH (LPT_TH + 12)
#define LPT_HW (LPT_TW + 4)
#define LPT_OH (LPT_TH + 4)
template <typename SRC>
__global__ __launch_bounds__(256) void lowpass_down_tile_kernel(const SRC *__restrict__ src, int width, int height,
                                                                int spitch, long long src_frame_stride,
                                                                float *__restrict__ dst, int dpitch,
                                                                long long dst_frame_stride, Taps5 t,
                                                                float *__restrict__ dst2, int dpitch2,
                                                                long long dst2_frame_stride, Taps5 t5, int tiles_x,
                                                                unsigned *__restrict__ zero_cnt, int nframes, int dst_al8)
{
  __shared__ float s_S[LPT_SH * LPT_SW];        // later: DH
  __shared__ float s_H[LPT_SH * LPT_HW];
  __shared__ float s_O[LPT_OH * LPT_HW];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, frame = blockIdx.y;
  if (zero_cnt && blockIdx.x == 0 && tid < CNT_STRIDE) {      // the first kernel of the call clears the counters
    zero_cnt[(size_t)frame * CNT_STRIDE + tid] = 0u;
    if (frame == 0)
      for (int b = 0; b < CNT_SPARE_BLOCKS; b++) zero_cnt[(size_t)(nframes + b) * CNT_STRIDE + tid] = 0u;
  }
  const SRC *img = src + (long long)frame * src_frame_stride;
  float *out = dst + (long long)frame * dst_frame_stride;
  float *out2 = dst2 + (long long)frame * dst2_frame_stride;
  const int x0 = tx * LPT_TW, y0 = ty * LPT_TH;              // tile origin in the prefiltered image (both even)
  const float k0 = t.k[0], k1 = t.k[1], k2 = t.k[2], k3 = t.k[3], k4 = t.k[4];
  const float d0 = t5.k[0], d1 = t5.k[1], d2 = t5.k[2];
  // ---- S: source pixels (x0 - 6 + i, y0 - 6 + j), clamped
  // (all of a thread's loads are issued before its first LDS store: a rolled load -> store loop is one memory round trip
  //  per trip, and fourteen of them were most of this kernel's 15 us)
  {
    constexpr int TRIPS = (LPT_SH * LPT_SW + 255) / 256;
    float v[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; k++) {
      const int idx = min(tid + 256 * k, LPT_SH * LPT_SW - 1);
      const int j = idx / LPT_SW, i = idx - j * LPT_SW;
      v[k] = (float)img[(size_t)clampi(y0 - 6 + j, 0, height - 1) * spitch + clampi(x0 - 6 + i, 0, width - 1)];
    }
#pragma unroll
    for (int k = 0; k < TRIPS; k++)
      if (tid + 256 * k < LPT_SH * LPT_SW) s_S[tid + 256 * k] = v[k];
  }
  __syncthreads();
  // The two 9-tap passes are register-blocked (a thread reads a run of LDS quads once and produces 4 / 8 outputs from
  // them: a third of the LDS instructions of one-output-per-thread loops, which bound the first version of this kernel).
  // Both run the REGULAR stencil everywhere — exact wherever the output pixel lies inside the image, because S holds the
  // clamped source — and the entries of O outside the image, which must hold the value of the CLAMPED pixel for the
  // decimation, are copied from it afterwards (edge tiles only).
  // ---- H: rows of S; quad g = columns x0 - 2 + 4g .. + 3, from S columns 4g .. 4g + 11 of the row
  for (int idx = tid; idx < LPT_SH * (LPT_HW / 4); idx += 256) {
    const int j = idx / (LPT_HW / 4), g = idx - j * (LPT_HW / 4);
    const float4 *r = reinterpret_cast<const float4 *>(s_S + j * LPT_SW + 4 * g);
    const float4 a = r[0], b = r[1], c = r[2];                // S columns 4g .. 4g+11; output i = 4g + m is centred on 4g + 4 + m
    float4 h;
    h.x = conv9_expr(k0, k1, k2, k3, k4, b.x, b.y + a.w, b.z + a.z, b.w + a.y, c.x + a.x);
    h.y = conv9_expr(k0, k1, k2, k3, k4, b.y, b.z + b.x, b.w + a.w, c.x + a.z, c.y + a.y);
    h.z = conv9_expr(k0, k1, k2, k3, k4, b.z, b.w + b.y, c.x + b.x, c.y + a.w, c.z + a.z);
    h.w = conv9_expr(k0, k1, k2, k3, k4, b.w, c.x + b.z, c.y + b.y, c.z + b.x, c.w + a.w);
    *reinterpret_cast<float4 *>(s_H + j * LPT_HW + 4 * g) = h;
  }
  __syncthreads();
  // ---- O: rows y0 - 2 + j; a thread takes two rows of one quad column: H rows j .. j + 9 (local)
  for (int idx = tid; idx < (LPT_OH / 2) * (LPT_HW / 4); idx += 256) {
    const int jp = idx / (LPT_HW / 4), g = idx - jp * (LPT_HW / 4), j = 2 * jp;
    const float4 *c = reinterpret_cast<const float4 *>(s_H + j * LPT_HW + 4 * g);      // H row j <-> source row y0 - 6 + j
    float4 w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = c[k * (LPT_HW / 4)];
    float4 o0, o1;                                            // O rows j and j + 1: centred on H rows j + 4 and j + 5
    o0.x = conv9_expr(k0, k1, k2, k3, k4, w[4].x, w[5].x + w[3].x, w[6].x + w[2].x, w[7].x + w[1].x, w[8].x + w[0].x);
    o0.y = conv9_expr(k0, k1, k2, k3, k4, w[4].y, w[5].y + w[3].y, w[6].y + w[2].y, w[7].y + w[1].y, w[8].y + w[0].y);
    o0.z = conv9_expr(k0, k1, k2, k3, k4, w[4].z, w[5].z + w[3].z, w[6].z + w[2].z, w[7].z + w[1].z, w[8].z + w[0].z);
    o0.w = conv9_expr(k0, k1, k2, k3, k4, w[4].w, w[5].w + w[3].w, w[6].w + w[2].w, w[7].w + w[1].w, w[8].w + w[0].w);
    o1.x = conv9_expr(k0, k1, k2, k3, k4, w[5].x, w[6].x + w[4].x, w[7].x + w[3].x, w[8].x + w[2].x, w[9].x + w[1].x);
    o1.y = conv9_expr(k0, k1, k2, k3, k4, w[5].y, w[6].y + w[4].y, w[7].y + w[3].y, w[8].y + w[2].y, w[9].y + w[1].y);
    o1.z = conv9_expr(k0, k1, k2, k3, k4, w[5].z, w[6].z + w[4].z, w[7].z + w[3].z, w[8].z + w[2].z, w[9].z + w[1].z);
    o1.w = conv9_expr(k0, k1, k2, k3, k4, w[5].w, w[6].w + w[4].w, w[7].w + w[3].w, w[8].w + w[2].w, w[9].w + w[1].w);
    *reinterpret_cast<float4 *>(s_O + j * LPT_HW + 4 * g) = o0;
    *reinterpret_cast<float4 *>(s_O + (j + 1) * LPT_HW + 4 * g) = o1;
    // the tile's own pixels (local rows 2 .. TH + 1, columns 2 .. TW + 1) go to `dst`
    const int x = x0 - 2 + 4 * g;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const int jj = j + rr, y = y0 - 2 + jj;
      if (jj < 2 || jj >= LPT_TH + 2 || y >= height) continue;
      const float4 o = rr ? o1 : o0;
      float *orow = out + (size_t)y * dpitch;
      if (dst_al8 && g >= 1 && g < LPT_HW / 4 - 1 && x + 3 < width) {   // a whole quad inside the tile and the image: x % 4 == 2 (8-byte aligned rows)
        LPT_STORE2(orow + x, make_float2(o.x, o.y));
        LPT_STORE2(orow + x + 2, make_float2(o.z, o.w));
      } else {
        const float e[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const int i = 4 * g + m;
          if (i >= 2 && i < LPT_TW + 2 && x + m < width) LPT_STORE1(orow + x + m, e[m]);
        }
      }
    }
  }
  __syncthreads();
  // ---- edge tiles: the entries of O that lie outside the image take the value of the clamped pixel
  if (x0 < 2 || y0 < 2 || x0 + LPT_TW + 2 > width || y0 + LPT_TH + 2 > height) {
    for (int idx = tid; idx < LPT_OH * LPT_HW; idx += 256) {
      const int j = idx / LPT_HW, i = idx - j * LPT_HW;
      const int x = x0 - 2 + i, y = y0 - 2 + j;
      const int cx = clampi(x, 0, width - 1), cy = clampi(y, 0, height - 1);
      if (cx != x || cy != y) s_O[idx] = s_O[(cy - (y0 - 2)) * LPT_HW + (cx - (x0 - 2))];      // source entries are inside: never written here
    }
    __syncthreads();
  }
  // ---- DH: decimated column X (global tx * TW/2 + X) of every row of O: prefiltered columns 2Xg - 2 .. 2Xg + 2
  float *s_DH = s_S;
  for (int idx = tid; idx < LPT_OH * (LPT_TW / 2); idx += 256) {
    const int j = idx / (LPT_TW / 2), X = idx - j * (LPT_TW / 2);
    const float *r = s_O + j * LPT_HW + 2 * X;                // local column 2X <-> prefiltered column 2Xg - 2
    const float s = __builtin_fmaf(d0, r[0] + r[4], d1 * (r[1] + r[3]));
    s_DH[idx] = __builtin_fmaf(d2, r[2], s);
  }
  __syncthreads();
  const int w2 = width / 2, h2 = height / 2;
  for (int idx = tid; idx < (LPT_TH / 2) * (LPT_TW / 2); idx += 256) {
    const int Y = idx / (LPT_TW / 2), X = idx - Y * (LPT_TW / 2);
    const int Yg = ty * (LPT_TH / 2) + Y, Xg = tx * (LPT_TW / 2) + X;
    if (Yg >= h2 || Xg >= w2) continue;
    const float *c = s_DH + (2 * Y) * (LPT_TW / 2) + X;       // local row 2Y <-> prefiltered row 2Yg - 2
    float v = __builtin_fmaf(d2, c[2 * (LPT_TW / 2)], d0 * (c[0] + c[4 * (LPT_TW / 2)]));
    v = __builtin_fmaf(d1, c[LPT_TW / 2] + c[3 * (LPT_TW / 2)], v);
    LPT_STORE1(out2 + (size_t)Yg * dpitch2 + Xg, v);
  }
}

// ---------------------------------------------------------------- ScaleDown
// 5-tap Gaussian (variance 0.5) + 2x decimation: horizontal then vertical.
// Geometry `g` describes the SOURCE image; strips/segments tile the OUTPUT (w/2, h/2).
template <int FAST>      // 0 = generic, 1 = fast (width % 4 == 0), 2 = fast with a ragged last quad
__global__ __launch_bounds__(256) void scaledown_kernel(const float *__restrict__ src, StripGeom g,
                                                        float *__restrict__ dst, int dpitch,
                                                        long long dst_frame_stride, Taps5 t, int src_aligned,
                                                        int dst_aligned)
{
  const ItemCoord it = decode_item(g);
  if (!it.valid) return;
  const int lane = threadIdx.x & 63;
  const int q = it.strip * OUT_LANES + lane - 1;          // output quad
  const int w2 = g.width / 2, h2 = g.height / 2;
  const float *img = src + (long long)it.frame * g.frame_stride;
  float *out = dst + (long long)it.frame * dst_frame_stride;
  const int y0 = it.seg * g.seg_rows;                     // output rows
  const int y1 = min(y0 + g.seg_rows, h2);
  const float k0 = t.k[0], k1 = t.k[1], k2 = t.k[2];      // t.k[2] = centre tap here (reference order)
  const bool sal = src_aligned != 0, dal = dst_aligned != 0;

  const QuadCol qa = make_quadcol(2 * q, g.width), qb = make_quadcol(2 * q + 1, g.width);
  struct Raw2 { float4 A, B; };
  auto ldraw = [&](int y) -> Raw2 {
    const float *row = img + (size_t)clampi(y, 0, g.height - 1) * g.pitch;
    Raw2 r;
    r.A = load_quad_t<FAST>(row, 2 * q, g.width, sal, qa);       // input px 8q   .. 8q+3
    r.B = load_quad_t<FAST>(row, 2 * q + 1, g.width, sal, qb);   // input px 8q+4 .. 8q+7
    return r;
  };
  auto hfilt = [&](const Raw2 &rw) -> float4 {
    const float4 A = rw.A, B = rw.B;
    const float lz = lane_from_left(B.z), lw = lane_from_left(B.w);   // px 8q-2, 8q-1
    const float rx = lane_from_right(A.x);                             // px 8q+8
    float4 h;
    float s;
    s = __builtin_fmaf(k0, lz + A.z, k1 * (lw + A.y));  h.x = __builtin_fmaf(k2, A.x, s);
    s = __builtin_fmaf(k0, A.x + B.x, k1 * (A.y + A.w)); h.y = __builtin_fmaf(k2, A.z, s);
    s = __builtin_fmaf(k0, A.z + B.z, k1 * (A.w + B.y)); h.z = __builtin_fmaf(k2, B.x, s);
    s = __builtin_fmaf(k0, B.x + rx, k1 * (B.y + B.w)); h.w = __builtin_fmaf(k2, B.z, s);
    return h;
  };
  auto vcomb = [&](float a0, float a1, float a2, float a3, float a4) -> float {
    float s = __builtin_fmaf(k2, a2, k0 * (a0 + a4));
    s = __builtin_fmaf(k1, a1 + a3, s);
    return s;
  };

  auto hrow = [&](int y) -> float4 { return hfilt(ldraw(y)); };
  const bool writer = lane >= 1 && lane <= OUT_LANES && 4 * q < w2;
  const bool fast_store = FAST && (w2 & 3) == 0 && dal;
  float4 t0 = hrow(2 * y0 - 2), t1 = hrow(2 * y0 - 1), t2 = hrow(2 * y0), t3, t4;
  Raw2 ra = ldraw(2 * y0 + 1), rb = ldraw(2 * y0 + 2), rc = ldraw(2 * y0 + 3), rd = ldraw(2 * y0 + 4);
  for (int y = y0; y < y1; y++) {
    const Raw2 na = ldraw(2 * y + 5), nb = ldraw(2 * y + 6);    // prefetch two output rows (four input rows) ahead
    t3 = hfilt(ra);
    t4 = hfilt(rb);
    ra = rc; rb = rd; rc = na; rd = nb;
    float4 o;
    o.x = vcomb(t0.x, t1.x, t2.x, t3.x, t4.x);
    o.y = vcomb(t0.y, t1.y, t2.y, t3.y, t4.y);
    o.z = vcomb(t0.z, t1.z, t2.z, t3.z, t4.z);
    o.w = vcomb(t0.w, t1.w, t2.w, t3.w, t4.w);
    if (writer) {
      if (fast_store) *reinterpret_cast<float4 *>(out + (size_t)y * dpitch + 4 * q) = o;
      else store_quad(out + (size_t)y * dpitch, q, w2, dal, o);
    }
    t0 = t2; t1 = t3; t2 = t4;
  }
}

// ------------------------------------------------- ScaleDown chain (small batches): see chain.hpp
__global__ __launch_bounds__(1024) void scaledown_chain_kernel(float *__restrict__ scratch, ChainGeom G, Taps5 t)
{
  __shared__ float s_lds[CHAIN_LDS_FLOATS_MAX];
  scaledown_chain_block(scratch, G, t, (int)blockIdx.x, (int)blockIdx.y, s_lds);
}

// ------------------------------------------------------------------ ScaleUp
template <typename SRC>
__global__ __launch_bounds__(256) void scaleup_kernel(const SRC *__restrict__ src0, int w, int h, int spitch,
                                                      long long src_frame_stride, float *__restrict__ dst0, int dpitch,
                                                      long long dst_frame_stride)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const SRC *src = src0 + (long long)blockIdx.z * src_frame_stride;          // one grid layer per frame of the batch
  float *dst = dst0 + (long long)blockIdx.z * dst_frame_stride;
  const int xr = min(x + 1, w - 1), yd = min(y + 1, h - 1);
  const float vul = (float)src[(size_t)y * spitch + x], vur = (float)src[(size_t)y * spitch + xr];
  const float vdl = (float)src[(size_t)yd * spitch + x], vdr = (float)src[(size_t)yd * spitch + xr];
  float2 top = make_float2(vul, 0.50f * (vul + vur));
  float2 bot = make_float2(0.50f * (vul + vdl), 0.25f * (vul + vur + vdl + vdr));
  *reinterpret_cast<float2 *>(dst + (size_t)(2 * y) * dpitch + 2 * x) = top;
  *reinterpret_cast<float2 *>(dst + (size_t)(2 * y + 1) * dpitch + 2 * x) = bot;
}

// ------------------------------------------------------------- host wrappers
static inline bool is_aligned16(const void *p, int pitch) { return (((uintptr_t)p) & 15) == 0 && (pitch & 3) == 0; }

static inline dim3 grid_for(const StripGeom &g)
{
  const long long nitems = (long long)g.nframes * g.nstrips * g.nsegs;
  return dim3((unsigned)((nitems + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
}

int launch_lowpass(misift_ctx *ctx, const void *src, int src_u8, const StripGeom &g, float *dst, int dpitch,
                   long long dst_frame_stride, const float k9[9], unsigned *zero_cnt)
{
  Taps5 t;
  for (int j = 0; j <= 4; j++) t.k[j] = k9[4 - j];    // centre first, then outward
  const int dal = is_aligned16(dst, dpitch) && (dst_frame_stride & 3) == 0;
  LaunchScope ls(ctx, "lowpass");
  if (src_u8) {
    // one dword (4 pixels) per lane: rows must be 4-byte aligned for the fast path
    const unsigned char *s8 = static_cast<const unsigned char *>(src);
    const int sal = (((uintptr_t)s8) & 3) == 0 && (g.pitch & 3) == 0 && (g.frame_stride & 3) == 0;
    if (sal && dal && (g.width & 3) == 0)
      hipLaunchKernelGGL((lowpass_kernel<true, unsigned char>), grid_for(g), dim3(256), 0, ctx->stream, s8, g, dst,
                         dpitch, dst_frame_stride, t, sal, dal, zero_cnt);
    else
      hipLaunchKernelGGL((lowpass_kernel<false, unsigned char>), grid_for(g), dim3(256), 0, ctx->stream, s8, g, dst,
                         dpitch, dst_frame_stride, t, sal, dal, zero_cnt);
    return ls.finish();
  }
  const float *sf = static_cast<const float *>(src);
  const int sal = is_aligned16(sf, g.pitch) && (g.frame_stride & 3) == 0;
  if (sal && dal && (g.width & 3) == 0)
    hipLaunchKernelGGL((lowpass_kernel<true, float>), grid_for(g), dim3(256), 0, ctx->stream, sf, g, dst, dpitch,
                       dst_frame_stride, t, sal, dal, zero_cnt);
  else
    hipLaunchKernelGGL((lowpass_kernel<false, float>), grid_for(g), dim3(256), 0, ctx->stream, sf, g, dst, dpitch,
                       dst_frame_stride, t, sal, dal, zero_cnt);
  return ls.finish();
}

// Fused prefilter + first decimation; returns MISIFT_OK with *done = 0 when the shape does not qualify
// (the caller then runs launch_lowpass + launch_scaledown).
int launch_lowpass_down(misift_ctx *ctx, const void *src, int src_u8, const StripGeom &g, float *dst, int dpitch,
                        long long dst_frame_stride, const float k9[9], float *dst2, int dpitch2,
                        long long dst2_frame_stride, const float k5[5], int *done, unsigned *zero_cnt)
{
  *done = 0;
  if (g.height < 8 || (g.seg_rows & 1)) return MISIFT_OK;      // (segments start on even rows: the decimation's phase)
  const int dal = is_aligned16(dst, dpitch) && (dst_frame_stride & 3) == 0;
  const int d2al = (((uintptr_t)dst2) & 7) == 0 && (dpitch2 & 1) == 0 && (dst2_frame_stride & 1) == 0;
  int sal;
  if (src_u8) sal = (((uintptr_t)src) & 3) == 0 && (g.pitch & 3) == 0 && (g.frame_stride & 3) == 0;
  else sal = is_aligned16(src, g.pitch) && (g.frame_stride & 3) == 0;
  if (!(sal && dal && d2al)) return MISIFT_OK;
  Taps5 t, t5;
  for (int j = 0; j <= 4; j++) t.k[j] = k9[4 - j];
  for (int j = 0; j < 5; j++) t5.k[j] = k5[j];
  LaunchScope ls(ctx, "lowpass_down");
  // (r05 tried this kernel with a capped grid — 1-3 workgroups per CU, each walking over several items, so that the
  //  HBM-bound prefilter leaves wave slots to the other batch's VALU-bound kernels: no gain, profiles/r05_lpd_persist_ab.txt)
  // widths that are not a multiple of 4 (r03): the same kernel with the ragged-quad loads; the aligned rows checked
  // above make the row pitch a multiple of 4, so the partial quad's dwordx4 stays inside the row
  const bool rag = (g.width & 3) != 0;
#define LPD_LAUNCH(T, M, P) hipLaunchKernelGGL((lowpass_down_kernel<T, M>), grid_for(g), dim3(256), (size_t)ctx->lds_pad_lpd, ctx->stream, \
                                               P, g, dst, dpitch, dst_frame_stride, t, dst2, dpitch2, dst2_frame_stride, t5, zero_cnt)
  if (src_u8) {
    const unsigned char *s8 = static_cast<const unsigned char *>(src);
    if (rag) LPD_LAUNCH(unsigned char, 2, s8); else LPD_LAUNCH(unsigned char, 1, s8);
  } else {
    const float *sf = static_cast<const float *>(src);
    if (rag) LPD_LAUNCH(float, 2, sf); else LPD_LAUNCH(float, 1, sf);
  }
#undef LPD_LAUNCH
  *done = 1;
  return ls.finish();
}

// the tiled form for small batches: any shape (no alignment or width conditions), always done
int launch_lowpass_down_tile(misift_ctx *ctx, const void *src, int src_u8, int width, int height, int spitch,
                             long long src_frame_stride, int nframes, float *dst, int dpitch, long long dst_frame_stride,
                             const float k9[9], float *dst2, int dpitch2, long long dst2_frame_stride, const float k5[5],
                             unsigned *zero_cnt)
{
  Taps5 t, t5;
  for (int j = 0; j <= 4; j++) t.k[j] = k9[4 - j];
  for (int j = 0; j < 5; j++) t5.k[j] = k5[j];
  const int tiles_x = (width + LPT_TW - 1) / LPT_TW, tiles_y = (height + LPT_TH - 1) / LPT_TH;
  const dim3 grid(tiles_x * tiles_y, nframes);
  // the pair stores need 8-byte aligned rows (every arena of ours is; a caller's odd pointer takes the scalar stores)
  const int dst_al8 = (((uintptr_t)dst) & 7) == 0 && (dpitch & 1) == 0 && (dst_frame_stride & 1) == 0;
  LaunchScope ls(ctx, "lowpass_down");
  if (src_u8)
    hipLaunchKernelGGL(lowpass_down_tile_kernel<unsigned char>, grid, dim3(256), 0, ctx->stream,
                       static_cast<const unsigned char *>(src), width, height, spitch, src_frame_stride, dst, dpitch,
                       dst_frame_stride, t, dst2, dpitch2, dst2_frame_stride, t5, tiles_x, zero_cnt, nframes, dst_al8);
  else
    hipLaunchKernelGGL(lowpass_down_tile_kernel<float>, grid, dim3(256), 0, ctx->stream, static_cast<const float *>(src),
                       width, height, spitch, src_frame_stride, dst, dpitch, dst_frame_stride, t, dst2, dpitch2,
                       dst2_frame_stride, t5, tiles_x, zero_cnt, nframes, dst_al8);
  return ls.finish();
}

int launch_scaledown(misift_ctx *ctx, const float *src, const StripGeom &g, float *dst, int dpitch,
                     long long dst_frame_stride, const float k5[5])
{
  Taps5 t;
  for (int j = 0; j < 5; j++) t.k[j] = k5[j];
  const int sal = is_aligned16(src, g.pitch) && (g.frame_stride & 3) == 0;
  const int dal = is_aligned16(dst, dpitch) && (dst_frame_stride & 3) == 0;
  LaunchScope ls(ctx, "scaledown");
  if (sal && (g.width & 3) == 0)
    hipLaunchKernelGGL(scaledown_kernel<1>, grid_for(g), dim3(256), 0, ctx->stream, src, g, dst, dpitch,
                       dst_frame_stride, t, sal, dal);
  else if (sal)                                   // ragged width: still one dwordx4 per quad (r03)
    hipLaunchKernelGGL(scaledown_kernel<2>, grid_for(g), dim3(256), 0, ctx->stream, src, g, dst, dpitch,
                       dst_frame_stride, t, sal, dal);
  else
    hipLaunchKernelGGL(scaledown_kernel<0>, grid_for(g), dim3(256), 0, ctx->stream, src, g, dst, dpitch,
                       dst_frame_stride, t, sal, dal);
  return ls.finish();
}

// ScaleDowns of pyramid levels src -> dst[0] -> dst[1] -> ... (nlev <= CHAIN_MAX_LEVELS) in one launch; levels are
// given as {w, h, pitch, float offset inside a frame's arena} with lv[0] the source.
int make_chain_geom(ChainGeom *G, long long frame_stride, const int (*dims)[3], const long long *offs, int nlev, int tile)
{
  if (nlev < 1 || nlev > CHAIN_MAX_LEVELS) {
    misift_set_error("scaledown chain: %d levels", nlev);
    return MISIFT_EINVAL;
  }
  memset(G, 0, sizeof(*G));
  G->K = nlev;
  G->T = tile << (CHAIN_MAX_LEVELS - nlev);            // T << K = 8 * tile whatever K (tile 8: source region <= 85 x 85)
  G->frame_stride = frame_stride;
  for (int k = 0; k <= nlev; k++) { G->lv[k].w = dims[k][0]; G->lv[k].h = dims[k][1]; G->lv[k].p = dims[k][2]; G->lv[k].off = offs[k]; }
  G->tiles_x = (G->lv[nlev].w + G->T - 1) / G->T;
  G->tiles_y = (G->lv[nlev].h + G->T - 1) / G->T;
  return MISIFT_OK;
}

int launch_scaledown_chain(misift_ctx *ctx, float *scratch, long long frame_stride, int nframes, const int (*dims)[3],
                           const long long *offs, int nlev, const float k5[5])
{
  ChainGeom G;
  const int rc = make_chain_geom(&G, frame_stride, dims, offs, nlev, 8);
  if (rc) return rc;
  Taps5 t;
  for (int j = 0; j < 5; j++) t.k[j] = k5[j];
  LaunchScope ls(ctx, "scaledown_chain");
  hipLaunchKernelGGL(scaledown_chain_kernel, dim3(G.tiles_x * G.tiles_y, nframes), dim3(1024), 0, ctx->stream, scratch, G, t);
  return ls.finish();
}

int launch_scaleup(misift_ctx *ctx, const void *src, int src_u8, int w, int h, int spitch, long long src_frame_stride,
                   int nframes, float *dst, int dpitch, long long dst_frame_stride)
{
  LaunchScope ls(ctx, "scaleup");
  const dim3 grid((w + 63) / 64, (h + 3) / 4, nframes);
  if (src_u8)
    hipLaunchKernelGGL(scaleup_kernel<unsigned char>, grid, dim3(256), 0, ctx->stream,
                       static_cast<const unsigned char *>(src), w, h, spitch, src_frame_stride, dst, dpitch,
                       dst_frame_stride);
  else
    hipLaunchKernelGGL(scaleup_kernel<float>, grid, dim3(256), 0, ctx->stream, static_cast<const float *>(src), w, h,
                       spitch, src_frame_stride, dst, dpitch, dst_frame_stride);
  return ls.finish();
}

// kernels_dog.hip — difference-of-Gaussians and scale-space extrema for gfx950.
//
//   laplace_kernel        replaces LaplaceMultiMem   (reference cudaSiftD.cu:1753-1793, host cudaSiftH.cu:460-487)
//   detect_kernel         replaces the 3x3x3 extremum search of FindPointsMultiNew (cudaSiftD.cu:1292-1366)
//   dog_scan_kernel       = laplace + a cheap NECESSARY extremum test fused: DoG values exist only in registers,
//   dog_scan_all_kernel     the < 0.1 % survivors go to a candidate list (..._all: every octave of every frame in
//                           one launch — the default path)
//   refine_kernel         replaces the edge test / sub-pixel refinement / append of FindPointsMultiNew
//   refine_all_kernel       (cudaSiftD.cu:1379-1430); after a scan it recomputes the 3x3x3 DoG neighbourhood of
//                           each candidate from the octave base image with the same fmaf chains (bit-identical
//                           values) and applies the reference's full 26-neighbour test first (..._all: 16 lanes
//                           per candidate, DPP row shifts)
//
// Streaming design (see common.hpp): a wavefront walks down a 256-px strip; the 9-row raw window lives in VGPRs
// (rotating names in a 3x unrolled loop), horizontal neighbours come from adjacent lanes by DPP, blurs are
// packed v_pk_fma_f32 over PAIRS OF SCALES with the tap pairs read from LDS.  Unfused laplace is HBM-write-bound
// (4 B read + 28 B written per px); unfused detect is HBM-read-bound (28 B/px); the scan reads ~5 B/px and is
// fp32-VALU-bound (DESIGN.md section 4 has the measurements behind each of these choices).
//
// Compiled with -ffp-contract=off: every multiply-add below that is meant to be
// fused is an explicit __builtin_fmaf, exactly as in oracle/sift_oracle.c.
#include <stdlib.h>
#include <string.h>
#include <math.h>
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

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Vertical 9-tap of one blur scale on a quad column, then the horizontal 9-tap
// using the neighbouring lanes' vertical results.
__device__ __forceinline__ float4 blur_quad(const float *k, float4 c, float4 p1, float4 p2, float4 p3, float4 p4)
{
  const float k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3], k4 = k[4];
  float4 v;
  v.x = conv9(k0, k1, k2, k3, k4, c.x, p1.x, p2.x, p3.x, p4.x);
  v.y = conv9(k0, k1, k2, k3, k4, c.y, p1.y, p2.y, p3.y, p4.y);
  v.z = conv9(k0, k1, k2, k3, k4, c.z, p1.z, p2.z, p3.z, p4.z);
  v.w = conv9(k0, k1, k2, k3, k4, c.w, p1.w, p2.w, p3.w, p4.w);
  const float4 l = quad_from_left(v);
  const float4 r = quad_from_right(v);
  float4 h;
  h.x = conv9(k0, k1, k2, k3, k4, v.x, l.w + v.y, l.z + v.z, l.y + v.w, l.x + r.x);
  h.y = conv9(k0, k1, k2, k3, k4, v.y, v.x + v.z, l.w + v.w, l.z + r.x, l.y + r.y);
  h.z = conv9(k0, k1, k2, k3, k4, v.z, v.y + v.w, v.x + r.x, l.w + r.y, l.z + r.z);
  h.w = conv9(k0, k1, k2, k3, k4, v.w, v.z + r.x, v.y + r.y, v.x + r.z, l.w + r.w);
  return h;
}

// Packed-math variant of blur_quad with the taps read from LDS as {k,k} pairs (one ds_read_b64 feeds
// v_pk_fma_f32 operands).  Keeping the 40 taps out of the scalar register file matters for
// dog_scan_kernel: with the taps as SGPR pairs (80 SGPRs) the detection code runs out of scalar
// registers and the compiler spills them through v_readlane/v_writelane inside the row loop.  Same
// fmaf chains (__builtin_elementwise_fma is llvm.fma — one rounding), hence the same bits as blur_quad().
typedef float v2f __attribute__((ext_vector_type(2)));
struct Quad2 { v2f lo, hi; };                 // pixels (x,y) and (z,w) of a quad

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }

__device__ __forceinline__ void fill_lds_taps(v2f *s_taps, const LaplaceTaps &taps)
{
  for (int i = threadIdx.x; i < NUM_BLURS * 5; i += blockDim.x) {
    const float k = taps.k[i / 5][i % 5];
    s_taps[i] = mk2(k, k);
  }
  __syncthreads();
}

struct Taps2 { v2f k0, k1, k2, k3, k4; };
__device__ __forceinline__ Taps2 load_taps2(const v2f *tk)
{
  Taps2 t;
  t.k0 = tk[0]; t.k1 = tk[1]; t.k2 = tk[2]; t.k3 = tk[3]; t.k4 = tk[4];
  return t;
}
__device__ __forceinline__ Quad2 blur_quad2(const Taps2 &t, Quad2 c, Quad2 p1, Quad2 p2, Quad2 p3, Quad2 p4)
{
  const v2f k0 = t.k0, k1 = t.k1, k2 = t.k2, k3 = t.k3, k4 = t.k4;
  v2f a = k0 * c.lo, b = k0 * c.hi;           // vertical pass: (x,y) and (z,w)
  a = pk_fma(k1, p1.lo, a); b = pk_fma(k1, p1.hi, b);
  a = pk_fma(k2, p2.lo, a); b = pk_fma(k2, p2.hi, b);
  a = pk_fma(k3, p3.lo, a); b = pk_fma(k3, p3.hi, b);
  a = pk_fma(k4, p4.lo, a); b = pk_fma(k4, p4.hi, b);
  const float vx = a.x, vy = a.y, vz = b.x, vw = b.y;
  const float lx = lane_from_left(vx), ly = lane_from_left(vy), lz = lane_from_left(vz), lw = lane_from_left(vw);
  const float rx = lane_from_right(vx), ry = lane_from_right(vy), rz = lane_from_right(vz), rw = lane_from_right(vw);
  // horizontal pass; pair sums (value at -j) + (value at +j) for the four pixels
  v2f h0 = k0 * a, h1 = k0 * b;
  h0 = pk_fma(k1, mk2(lw, vx) + mk2(vy, vz), h0); h1 = pk_fma(k1, mk2(vy, vz) + mk2(vw, rx), h1);
  h0 = pk_fma(k2, mk2(lz, lw) + mk2(vz, vw), h0); h1 = pk_fma(k2, mk2(vx, vy) + mk2(rx, ry), h1);
  h0 = pk_fma(k3, mk2(ly, lz) + mk2(vw, rx), h0); h1 = pk_fma(k3, mk2(lw, vx) + mk2(ry, rz), h1);
  h0 = pk_fma(k4, mk2(lx, ly) + mk2(rx, ry), h0); h1 = pk_fma(k4, mk2(lz, lw) + mk2(rz, rw), h1);
  Quad2 o;
  o.lo = h0; o.hi = h1;
  return o;
}
__device__ __forceinline__ Quad2 q2(float4 v) { Quad2 r; r.lo = mk2(v.x, v.y); r.hi = mk2(v.z, v.w); return r; }
__device__ __forceinline__ Quad2 add_q2(Quad2 a, Quad2 b) { Quad2 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi; return r; }
__device__ __forceinline__ float4 add4p(float4 a, float4 b)      // two v_pk_add_f32
{
  const v2f l = mk2(a.x, a.y) + mk2(b.x, b.y), h = mk2(a.z, a.w) + mk2(b.z, b.w);
  return make_float4(l.x, l.y, h.x, h.y);
}
__device__ __forceinline__ float4 sub_q2(Quad2 a, Quad2 b)
{
  const v2f l = a.lo - b.lo, h = a.hi - b.hi;
  return make_float4(l.x, l.y, h.x, h.y);
}

// Scale-paired variant for the fused scan: the two halves of every packed operand are the SAME pixel at two
// consecutive blur scales (taps {k_s[j], k_s+1[j]}), not two pixels of one scale.  Every operand of the
// horizontal pass is then a register pair as it stands — the pixel-paired form needs pairs such as (y,z) or
// (w, right.x) that straddle the quad's pairs and cost ~50 v_mov/v_pk_mov per row to assemble.  Same fmaf
// chain per value as blur_quad().
struct Pair4 { v2f x, y, z, w; };
__device__ __forceinline__ v2f lane_from_left2(v2f v) { return mk2(lane_from_left(v.x), lane_from_left(v.y)); }
__device__ __forceinline__ v2f lane_from_right2(v2f v) { return mk2(lane_from_right(v.x), lane_from_right(v.y)); }
__device__ __forceinline__ v2f conv9p(const Taps2 &t, v2f c, v2f p1, v2f p2, v2f p3, v2f p4)
{
  v2f s = t.k0 * c;
  s = pk_fma(t.k1, p1, s);
  s = pk_fma(t.k2, p2, s);
  s = pk_fma(t.k3, p3, s);
  s = pk_fma(t.k4, p4, s);
  return s;
}
__device__ __forceinline__ Pair4 blur_pair(const Taps2 &t, float4 c, float4 p1, float4 p2, float4 p3, float4 p4)
{
  Pair4 v;
  v.x = conv9p(t, mk2(c.x, c.x), mk2(p1.x, p1.x), mk2(p2.x, p2.x), mk2(p3.x, p3.x), mk2(p4.x, p4.x));
  v.y = conv9p(t, mk2(c.y, c.y), mk2(p1.y, p1.y), mk2(p2.y, p2.y), mk2(p3.y, p3.y), mk2(p4.y, p4.y));
  v.z = conv9p(t, mk2(c.z, c.z), mk2(p1.z, p1.z), mk2(p2.z, p2.z), mk2(p3.z, p3.z), mk2(p4.z, p4.z));
  v.w = conv9p(t, mk2(c.w, c.w), mk2(p1.w, p1.w), mk2(p2.w, p2.w), mk2(p3.w, p3.w), mk2(p4.w, p4.w));
  const v2f lx = lane_from_left2(v.x), ly = lane_from_left2(v.y), lz = lane_from_left2(v.z), lw = lane_from_left2(v.w);
  const v2f rx = lane_from_right2(v.x), ry = lane_from_right2(v.y), rz = lane_from_right2(v.z),
            rw = lane_from_right2(v.w);
  Pair4 h;
  h.x = conv9p(t, v.x, lw + v.y, lz + v.z, ly + v.w, lx + rx);
  h.y = conv9p(t, v.y, v.x + v.z, lw + v.w, lz + rx, ly + ry);
  h.z = conv9p(t, v.z, v.y + v.w, v.x + rx, lw + ry, lz + rz);
  h.w = conv9p(t, v.w, v.z + rx, v.y + ry, v.x + rz, lw + rw);
  return h;
}
// ---- the same blur_pair in two halves with the neighbour lanes' vertical sums travelling through LDS (r06, SCAN_XCH).
// The 16 DPP moves per scale pair cost the SIMD as much as 16 packed FMAs (4.7 issue cycles each: 21 % of the row);
// two ds_write_b128 + four ds_read_b128 take no VALU issue slot.  Layout per wavefront: two planes of 66 float4 slots
// (plane 0: the (x, y) pixels' scale pairs, plane 1: (z, w)); lane l owns slot l + 1, slots 0 and 65 stay zero (what
// DPP's bound_ctrl hands lanes 0 and 63).  A wavefront's DS instructions execute in issue order, so the ONE buffer serves
// the three scale pairs of a row back to back — write pair p+1 behind the reads of pair p — with no wait in between;
// the wavefront-scope fences only keep the compiler from reordering accesses that alias across lanes.
#define XCH_PLANE 66
#define XCH_FLOAT4S (2 * XCH_PLANE)
__device__ __forceinline__ void wave_lds_order()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ Pair4 vert_pair(const Taps2 &t, float4 c, float4 p1, float4 p2, float4 p3, float4 p4)
{
  Pair4 v;
  v.x = conv9p(t, mk2(c.x, c.x), mk2(p1.x, p1.x), mk2(p2.x, p2.x), mk2(p3.x, p3.x), mk2(p4.x, p4.x));
  v.y = conv9p(t, mk2(c.y, c.y), mk2(p1.y, p1.y), mk2(p2.y, p2.y), mk2(p3.y, p3.y), mk2(p4.y, p4.y));
  v.z = conv9p(t, mk2(c.z, c.z), mk2(p1.z, p1.z), mk2(p2.z, p2.z), mk2(p3.z, p3.z), mk2(p4.z, p4.z));
  v.w = conv9p(t, mk2(c.w, c.w), mk2(p1.w, p1.w), mk2(p2.w, p2.w), mk2(p3.w, p3.w), mk2(p4.w, p4.w));
  return v;
}
struct Nbr4 { v2f lx, ly, lz, lw, rx, ry, rz, rw; };
__device__ __forceinline__ void xch_put(float4 *xch, const Pair4 &v)       // xch = the wavefront's buffer + lane
{
  wave_lds_order();
  xch[1] = make_float4(v.x.x, v.x.y, v.y.x, v.y.y);
  xch[XCH_PLANE + 1] = make_float4(v.z.x, v.z.y, v.w.x, v.w.y);
  wave_lds_order();
}
__device__ __forceinline__ Nbr4 xch_get(const float4 *xch)
{
  const float4 la = xch[0], lb = xch[XCH_PLANE], ra = xch[2], rb = xch[XCH_PLANE + 2];
  Nbr4 n;
  n.lx = mk2(la.x, la.y); n.ly = mk2(la.z, la.w); n.lz = mk2(lb.x, lb.y); n.lw = mk2(lb.z, lb.w);
  n.rx = mk2(ra.x, ra.y); n.ry = mk2(ra.z, ra.w); n.rz = mk2(rb.x, rb.y); n.rw = mk2(rb.z, rb.w);
  return n;
}
__device__ __forceinline__ Nbr4 dpp_get(const Pair4 &v)
{
  Nbr4 n;
  n.lx = lane_from_left2(v.x); n.ly = lane_from_left2(v.y); n.lz = lane_from_left2(v.z); n.lw = lane_from_left2(v.w);
  n.rx = lane_from_right2(v.x); n.ry = lane_from_right2(v.y); n.rz = lane_from_right2(v.z); n.rw = lane_from_right2(v.w);
  return n;
}
__device__ __forceinline__ Pair4 horiz_pair(const Taps2 &t, const Pair4 &v, const Nbr4 &n)
{
  Pair4 h;
  h.x = conv9p(t, v.x, n.lw + v.y, n.lz + v.z, n.ly + v.w, n.lx + n.rx);
  h.y = conv9p(t, v.y, v.x + v.z, n.lw + v.w, n.lz + n.rx, n.ly + n.ry);
  h.z = conv9p(t, v.z, v.y + v.w, v.x + n.rx, n.lw + n.ry, n.lz + n.rz);
  h.w = conv9p(t, v.w, v.z + n.rx, v.y + n.ry, v.x + n.rz, n.lw + n.rw);
  return h;
}
// tap pairs of the three scale pairs (1,2), (3,4), (5,6) the scan needs: dst[5*p + j] = {k[1+2p][j], k[2+2p][j]}
#define NUM_SCAN_PAIRS 3
__device__ __forceinline__ v2f scan_pair_tap(const LaplaceTaps &taps, int i)
{
  const int p = i / 5, j = i - 5 * p;
  return mk2(taps.k[1 + 2 * p][j], taps.k[2 + 2 * p][j]);
}

// ------------------------------------------------------------------ Laplace
template <bool FAST>
__global__ __launch_bounds__(256) void laplace_kernel(const float *__restrict__ base, StripGeom g,
                                                      float *__restrict__ dog, long long dog_frame_stride,
                                                      LaplaceTaps taps, int aligned)
{
  __shared__ v2f s_taps[NUM_BLURS * 5];
  fill_lds_taps(s_taps, taps);
  const v2f *tk = s_taps;
  const ItemCoord it = decode_item(g);
  if (!it.valid) return;
  const int lane = threadIdx.x & 63;
  const int q = it.strip * OUT_LANES + lane - 1;
  const float *img = base + (long long)it.frame * g.frame_stride;
  float *out = dog + (long long)it.frame * dog_frame_stride;
  const size_t plane = (size_t)g.height * g.pitch;
  const int y0 = it.seg * g.seg_rows;
  const int y1 = min(y0 + g.seg_rows, g.height);
  const bool al = aligned != 0;
  const QuadCol qc = make_quadcol(q, g.width);
  auto ld = [&](int y) -> float4 {
    return load_quad_t<FAST>(img + (size_t)clampi(y, 0, g.height - 1) * g.pitch, q, g.width, al, qc);
  };
  float4 r0 = ld(y0 - 4), r1 = ld(y0 - 3), r2 = ld(y0 - 2), r3 = ld(y0 - 1), r4 = ld(y0);
  float4 r5 = ld(y0 + 1), r6 = ld(y0 + 2), r7 = ld(y0 + 3), r8 = ld(y0 + 4);
  const bool writer = lane >= 1 && lane <= OUT_LANES && 4 * q < g.width;
  for (int y = y0; y < y1; y++) {
    const float4 rnext = ld(y + 5);              // prefetch: latency hides under this row's math
    const Quad2 c = q2(r4), p1 = add_q2(q2(r3), q2(r5)), p2 = add_q2(q2(r2), q2(r6)), p3 = add_q2(q2(r1), q2(r7)),
                p4 = add_q2(q2(r0), q2(r8));
    // tap pairs of scale s+1 are fetched from LDS while scale s is computed
    Taps2 tcur = load_taps2(tk), tnext = load_taps2(tk + 5);
    __builtin_amdgcn_sched_barrier(0);
    Quad2 old = blur_quad2(tcur, c, p1, p2, p3, p4);
#pragma unroll
    for (int s = 1; s < NUM_BLURS; s++) {
      tcur = tnext;
      if (s + 1 < NUM_BLURS) tnext = load_taps2(tk + 5 * (s + 1));
      __builtin_amdgcn_sched_barrier(0);
      const Quad2 res = blur_quad2(tcur, c, p1, p2, p3, p4);
      float *dst = out + (size_t)(s - 1) * plane + (size_t)y * g.pitch;
      if (writer) {
        if (FAST) *reinterpret_cast<float4 *>(dst + 4 * q) = sub_q2(res, old);
        else store_quad(dst, q, g.width, al, sub_q2(res, old));
      }
      old = res;
    }
    r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6; r6 = r7; r7 = r8; r8 = rnext;
  }
}

// ------------------------------------------------------------------- detect
// One row of one DoG plane as seen by a lane: its quad plus the pixel left of it
// and the pixel right of it.
struct Row6 { float l, x, y, z, w, r; };

__device__ __forceinline__ Row6 make_row6(float4 c)
{
  Row6 o;
  o.x = c.x; o.y = c.y; o.z = c.z; o.w = c.w;
  o.l = lane_from_left(c.w);
  o.r = lane_from_right(c.x);
  return o;
}
// Fused path: the DoG quad was COMPUTED (not loaded), so the clamp-to-edge addressing of the
// reference's neighbour reads (cudaSiftD.cu:1308, xpos clamped) must be applied in DoG space:
// pixels right of the image take DoG(width-1), the pixel left of column 0 takes DoG(0).
__device__ __forceinline__ Row6 make_row6_edge(float4 c, int q, int width)
{
  const int x = 4 * q;
  if (x < width && x + 3 >= width) {           // this quad holds the last image column
    const int last = width - 1 - x;            // 0..3
    const float e = last == 0 ? c.x : (last == 1 ? c.y : (last == 2 ? c.z : c.w));
    if (last < 1) c.y = e;
    if (last < 2) c.z = e;
    if (last < 3) c.w = e;
  }
  Row6 o;
  o.x = c.x; o.y = c.y; o.z = c.z; o.w = c.w;
  const float fl = lane_from_left(c.w), fr = lane_from_right(c.x);   // all lanes execute the DPP
  o.l = (x - 1 < 0) ? c.x : fl;
  o.r = (x + 4 >= width) ? c.w : fr;
  return o;
}
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

struct Col6 { float v[6]; };

// 3x3x3 strict-extremum test of the centre row `b` (rows a, b, c = y-1, y, y+1) of
// all 7 planes; returns a 20-bit mask, bit (5*i + s) set when pixel i of the quad is a
// candidate at DoG scale s (centre plane s+1).  Equivalent to the 26-neighbour test of
// the reference (cudaSiftD.cu:1337-1360): box minima of the planes below/above, ring
// minimum of the centre plane.
__device__ __forceinline__ unsigned extrema_mask(const Row6 (&a)[NUM_DOG], const Row6 (&b)[NUM_DOG],
                                                 const Row6 (&c)[NUM_DOG], float thresh)
{
  float bmin[NUM_DOG][4], bmax[NUM_DOG][4];     // 3x3 box min/max per plane and pixel
  float cmin[NUM_DOG][6], cmax[NUM_DOG][6];     // 3-row column min/max
#pragma unroll
  for (int p = 0; p < NUM_DOG; p++) {
    const float av[6] = {a[p].l, a[p].x, a[p].y, a[p].z, a[p].w, a[p].r};
    const float bv[6] = {b[p].l, b[p].x, b[p].y, b[p].z, b[p].w, b[p].r};
    const float cv[6] = {c[p].l, c[p].x, c[p].y, c[p].z, c[p].w, c[p].r};
#pragma unroll
    for (int j = 0; j < 6; j++) {
      cmin[p][j] = min3f(av[j], bv[j], cv[j]);
      cmax[p][j] = max3f(av[j], bv[j], cv[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bmin[p][i] = min3f(cmin[p][i], cmin[p][i + 1], cmin[p][i + 2]);
      bmax[p][i] = max3f(cmax[p][i], cmax[p][i + 1], cmax[p][i + 2]);
    }
  }
  unsigned mask = 0;
#pragma unroll
  for (int s = 0; s < NUM_SCALES; s++) {
    const int p = s + 1;
    const float av[6] = {a[p].l, a[p].x, a[p].y, a[p].z, a[p].w, a[p].r};
    const float bv[6] = {b[p].l, b[p].x, b[p].y, b[p].z, b[p].w, b[p].r};
    const float cv[6] = {c[p].l, c[p].x, c[p].y, c[p].z, c[p].w, c[p].r};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float v = bv[i + 1];
      // ring of 8 in the centre plane: left/right columns (3 rows each) + above/below
      float rmin = fminf(fminf(cmin[p][i], cmin[p][i + 2]), fminf(av[i + 1], cv[i + 1]));
      float rmax = fmaxf(fmaxf(cmax[p][i], cmax[p][i + 2]), fmaxf(av[i + 1], cv[i + 1]));
      const float minv = min3f(rmin, bmin[p - 1][i], bmin[p + 1][i]);
      const float maxv = max3f(rmax, bmax[p - 1][i], bmax[p + 1][i]);
      const bool cand = (v < fminf(-thresh, minv)) | (v > fmaxf(thresh, maxv));
      mask |= (cand ? 1u : 0u) << (5 * i + s);
    }
  }
  return mask;
}

// Append the candidates of one quad row to the frame's candidate list.
// code = x | y << 14 | s << 28.
__device__ __forceinline__ void push_candidates(unsigned mask, int q, int y, int width, unsigned *cnt,
                                                unsigned *list, unsigned cap, int octave)
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = 4 * q + i;
#pragma unroll
    for (int s = 0; s < NUM_SCALES; s++) {
      if (((mask >> (5 * i + s)) & 1u) && x < width) {
        const unsigned idx = atomicAdd(&cnt[CNT_CAND + octave], 1u);
        if (idx < cap) list[idx] = (unsigned)x | ((unsigned)y << 14) | ((unsigned)s << 28);
        else atomicAdd(&cnt[CNT_CANDOVF], 1u);
      }
    }
  }
}

__global__ __launch_bounds__(256) void detect_kernel(const float *__restrict__ dog, StripGeom g,
                                                     long long dog_frame_stride, float thresh, int octave,
                                                     unsigned *__restrict__ counters, unsigned *__restrict__ cand,
                                                     unsigned cand_cap, int aligned)
{
  const ItemCoord it = decode_item(g);
  if (!it.valid) return;
  const int lane = threadIdx.x & 63;
  const int q = it.strip * OUT_LANES + lane - 1;
  const float *planes = dog + (long long)it.frame * dog_frame_stride;
  const size_t plane = (size_t)g.height * g.pitch;
  unsigned *cnt = counters + (size_t)it.frame * CNT_STRIDE;
  unsigned *list = cand + (size_t)it.frame * cand_cap;
  const int y0 = it.seg * g.seg_rows;
  const int y1 = min(y0 + g.seg_rows, g.height);
  const bool al = aligned != 0;
  const bool tester = lane >= 1 && lane <= OUT_LANES && 4 * q < g.width;

  Row6 ra[NUM_DOG], rb[NUM_DOG], rc[NUM_DOG];
  auto ldrow = [&](Row6 (&dst)[NUM_DOG], int y) {
    const size_t off = (size_t)clampi(y, 0, g.height - 1) * g.pitch;
#pragma unroll
    for (int p = 0; p < NUM_DOG; p++) dst[p] = make_row6(load_quad(planes + p * plane + off, q, g.width, al));
  };
  ldrow(ra, y0 - 1);
  ldrow(rb, y0);
  for (int y = y0; y < y1; y++) {
    ldrow(rc, y + 1);
    // wave-uniform early-out: nothing in this row of the strip exceeds the threshold
    float amax = 0.0f;
#pragma unroll
    for (int p = 1; p <= NUM_SCALES; p++)
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(rb[p].x), fabsf(rb[p].y)), fmaxf(fabsf(rb[p].z), fabsf(rb[p].w))));
    if (__any(amax > thresh)) {
      const unsigned mask = extrema_mask(ra, rb, rc, thresh);
      if (tester && mask) push_candidates(mask, q, y, g.width, cnt, list, cand_cap, octave);
    }
#pragma unroll
    for (int p = 0; p < NUM_DOG; p++) { ra[p] = rb[p]; rb[p] = rc[p]; }
  }
}

#ifndef SCAN_UNROLL3
#define SCAN_UNROLL3 1
#endif
#ifndef SCAN_UNROLL9
#define SCAN_UNROLL9 0           // SCAN_RING = 0: the register window walks nine rows per trip with rotating names (r06)
#endif
// ------------------------------------------------------- fused DoG + scan
// dog_scan_kernel: blur -> DoG in registers (nothing but the base image is read, nothing but
// a short candidate list is written) and a cheap NECESSARY test per pixel and scale:
//   |v| > thresh  and  v is a strict extremum of its neighbours IN THE SAME ROW
//   (3 columns of the centre plane and of the adjacent centre planes).
// Survivors ("pre-candidates", typically < 0.1 % of the pixels) go to the frame's candidate
// list; refine_kernel<true> recomputes their 3x3x3 DoG neighbourhood bit-identically and
// applies the reference's full 26-neighbour test (cudaSiftD.cu:1337-1360) before refining.
// Keeping only the current DoG row in registers (no 3-row window, no box minima) leaves the
// kernel at ~1/2 the registers and ~1/3 the instructions of a full in-register 3x3x3 test.
// One wavefront's strip/segment of the fused DoG scan (see dog_scan_kernel above for the method).
// Tap pairs of the three scale pairs, re-read from LDS every row.  (Keeping them in scalar registers instead — 30 SGPRs,
// 140 instead of 168 VGPRs, 4 waves/SIMD within reach — does not work: hipcc folds the SGPR pair straight into
// v_pk_fma_f32 with op_sel_hi set, and gfx950 reads the LOW dword of an SGPR source for both halves.  r02: wrong blurs.)
struct LdsTaps {
  const v2f *tk;
  __device__ __forceinline__ Taps2 pair(int p) const { return load_taps2(tk + 5 * p); }
};

// The same 15 tap pairs held in registers for the whole segment (SCAN_TAPS_REG: 30 VGPRs, no LDS re-reads).  They are
// VGPR pairs, not SGPR pairs, so the op_sel_hi trap described above does not apply.
struct RegTaps {
  Taps2 t[NUM_SCAN_PAIRS];
  __device__ __forceinline__ explicit RegTaps(const v2f *tk) { for (int p = 0; p < NUM_SCAN_PAIRS; p++) t[p] = load_taps2(tk + 5 * p); }
  __device__ __forceinline__ const Taps2 &pair(int p) const { return t[p]; }
};

#ifndef SCAN_RING
#define SCAN_RING 1
#endif
#ifndef SCAN_DO_TEST
#define SCAN_DO_TEST 1
#endif
#ifndef SCAN_TAP_PREFETCH
#define SCAN_TAP_PREFETCH (!SCAN_RING)
#endif
#ifndef SCAN_XCH
#define SCAN_XCH 0               // 1 = neighbour lanes' vertical sums through LDS instead of DPP (r06)
#endif
#ifndef SCAN_XCH_DPP
#define SCAN_XCH_DPP 0           // SCAN_XCH: how many of the three scale pairs (the last ones) still use DPP
#endif
#ifndef SCAN_PREFETCH
#define SCAN_PREFETCH 0          // 1 (needs SCAN_XCH: vertical passes first): the ring reads + pair sums of row y+1 are issued
#endif                           //   between row y's vertical and horizontal passes, into the registers row y's sums just left
#ifndef SCAN_TAPS_REG
#define SCAN_TAPS_REG 0          // 1 = the 15 tap pairs live in 30 VGPRs across the row loop instead of being re-read from LDS
#endif
// d = a - b as ONE v_sub_f32.  Written as asm because the SLP vectoriser otherwise pairs two of the row's twenty DoG
// subtractions into a v_pk_add_f32 with a negated operand and spends three v_mov assembling its register pairs (r03 ISA:
// 10 v_mov + 5 v_pk_add + 10 v_sub per row where 20 v_sub do).  Same IEEE subtraction.
__device__ __forceinline__ float sub1(float a, float b)
{
  float r;
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(|a|, |b|, |c|) and max(m, |w|) as ONE instruction each (source modifiers): fmaxf/fabsf chains compile to twice as
// many (separate |.| pairs plus canonicalising v_max x, x) — 22 instead of 12 per row for the five planes' maxima.
__device__ __forceinline__ float absmax3(float a, float b, float c)
{
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max_abs(float m, float w)
{
  float r;
  asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(w));
  return r;
}
__device__ __forceinline__ float max3v(float a, float b, float c)
{
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
struct ScanRowCtx {
  int width, height, q;
  bool tester;
  unsigned long long tester_mask;        // ballot of `tester`, once per strip
  float thresh;
  unsigned *cnt, *list;
  unsigned cand_cap;
  int octave;
  unsigned *wq;                          // this wavefront's candidate queue (LDS, CQ_CAP words)
  float4 *xch;                           // SCAN_XCH: this wavefront's neighbour-exchange buffer + lane (LDS)
};
// ---- candidate queue (r04).  Every appended candidate used to cost its lane an atomicAdd-with-return on ONE word per
// (frame, octave): harmless in a 64-frame batch (64 x 5 words, four wavefronts per SIMD to hide the round trip) but the
// single-call path has ONE frame — a few thousand same-address atomics serialise at the memory side and every
// appending row stalls its lone wavefront for the round trip (r04 single-call sweep: dog_scan 30 us with candidates,
// 18 us without).  Now a wavefront parks its codes in LDS and asks for list space once per CQ_CAP candidates, and at the
// end of its segment once per WORKGROUP (scan_queue_finish).
#define CQ_CAP 64
#ifndef SCAN_DIRECT_APPEND
#define SCAN_DIRECT_APPEND 0     // A/B builds (tools/variants.sh): 1 = r03's per-lane atomics, same results
#endif
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// write the first qn queued codes to list[base ...] (qn <= CQ_CAP = the wavefront's width)
__device__ __forceinline__ void scan_queue_store(const unsigned *wq, unsigned qn, unsigned base, unsigned *cnt,
                                                 unsigned *list, unsigned cand_cap)
{
  const unsigned lane = threadIdx.x & 63;
  wave_lds_fence();
  if (lane < qn) {
    const unsigned idx = base + lane;
    if (idx < cand_cap) list[idx] = wq[lane];
    else atomicAdd(&cnt[CNT_CANDOVF], 1u);
  }
  wave_lds_fence();                                      // the queue may be refilled from here on
}
__device__ __forceinline__ void scan_queue_flush(const ScanRowCtx &g, unsigned &qn)
{
  if (qn == 0) return;
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(&g.cnt[CNT_CAND + g.octave], qn);
  base = __builtin_amdgcn_readfirstlane(base);
  scan_queue_store(g.wq, qn, base, g.cnt, g.list, g.cand_cap);
  qn = 0;
}
// End of a workgroup's scan: the wavefronts that still hold candidates for the same (frame, octave) counter share ONE
// atomic.  Every wavefront of the workgroup must call this (qn = 0 and ctr = nullptr when it had no item).
__device__ __forceinline__ void scan_queue_finish(const unsigned *wq, unsigned qn, unsigned *ctr, unsigned *cnt,
                                                  unsigned *list, unsigned cand_cap)
{
  __shared__ unsigned s_qn[WAVES_PER_BLOCK], s_qbase[WAVES_PER_BLOCK];
  __shared__ unsigned *s_ctr[WAVES_PER_BLOCK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_qn[wave] = qn; s_ctr[wave] = ctr; }
  __syncthreads();
  if (lane == 0 && qn) {
    bool leader = true;
    unsigned sum = 0;
    for (int w = 0; w < WAVES_PER_BLOCK; w++)
      if (s_ctr[w] == ctr && s_qn[w]) {
        if (w < wave) leader = false;
        sum += s_qn[w];
      }
    if (leader) {
      unsigned run = atomicAdd(ctr, sum);
      for (int w = 0; w < WAVES_PER_BLOCK; w++)
        if (s_ctr[w] == ctr && s_qn[w]) { s_qbase[w] = run; run += s_qn[w]; }
    }
  }
  __syncthreads();
  if (qn) scan_queue_store(wq, qn, s_qbase[wave], cnt, list, cand_cap);
}
// One row of the scan from the centre row `c` and the four vertical pair sums p1..p4 of its 9-row window.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <typename TAPS, typename MID = NoMid>
__device__ __forceinline__ void scan_row(const TAPS &taps_src, const ScanRowCtx &g, const float4 c, const float4 p1,
                                         const float4 p2, const float4 p3, const float4 p4, const int y, unsigned &qn,
                                         const MID &mid = MID())
{
  const bool tester = g.tester;
  const int q = g.q;
  const float thresh = g.thresh;
  unsigned *const cnt = g.cnt, *const list = g.list;
  const unsigned cand_cap = g.cand_cap;
  const int octave = g.octave;
  (void)list; (void)cand_cap;
  // Only blurs 1..6 are computed here: they give the five centre DoG planes d[0..4] (= reference planes
  // 1..5), which is all the necessary condition below needs; the outermost planes 0 and 6 (blurs 0 and 7)
  // are evaluated only for the survivors, by refine.  A quarter of the blur work of the dense path is saved.
  // The six blurs are computed as three scale pairs (see blur_pair); the tap pairs of the next scale pair
  // are fetched from LDS while the current one is computed.
  float4 d[NUM_SCALES];
#if SCAN_XCH
  // r06: the three vertical passes first, each handing its sums to the neighbour lanes through LDS while the next one
  // is computed; then the three horizontal passes.  SCAN_XCH_DPP pairs (the last ones) still travel by DPP, which
  // balances the VALU against the LDS pipe.
  Pair4 b0, b1, b2;
  {
    const Pair4 v0 = vert_pair(taps_src.pair(0), c, p1, p2, p3, p4);
    if (SCAN_XCH_DPP < 3) xch_put(g.xch, v0);
    const Pair4 v1 = vert_pair(taps_src.pair(1), c, p1, p2, p3, p4);
    Nbr4 n0, n1, n2;
    if (SCAN_XCH_DPP < 3) n0 = xch_get(g.xch);
    if (SCAN_XCH_DPP < 2) xch_put(g.xch, v1);
    const Pair4 v2 = vert_pair(taps_src.pair(2), c, p1, p2, p3, p4);
    if (SCAN_XCH_DPP < 2) n1 = xch_get(g.xch);
    if (SCAN_XCH_DPP < 1) { xch_put(g.xch, v2); n2 = xch_get(g.xch); }
    mid();                              // c, p1..p4 are dead from here on: the caller may fetch the next row's window (SCAN_PREFETCH)
    if (SCAN_XCH_DPP >= 3) n0 = dpp_get(v0);
    b0 = horiz_pair(taps_src.pair(0), v0, n0);
    if (SCAN_XCH_DPP >= 2) n1 = dpp_get(v1);
    b1 = horiz_pair(taps_src.pair(1), v1, n1);
    if (SCAN_XCH_DPP >= 1) n2 = dpp_get(v2);
    b2 = horiz_pair(taps_src.pair(2), v2, n2);
  }
  d[0] = make_float4(sub1(b0.x.y, b0.x.x), sub1(b0.y.y, b0.y.x), sub1(b0.z.y, b0.z.x), sub1(b0.w.y, b0.w.x));
  d[1] = make_float4(sub1(b1.x.x, b0.x.y), sub1(b1.y.x, b0.y.y), sub1(b1.z.x, b0.z.y), sub1(b1.w.x, b0.w.y));
  d[2] = make_float4(sub1(b1.x.y, b1.x.x), sub1(b1.y.y, b1.y.x), sub1(b1.z.y, b1.z.x), sub1(b1.w.y, b1.w.x));
#elif SCAN_TAP_PREFETCH
  Taps2 tcur = taps_src.pair(0), tnext = taps_src.pair(1);
  __builtin_amdgcn_sched_barrier(0);
  const Pair4 b0 = blur_pair(tcur, c, p1, p2, p3, p4);          // blurs 1, 2
  tcur = tnext;
  asm volatile("" ::: "memory");                                 // re-read from LDS: do not pin tap pairs across the row loop
  tnext = taps_src.pair(2);
  __builtin_amdgcn_sched_barrier(0);
  d[0] = make_float4(sub1(b0.x.y, b0.x.x), sub1(b0.y.y, b0.y.x), sub1(b0.z.y, b0.z.x), sub1(b0.w.y, b0.w.x));
  const Pair4 b1 = blur_pair(tcur, c, p1, p2, p3, p4);          // blurs 3, 4
  tcur = tnext;
  __builtin_amdgcn_sched_barrier(0);
  d[1] = make_float4(sub1(b1.x.x, b0.x.y), sub1(b1.y.x, b0.y.y), sub1(b1.z.x, b0.z.y), sub1(b1.w.x, b0.w.y));
  d[2] = make_float4(sub1(b1.x.y, b1.x.x), sub1(b1.y.y, b1.y.x), sub1(b1.z.y, b1.z.x), sub1(b1.w.y, b1.w.x));
  const Pair4 b2 = blur_pair(tcur, c, p1, p2, p3, p4);          // blurs 5, 6
#else
  // one set of tap pairs at a time (10 registers instead of 20): with four wavefronts per SIMD the LDS latency of the
  // re-read hides under the other wavefronts
  Taps2 tcur = taps_src.pair(0);
  __builtin_amdgcn_sched_barrier(0);
  const Pair4 b0 = blur_pair(tcur, c, p1, p2, p3, p4);          // blurs 1, 2
  asm volatile("" ::: "memory");
  tcur = taps_src.pair(1);
  __builtin_amdgcn_sched_barrier(0);
  d[0] = make_float4(sub1(b0.x.y, b0.x.x), sub1(b0.y.y, b0.y.x), sub1(b0.z.y, b0.z.x), sub1(b0.w.y, b0.w.x));
  const Pair4 b1 = blur_pair(tcur, c, p1, p2, p3, p4);          // blurs 3, 4
  asm volatile("" ::: "memory");
  tcur = taps_src.pair(2);
  __builtin_amdgcn_sched_barrier(0);
  d[1] = make_float4(sub1(b1.x.x, b0.x.y), sub1(b1.y.x, b0.y.y), sub1(b1.z.x, b0.z.y), sub1(b1.w.x, b0.w.y));
  d[2] = make_float4(sub1(b1.x.y, b1.x.x), sub1(b1.y.y, b1.y.x), sub1(b1.z.y, b1.z.x), sub1(b1.w.y, b1.w.x));
  const Pair4 b2 = blur_pair(tcur, c, p1, p2, p3, p4);          // blurs 5, 6
#endif
  d[3] = make_float4(sub1(b2.x.x, b1.x.y), sub1(b2.y.x, b1.y.y), sub1(b2.z.x, b1.z.y), sub1(b2.w.x, b1.w.y));
  d[4] = make_float4(sub1(b2.x.y, b2.x.x), sub1(b2.y.y, b2.y.x), sub1(b2.z.y, b2.z.x), sub1(b2.w.y, b2.w.x));
  // |v| maximum of every plane (two instructions each), then of the row
  float am[NUM_SCALES];
#pragma unroll
  for (int p = 0; p < NUM_SCALES; p++) am[p] = max_abs(absmax3(d[p].x, d[p].y, d[p].z), d[p].w);
  const float amax = max3v(max3v(am[0], am[1], am[2]), am[3], am[4]);
  // border rows can never hold an extremum (a clamped neighbour equals the pixel itself)
  // (tester lanes only: the halo lanes' blurs see zeros beyond the wavefront and would trip the test in every row —
  //  with them masked, 99 % of the finest level's rows of a typical frame skip the extremum tests)
  if (!SCAN_DO_TEST) {                           // register-pressure probe only (tools/variants.sh ... "-DSCAN_DO_TEST=0")
    if (wave_any(tester && amax > thresh)) {
#pragma unroll
      for (int p = 0; p < NUM_SCALES; p++) reinterpret_cast<float4 *>(list)[p * 64 + (q & 63)] = d[p];
    }
  } else if (y >= 1 && y <= g.height - 2 && (__builtin_amdgcn_fcmpf(amax, thresh, 2 /* ogt */) & g.tester_mask) != 0ull) {
    unsigned mask = 0;
#pragma unroll
    for (int s = 0; s < NUM_SCALES; s++) {
      // a row that trips the threshold usually does so in one or two planes only: the others are skipped wave-uniformly
      // (every VALU instruction costs the SIMD 3-5 cycles whatever it does; the tests were 15 % of the kernel's time)
      if ((__builtin_amdgcn_fcmpf(am[s], thresh, 2 /* ogt */) & g.tester_mask) == 0ull) continue;
      // in-row neighbourhood of centre plane d[s]: columns x-1, x, x+1 of d[s-1], d[s], d[s+1] (where available)
      float lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { lo[i] = -thresh; hi[i] = thresh; }       // the threshold rides in the min/max chains
#pragma unroll
      for (int dp = -1; dp <= 1; dp += 2) {
        if (s + dp < 0 || s + dp >= NUM_SCALES) continue;
        const float4 e = d[s + dp];
        const float v[6] = {lane_from_left(e.w), e.x, e.y, e.z, e.w, lane_from_right(e.x)};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          lo[i] = fminf(lo[i], min3f(v[i], v[i + 1], v[i + 2]));
          hi[i] = fmaxf(hi[i], max3f(v[i], v[i + 1], v[i + 2]));
        }
      }
      const float4 c = d[s];
      const float v[6] = {lane_from_left(c.w), c.x, c.y, c.z, c.w, lane_from_right(c.x)};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float nmax = max3f(hi[i], v[i], v[i + 2]);
        const float nmin = min3f(lo[i], v[i], v[i + 2]);
        const float cv = v[i + 1];
        const bool pre = cv > nmax || cv < nmin;          // nmax >= thresh, nmin <= -thresh: |cv| > thresh is implied
        mask |= (pre ? 1u : 0u) << (5 * i + s);
      }
    }
    // columns 0 and width-1 can never hold an extremum either
    if (!tester) mask = 0;
    if (4 * q == 0) mask &= ~31u;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (4 * q + i >= g.width - 1) mask &= ~(31u << (5 * i));
    if (__builtin_amdgcn_ballot_w64(mask != 0) != 0ull) {       // wave-uniform: some lane holds a candidate
      // exclusive prefix of the lanes' candidate counts (<= 20 each) from five ballots; the total is scalar
      const unsigned n = __popc(mask);
      unsigned excl = 0, total = 0;
#pragma unroll
      for (int b = 0; b < 5; b++) {
        const unsigned long long bl = __builtin_amdgcn_ballot_w64(((n >> b) & 1u) != 0u);
        excl += __builtin_amdgcn_mbcnt_hi((unsigned)(bl >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bl, 0u)) << b;
        total += (unsigned)__builtin_popcountll(bl) << b;
      }
      if (SCAN_DIRECT_APPEND || total > CQ_CAP) {
        // more candidates in ONE row of a strip than the queue holds (white noise, tiny thresholds): straight to the list
        scan_queue_flush(g, qn);
        if (mask) {
          unsigned idx = atomicAdd(&cnt[CNT_CAND + octave], n);
          while (mask) {
            const int b = __ffs(mask) - 1;
            mask &= mask - 1;
            const unsigned code = (unsigned)(4 * q + b / 5) | ((unsigned)y << 14) | ((unsigned)(b % 5) << 28);
            if (idx < cand_cap) list[idx] = code;
            else atomicAdd(&cnt[CNT_CANDOVF], 1u);
            idx++;
          }
        }
      } else {
        if (qn + total > CQ_CAP) scan_queue_flush(g, qn);
        unsigned pos = qn + excl;
        while (mask) {
          const int b = __ffs(mask) - 1;
          mask &= mask - 1;
          g.wq[pos++] = (unsigned)(4 * q + b / 5) | ((unsigned)y << 14) | ((unsigned)(b % 5) << 28);
        }
        qn += total;
      }
    }
  }
}

template <int FAST, typename TAPS>
__device__ __forceinline__ void scan_strip(const float *img, int width, int height, int pitch, int q, int lane,
                                           int y0, int y1, const TAPS &taps_src, float thresh, unsigned *cnt,
                                           unsigned *list, unsigned cand_cap, int octave, bool al, unsigned *wq,
                                           unsigned &qn, float4 *xch = nullptr)
{
  struct { int width, height, pitch; } g = {width, height, pitch};
  const bool tester = lane >= 2 && lane <= OUT_LANES - 1 && 4 * q < g.width;
  const ScanRowCtx rc = {width, height, q, tester, __builtin_amdgcn_ballot_w64(tester), thresh, cnt, list, cand_cap, octave, wq, xch};
  const QuadCol qc = make_quadcol(q, g.width);
  auto ld = [&](int y) -> float4 {
    return load_quad_t<FAST>(img + (size_t)clampi(y, 0, g.height - 1) * g.pitch, q, g.width, al, qc);
  };
  float4 r0 = ld(y0 - 4), r1 = ld(y0 - 3), r2 = ld(y0 - 2), r3 = ld(y0 - 1), r4 = ld(y0);
  float4 r5 = ld(y0 + 1), r6 = ld(y0 + 2), r7 = ld(y0 + 3), r8 = ld(y0 + 4);
  // One row of the scan; the 9-row window is passed by name so that the unrolled loop below can rotate the
  // names instead of the registers (the rotation was 36 moves per row, 6 % of the loop's VALU issue slots).
  auto row = [&](const float4 &r0, const float4 &r1, const float4 &r2, const float4 &r3, const float4 &r4,
                 const float4 &r5, const float4 &r6, const float4 &r7, const float4 &r8, const int y) {
    // re-read the taps from LDS every row instead of pinning 80 VGPRs across the loop
    if (!SCAN_TAPS_REG) asm volatile("" ::: "memory");
    const float4 c = r4, p1 = add4p(r3, r5), p2 = add4p(r2, r6), p3 = add4p(r1, r7), p4 = add4p(r0, r8);
    scan_row(taps_src, rc, c, p1, p2, p3, p4, y, qn);
  };
  int y = y0;
#if SCAN_UNROLL9
  // r06: nine rows per trip with the window's NAMES rotating — the row that leaves the window (only its pair sum with
  // the newest row needs it) hands its registers to the load of the row that enters five rows later: no copies at all
  // (the 3x form below moves 36 registers per three rows).
  {
    float4 w0 = r0, w1 = r1, w2 = r2, w3 = r3, w4 = r4, w5 = r5, w6 = r6, w7 = r7, w8 = r8;
    auto step = [&](float4 &a0, const float4 &a1, const float4 &a2, const float4 &a3, const float4 &a4,
                    const float4 &a5, const float4 &a6, const float4 &a7, const float4 &a8, const int yy)
                    __attribute__((always_inline)) {
      if (!SCAN_TAPS_REG) asm volatile("" ::: "memory");
      const float4 c = a4, p1 = add4p(a3, a5), p2 = add4p(a2, a6), p3 = add4p(a1, a7), p4 = add4p(a0, a8);
      a0 = ld(yy + 5);                           // lands under this row's math; first needed at the start of the next row
      scan_row(taps_src, rc, c, p1, p2, p3, p4, yy, qn);
    };
    for (; y + 8 < y1; y += 9) {
      step(w0, w1, w2, w3, w4, w5, w6, w7, w8, y);
      step(w1, w2, w3, w4, w5, w6, w7, w8, w0, y + 1);
      step(w2, w3, w4, w5, w6, w7, w8, w0, w1, y + 2);
      step(w3, w4, w5, w6, w7, w8, w0, w1, w2, y + 3);
      step(w4, w5, w6, w7, w8, w0, w1, w2, w3, y + 4);
      step(w5, w6, w7, w8, w0, w1, w2, w3, w4, y + 5);
      step(w6, w7, w8, w0, w1, w2, w3, w4, w5, y + 6);
      step(w7, w8, w0, w1, w2, w3, w4, w5, w6, y + 7);
      step(w8, w0, w1, w2, w3, w4, w5, w6, w7, y + 8);
    }
    r0 = w0; r1 = w1; r2 = w2; r3 = w3; r4 = w4; r5 = w5; r6 = w6; r7 = w7; r8 = w8;
  }
#elif SCAN_UNROLL3
  for (; y + 2 < y1; y += 3) {
    const float4 n0 = ld(y + 5);                 // prefetch: latency hides under the row's math
    row(r0, r1, r2, r3, r4, r5, r6, r7, r8, y);
    const float4 n1 = ld(y + 6);
    row(r1, r2, r3, r4, r5, r6, r7, r8, n0, y + 1);
    const float4 n2 = ld(y + 7);
    row(r2, r3, r4, r5, r6, r7, r8, n0, n1, y + 2);
    r0 = r3; r1 = r4; r2 = r5; r3 = r6; r4 = r7; r5 = r8; r6 = n0; r7 = n1; r8 = n2;
  }
#endif
  for (; y < y1; y++) {
    const float4 rnext = ld(y + 5);
    row(r0, r1, r2, r3, r4, r5, r6, r7, r8, y);
    r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6; r6 = r7; r7 = r8; r8 = rnext;
  }
}

#ifndef SCAN_RING
#define SCAN_RING 1
#endif
#define RING_ROWS 9
#define RING_FLOAT4S (RING_ROWS * 64)            // per wavefront: 9 rows x 64 lanes x 16 B = 9 KiB
// The same scan with the 9-row window in LDS instead of 36 VGPRs: every lane parks its quad of each incoming row in a
// ring of nine slots (its own 16 bytes of each slot: no other lane ever reads them, so no barriers) and re-reads the
// nine quads at the start of a row.  168 -> <= 128 registers (3 -> 4 wavefronts per SIMD) — which, measured, buys
// nothing by itself: the kernel is bound by VALU instruction issue and a SIMD is saturated from 2-3 wavefronts on
// (DESIGN.md section 4).  What it does save is the register rotation of the window (-3.4 % instructions).  The row
// loop is unrolled nine times so that every slot is an immediate offset of the ds_read_b128 / ds_write_b128.
template <int FAST, typename TAPS>      // 0 = generic loads, 1 = fast (width % 4 == 0), 2 = fast with ragged widths
__device__ __forceinline__ void scan_strip_ring(const float *img, int width, int height, int pitch, int q, int lane,
                                                int y0, int y1, const TAPS &taps_src, float thresh, unsigned *cnt,
                                                unsigned *list, unsigned cand_cap, int octave, bool al, float4 *mine,
                                                unsigned *wq, unsigned &qn, float4 *xch = nullptr)
{
  struct { int width, height, pitch; } g = {width, height, pitch};
  const bool tester = lane >= 2 && lane <= OUT_LANES - 1 && 4 * q < g.width;
  const ScanRowCtx rc = {width, height, q, tester, __builtin_amdgcn_ballot_w64(tester), thresh, cnt, list, cand_cap, octave, wq, xch};
  const QuadCol qc = make_quadcol(q, g.width);
  auto ld = [&](int y) -> float4 {
    return load_quad_t<FAST>(img + (size_t)clampi(y, 0, g.height - 1) * g.pitch, q, g.width, al, qc);
  };
#pragma unroll
  for (int k = 0; k < RING_ROWS; k++) mine[k * 64] = ld(y0 - 4 + k);          // slot k = row y0 - 4 + k
  float4 n = ld(y0 + 5);                                                      // one row ahead, in registers
#if SCAN_PREFETCH
  // window sums of the row about to be computed (formed during the PREVIOUS row, see below)
  float4 wc = mine[4 * 64], w1 = add4p(mine[3 * 64], mine[5 * 64]), w2 = add4p(mine[2 * 64], mine[6 * 64]),
         w3 = add4p(mine[1 * 64], mine[7 * 64]), w4 = add4p(mine[0], mine[8 * 64]);
  // one row: o0..o8 = slots (in float4 units) of rows y-4 .. y+4.  Row y's sums are in wc, w1..w4 already; slot o0 (row
  // y-4, consumed when they were formed) takes row y+5 at once, and the sums of row y+1 — slots o1..o8 and o0 — are
  // fetched in the middle of this row, when the vertical passes have released wc, w1..w4.
  auto row = [&](const int o0, const int o1, const int o2, const int o3, const int o4, const int o5, const int o6,
                 const int o7, const int o8, const int y) __attribute__((always_inline)) {
    (void)o4;
    mine[o0] = n;
    n = ld(y + 6);
    const float4 c = wc, p1 = w1, p2 = w2, p3 = w3, p4 = w4;
    scan_row(taps_src, rc, c, p1, p2, p3, p4, y, qn, [&]() __attribute__((always_inline)) {
      wc = mine[o5];
      w1 = add4p(mine[o4], mine[o6]);
      w2 = add4p(mine[o3], mine[o7]);
      w3 = add4p(mine[o2], mine[o8]);
      w4 = add4p(mine[o1], mine[o0]);
    });
  };
#else
  // one row: o0..o8 = slots (in float4 units) of rows y-4 .. y+4
  auto row = [&](const int o0, const int o1, const int o2, const int o3, const int o4, const int o5, const int o6,
                 const int o7, const int o8, const int y) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    const float4 p4 = add4p(mine[o0], mine[o8]), p3 = add4p(mine[o1], mine[o7]), p2 = add4p(mine[o2], mine[o6]),
                 p1 = add4p(mine[o3], mine[o5]), c = mine[o4];
    scan_row(taps_src, rc, c, p1, p2, p3, p4, y, qn);
    asm volatile("" ::: "memory");
    mine[o0] = n;                              // row y+5 (loaded one row ago) takes the slot of row y-4,
    n = ld(y + 6);                             // then ITS registers take the next prefetch: no second set, no copies
  };
#endif
#define RING_SLOT(J, K) ((((J) + (K)) % RING_ROWS) * 64)
#define RING_STEP(J)                                                                                                  \
  row(RING_SLOT(J, 0), RING_SLOT(J, 1), RING_SLOT(J, 2), RING_SLOT(J, 3), RING_SLOT(J, 4), RING_SLOT(J, 5),            \
      RING_SLOT(J, 6), RING_SLOT(J, 7), RING_SLOT(J, 8), y + (J))
  int y = y0;
  for (; y + RING_ROWS - 1 < y1; y += RING_ROWS) {
    RING_STEP(0); RING_STEP(1); RING_STEP(2); RING_STEP(3); RING_STEP(4); RING_STEP(5); RING_STEP(6); RING_STEP(7);
    RING_STEP(8);
  }
  // the last rows of the segment (fewer than nine): same thing with the ring phase in a scalar register
  int ph = 0;
  for (; y < y1; y++) {
    int o[RING_ROWS];
#pragma unroll
    for (int k = 0; k < RING_ROWS; k++) {
      const int t = ph + k;
      o[k] = __builtin_amdgcn_readfirstlane((t >= RING_ROWS ? t - RING_ROWS : t) * 64);
    }
    row(o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], y);
    ph = ph + 1 == RING_ROWS ? 0 : ph + 1;
  }
#undef RING_STEP
#undef RING_SLOT
}

template <bool FAST, int OCC>
__global__ __launch_bounds__(256, OCC) void dog_scan_kernel(const float *__restrict__ base, StripGeom g,
                                                            LaplaceTaps taps, float thresh, int octave,
                                                            unsigned *__restrict__ counters,
                                                            unsigned *__restrict__ cand, unsigned cand_cap, int aligned)
{
  __shared__ v2f s_taps[NUM_SCAN_PAIRS * 5];
  __shared__ unsigned s_cq[WAVES_PER_BLOCK][CQ_CAP];
  if (threadIdx.x < NUM_SCAN_PAIRS * 5) s_taps[threadIdx.x] = scan_pair_tap(taps, threadIdx.x);
  float4 *xch = nullptr;
#if SCAN_XCH && SCAN_XCH_DPP < 3
  __shared__ float4 s_xch[WAVES_PER_BLOCK][XCH_FLOAT4S];
  {
    const int w_ = threadIdx.x >> 6, l_ = threadIdx.x & 63;
    if (l_ < 4) s_xch[w_][(l_ >> 1) * XCH_PLANE + (l_ & 1) * (XCH_PLANE - 1)] = make_float4(0.f, 0.f, 0.f, 0.f);
    xch = &s_xch[w_][l_];
  }
#endif
  __syncthreads();
  const ItemCoord it = decode_item(g);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned qn = 0;
  unsigned *cnt = nullptr, *list = nullptr;
  if (it.valid) {
    // lanes 0,63: blur halo; lanes 1,62: DoG column-neighbour halo; lanes 2..61 test their quads
    const int q = it.strip * (OUT_LANES - 2) + lane - 2;
    const int y0 = it.seg * g.seg_rows;
    cnt = counters + (size_t)it.frame * CNT_STRIDE;
    list = cand + (size_t)it.frame * cand_cap;
#if SCAN_TAPS_REG
    const RegTaps tsrc(s_taps);
#else
    const LdsTaps tsrc{s_taps};
#endif
#if SCAN_RING
    __shared__ float4 s_win[WAVES_PER_BLOCK][RING_FLOAT4S];
    scan_strip_ring<FAST>(base + (long long)it.frame * g.frame_stride, g.width, g.height, g.pitch, q, lane, y0,
                          min(y0 + g.seg_rows, g.height), tsrc, thresh, cnt, list, cand_cap, octave,
                          aligned != 0, &s_win[wave][lane], s_cq[wave], qn, xch);
#else
    scan_strip<FAST>(base + (long long)it.frame * g.frame_stride, g.width, g.height, g.pitch, q, lane, y0,
                     min(y0 + g.seg_rows, g.height), tsrc, thresh, cnt, list, cand_cap, octave, aligned != 0,
                     s_cq[wave], qn, xch);
#endif
  }
  scan_queue_finish(s_cq[wave], qn, it.valid ? cnt + CNT_CAND + octave : nullptr, cnt, list, cand_cap);
}

// ---- merged-octave scan: ONE launch walks the strips of every pyramid level of every frame.
struct AllTaps { LaplaceTaps t[MISIFT_MAX_OCTAVES + 1]; };      // index = reference octave number
struct ScanOct {
  int w, h, p, nstrips, nsegs, seg_rows, octave;
  long long img_off, item_begin;
  unsigned cand_off, cand_cap;
};
struct ScanAllGeom {
  int nlev, nframes;
  int wait_lev;                               // CHAIN: levels (index into o[]) from here on wait for the embedded chain
  unsigned wait_ticks;                        // CHAIN: the wait gives up after this many ticks of the 100 MHz wall clock
  long long frame_stride, total_items;
  unsigned cand_stride;                       // candidate words per frame (all octaves)
  // options.reference_cap: refine_all's per-block extremum counters, cleared HERE (grid-stride, a few words per workgroup)
  // instead of by a hipMemsetAsync of their own — two more dispatches on the single call's critical path, +10 us (r06)
  unsigned *cap_clear;
  unsigned cap_clear_words;
  ScanOct o[MISIFT_MAX_OCTAVES];              // o[0] = finest level: the long items are dispatched first
};

#ifndef SCAN_OCC
#define SCAN_OCC (SCAN_RING ? 4 : 3)
#endif
// CHAIN (small batches, r04): the first `nchain` workgroups of the launch are the ScaleDown chain that PRODUCES the
// coarse pyramid levels; the scan items of the two finest levels do not depend on it, those of the coarse levels
// (G.wait_lev on) wait for its completion count.  Workgroups are dispatched in index order and the whole grid is
// resident, so the chain is always running before anything waits for it; by the time the coarse items (the last of
// the grid) start, it has long finished — one dependent dispatch and the chain's own duration leave the critical path.
#ifndef SCAN_STAMPS
#define SCAN_STAMPS 0            // developer build (tools/variants.sh -DSCAN_STAMPS=1): 100 MHz time stamps of the embedded chain
#endif
#if SCAN_STAMPS
#define STAMP_MAX(slot) do { if ((threadIdx.x & 63) == 0) atomicMax(&counters[(size_t)G.nframes * CNT_STRIDE + (slot)], (unsigned)wall_clock64()); } while (0)
#else
#define STAMP_MAX(slot) do { } while (0)
#endif
template <int FAST, bool CHAIN>
__global__ __launch_bounds__(256, SCAN_OCC) void dog_scan_all_kernel(const float *__restrict__ scratch, ScanAllGeom G,
                                                              AllTaps taps, float thresh,
                                                              unsigned *__restrict__ counters,
                                                              unsigned *__restrict__ cand, ChainGeom C, Taps5 k5,
                                                              int nchain)
{
  __shared__ v2f s_taps[WAVES_PER_BLOCK][NUM_SCAN_PAIRS * 5];
#if SCAN_RING
  __shared__ float4 s_win[WAVES_PER_BLOCK][RING_FLOAT4S];
  static_assert(sizeof(float4) * WAVES_PER_BLOCK * RING_FLOAT4S >= sizeof(float) * CHAIN_LDS_FLOATS_EMBED,
                "the chain borrows the scan's row ring");
#else
  // (register window: the embedded chain gets an LDS area of its own in the CHAIN instantiations)
  __shared__ float s_chain_lds[CHAIN ? CHAIN_LDS_FLOATS_EMBED : 4];
#endif
#if SCAN_XCH && SCAN_XCH_DPP < 3
  __shared__ float4 s_xch[WAVES_PER_BLOCK][XCH_FLOAT4S];
#endif
  if (G.cap_clear_words)
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < G.cap_clear_words; i += gridDim.x * 256u) G.cap_clear[i] = 0u;
  // no XCD remap here: items of different levels cost differently, and the hardware's round-robin
  // block -> XCD placement is what keeps the eight XCDs evenly loaded across the level boundaries
  unsigned lb = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  bool chain_failed = false;
  float4 *xch = nullptr;
#if SCAN_XCH && SCAN_XCH_DPP < 3
  if (lane < 4) s_xch[wave][(lane >> 1) * XCH_PLANE + (lane & 1) * (XCH_PLANE - 1)] = make_float4(0.f, 0.f, 0.f, 0.f);
  xch = &s_xch[wave][lane];
#endif
  if (CHAIN) {
#if SCAN_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[(size_t)G.nframes * CNT_STRIDE + 8] = (unsigned)wall_clock64();
    if (threadIdx.x == 0) STAMP_MAX(14);        // start of the last workgroup to start
    if (threadIdx.x == 0 && lb + 1 == gridDim.x) STAMP_MAX(16);     // start of the workgroup with the highest index
    if (threadIdx.x == 0 && lb == (unsigned)nchain) STAMP_MAX(17);  // start of the first scan workgroup
#endif
    if (lb < (unsigned)nchain) {                  // workgroup-uniform
      const int tiles = C.tiles_x * C.tiles_y;
#if SCAN_RING
      float *chain_lds = reinterpret_cast<float *>(&s_win[0][0]);
#else
      float *chain_lds = s_chain_lds;
#endif
      scaledown_chain_block(const_cast<float *>(scratch), C, k5, (int)(lb % tiles), (int)(lb / tiles), chain_lds);
      // the chain's stores are write-through (chain.hpp): once they have been acknowledged they are in memory, where
      // every XCD finds them — no L2 write-back (an agent-scope release fence per workgroup: +45 us per frame, r04).
      // "Once they have been acknowledged" is THIS wait: the workgroup-scope fence compiles to s_waitcnt lgkmcnt(0) only,
      // and without vmcnt(0) the ticket below could overtake the stores (advisor r04: global_store ... sc1 -> s_barrier ->
      // global_atomic_add with no wait in between).  Each wavefront waits for its own stores; cheap.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      STAMP_MAX(9);
      // Completion count in two levels: 16 ticket words in cache lines of their own, then one more for the workgroups
      // that drew the last ticket of theirs; the very last raises the flag the coarse items poll.  Same-address atomics
      // are served one at a time (10-50 ns each): 510 tickets on ONE word held the flag back by 14 us, and polling the
      // ticket word itself starves the tickets (r04, SCAN_STAMPS).  All of it lives in the spare counter blocks behind
      // the last frame's (cleared by the prefilter).
      if (threadIdx.x == 0) {
        unsigned *spare = counters + (size_t)G.nframes * CNT_STRIDE;
        const unsigned sub = lb & 15u, members = ((unsigned)nchain - sub + 15u) / 16u;
        if (atomicAdd(&spare[64 + 32 * sub], 1u) == members - 1u) {
          const unsigned nsub = (unsigned)nchain < 16u ? (unsigned)nchain : 16u;
          if (atomicAdd(&spare[32], 1u) == nsub - 1u && G.wait_ticks != 0u) {     // (a bound of 0 = the fallback's test: the
            __hip_atomic_fetch_or(&spare[0], 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    //  flag is never raised)
            STAMP_MAX(18);
          }
        }
      }
      STAMP_MAX(19);
      return;
    }
    lb -= (unsigned)nchain;
    // a workgroup that holds an item of a coarse level waits for the flag: ONE lane polls, the others sit at the barrier
    if (G.wait_lev < G.nlev && (long long)lb * WAVES_PER_BLOCK + (WAVES_PER_BLOCK - 1) >= G.o[G.wait_lev].item_begin) {
      STAMP_MAX(15);
      __shared__ unsigned s_chain_ok;
      if (threadIdx.x == 0) {
        // (a read-modify-write: it is performed at the memory side.  An agent-scope atomic LOAD may be served from this
        //  XCD's L2, which is not coherent with the other XCDs' — the first poll caches the line and the loop then spins on
        //  the stale copy until it happens to be evicted: 40 us, measured with SCAN_STAMPS)
        //  — and an add of ZERO is folded into such a load by the compiler: every poll adds one, the flag is the top bit)
        // BOUNDED: forward progress rests on the chain workgroups (lowest block ids) having been dispatched before the
        // ones that wait for them.  Should that ever fail (a partition mode, a debugger, a future dispatcher), the wait
        // gives up after `wait_ticks` of the 100 MHz clock, the workgroup skips its items and raises CNT_CHAINTMO in
        // frame 0's counter block; the host sees it with the counts and re-runs the call with a stand-alone chain launch.
        const unsigned long long t0 = wall_clock64();
        unsigned ok = 1u;
        while ((__hip_atomic_fetch_add(&counters[(size_t)G.nframes * CNT_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
                0x80000000u) == 0u) {
          if (wall_clock64() - t0 > (unsigned long long)G.wait_ticks) { ok = 0u; break; }
          __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) atomicAdd(&counters[CNT_CHAINTMO], 1u);
        s_chain_ok = ok;
      }
      __syncthreads();
      STAMP_MAX(13);
      // nothing of the coarse levels can be in this XCD's L2 yet (caches are invalidated at kernel start and only
      // wavefronts behind this wait read those levels): a compiler-level acquire is all that is needed
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      chain_failed = s_chain_ok == 0u;             // workgroup-uniform
    }
  }
  long long item = (long long)lb * WAVES_PER_BLOCK + wave;
  const bool valid = item < G.total_items && !chain_failed;      // (no early return: scan_queue_finish is a workgroup barrier)
  __shared__ unsigned s_cq[WAVES_PER_BLOCK][CQ_CAP];
  unsigned qn = 0;
  unsigned *cnt = nullptr, *list = nullptr;
  unsigned cand_cap = 0;
  int octave = 0;
  bool stamp_coarse = false;
  (void)stamp_coarse;
  if (valid) {
    int lev = 0;
    for (int k = 1; k < G.nlev; k++)
      if (item >= G.o[k].item_begin) lev = k;
    lev = __builtin_amdgcn_readfirstlane(lev);
    stamp_coarse = lev >= G.wait_lev;
    const ScanOct &L = G.o[lev];
    item -= L.item_begin;
    const int seg = (int)(item % L.nsegs);
    const long long r = item / L.nsegs;
    const int strip = (int)(r % L.nstrips);
    const int frame = (int)(r / L.nstrips);
    // this wavefront's private copy of its octave's tap pairs
    if (lane < NUM_SCAN_PAIRS * 5) s_taps[wave][lane] = scan_pair_tap(taps.t[L.octave], lane);
    wave_lds_fence();
    const int q = strip * (OUT_LANES - 2) + lane - 2;
    const int y0 = seg * L.seg_rows;
    cnt = counters + (size_t)frame * CNT_STRIDE;
    list = cand + (size_t)frame * G.cand_stride + L.cand_off;
    cand_cap = L.cand_cap;
    octave = L.octave;
#if SCAN_TAPS_REG
    const RegTaps tsrc(s_taps[wave]);
#else
    const LdsTaps tsrc{s_taps[wave]};
#endif
#if SCAN_RING
    scan_strip_ring<FAST>(scratch + (long long)frame * G.frame_stride + L.img_off, L.w, L.h, L.p, q, lane, y0,
                          min(y0 + L.seg_rows, L.h), tsrc, thresh, cnt, list, cand_cap, octave, true,
                          &s_win[wave][lane], s_cq[wave], qn, xch);
#else
    scan_strip<FAST>(scratch + (long long)frame * G.frame_stride + L.img_off, L.w, L.h, L.p, q, lane, y0,
                     min(y0 + L.seg_rows, L.h), tsrc, thresh, cnt, list, cand_cap, octave, true,
                     s_cq[wave], qn, xch);
#endif
  }
#if SCAN_STAMPS
  if (CHAIN && valid) { if (stamp_coarse) STAMP_MAX(12); else STAMP_MAX(11); }
#endif
  scan_queue_finish(s_cq[wave], qn, valid ? cnt + CNT_CAND + octave : nullptr, cnt, list, cand_cap);
}

// ------------------------------------------------------------------- refine
struct RefineParams {
  int width, height, pitch, nframes;
  float edge_limit, factor, lowest_scale, subsampling, thresh;
  float scmul[NUM_SCALES];     // powf(2, s/NUM_SCALES), computed on the host like the oracle
  int octave, max_pts;
  unsigned cand_cap;
};

// 3x3x3 DoG neighbourhood of candidate (x, y, scale s): d[p] = B[s+p+1] - B[s+p], the blurs recomputed from
// the octave base image with the chains of blur_quad2() / oracle orc_laplace (vertical pass first, then
// horizontal; clamp-to-edge) — bit-identical to the streamed DoG.
__device__ __forceinline__ void dog_patch(const float *img, int w, int h, int pitch, const float (&tk)[4][5],
                                          int x, int y, float (&d)[3][3][3])
{
  int xo[11];
#pragma unroll
  for (int cx = 0; cx < 11; cx++) xo[cx] = clampi(x + cx - 5, 0, w - 1);
  float prev[3][3];
#pragma unroll
  for (int bs = 0; bs < 4; bs++) {
    const float k0 = tk[bs][0], k1 = tk[bs][1], k2 = tk[bs][2], k3 = tk[bs][3], k4 = tk[bs][4];
    float vres[3][11];                   // vertical results: [row y-1..y+1][col x-5..x+5]
#pragma unroll
    for (int cx = 0; cx < 11; cx++) {
      const float *col = img + xo[cx];
      float r[11];                       // rows y-5 .. y+5 (clamped)
#pragma unroll
      for (int j = 0; j < 11; j++) r[j] = col[(size_t)clampi(y + j - 5, 0, h - 1) * pitch];
#pragma unroll
      for (int dy = 0; dy < 3; dy++)
        vres[dy][cx] = conv9(k0, k1, k2, k3, k4, r[dy + 4], r[dy + 3] + r[dy + 5], r[dy + 2] + r[dy + 6],
                             r[dy + 1] + r[dy + 7], r[dy + 0] + r[dy + 8]);
    }
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const float *v = &vres[dy][dx];       // v[4] is the centre column
        const float cur = conv9(k0, k1, k2, k3, k4, v[4], v[3] + v[5], v[2] + v[6], v[1] + v[7], v[0] + v[8]);
        if (bs > 0) d[bs - 1][dy][dx] = cur - prev[dy][dx];
        prev[dy][dx] = cur;
      }
  }
}

// Full 26-neighbour test + edge test + 3-D quadratic refinement of one candidate whose 3x3x3 DoG
// neighbourhood is d[plane s..s+2][dy][dx] (cudaSiftD.cu:1337-1360, :1383-1417; same expression order
// as oracle orc_findpoints()).  Returns false if the candidate is rejected.
// 2^x as the written-out fmaf chain of oracle det_exp2() (sift_oracle.c): the keypoint scale then agrees bit for bit
// with the oracle's, and with it everything the orientation / descriptor kernels decide from it.
__device__ __forceinline__ float det_exp2(float x)
{
  const bool tiny = x < -125.0f;
  x = x > 126.0f ? 126.0f : x;         // not fminf: a NaN exponent (singular Hessian) must stay NaN and be rejected, like exp2f
  x = tiny ? 0.0f : x;
  const float n = rintf(x);
  const float r = x - n;
  float p = __builtin_fmaf(1.535336188319500e-4f, r, 1.339887440266574e-3f);
  p = __builtin_fmaf(p, r, 9.618437357674640e-3f);
  p = __builtin_fmaf(p, r, 5.550332471162809e-2f);
  p = __builtin_fmaf(p, r, 2.402264791363012e-1f);
  p = __builtin_fmaf(p, r, 6.931472028550421e-1f);
  p = __builtin_fmaf(p, r, 1.0f);
  const float sc = __builtin_bit_cast(float, ((int)n + 127) << 23);
  return tiny ? 0.0f : p * sc;
}

struct Refined { float xpos, ypos, scale, sharpness, edgeness; };
__device__ __forceinline__ bool refine_math(const float (&d)[3][3][3], int x, int y, int s, float thresh,
                                            float edge_limit, float factor, float lowest_scale,
                                            const float (&scmul)[NUM_SCALES], Refined &out, bool *extremum = nullptr)
{
  const float val = d[1][1][1];
  if (extremum) *extremum = false;
  {
    // candidates are interior pixels, so the reference's clamped neighbour addressing never applies here
    float minv = INFINITY, maxv = -INFINITY;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
          if (!(p == 1 && dy == 1 && dx == 1)) {
            minv = fminf(minv, d[p][dy][dx]);
            maxv = fmaxf(maxv, d[p][dy][dx]);
          }
    if (!((val < fminf(-thresh, minv)) || (val > fmaxf(thresh, maxv)))) return false;
    if (extremum) *extremum = true;      // what the reference's 32-per-block cap counts (cudaSiftD.cu:1354-1377)
  }
  const float dxx = 2.0f * val - d[1][1][0] - d[1][1][2];
  const float dyy = 2.0f * val - d[1][0][1] - d[1][2][1];
  const float dxy = 0.25f * (d[1][2][2] + d[1][0][0] - d[1][0][2] - d[1][2][0]);
  const float tra = dxx + dyy;
  const float det = dxx * dyy - dxy * dxy;
  if (!(tra * tra < edge_limit * det)) return false;
  const float edge = (tra * tra) / det;
  const float dx = 0.5f * (d[1][1][2] - d[1][1][0]);
  const float dy = 0.5f * (d[1][2][1] - d[1][0][1]);
  const float ds = 0.5f * (d[0][1][1] - d[2][1][1]);
  const float dss = 2.0f * val - d[2][1][1] - d[0][1][1];
  const float dxs = 0.25f * (d[2][1][2] + d[0][1][0] - d[0][1][2] - d[2][1][0]);
  const float dys = 0.25f * (d[2][2][1] + d[0][0][1] - d[2][0][1] - d[0][2][1]);
  const float idxx = dyy * dss - dys * dys;
  const float idxy = dys * dxs - dxy * dss;
  const float idxs = dxy * dys - dyy * dxs;
  const float idet = 1.0f / (idxx * dxx + idxy * dxy + idxs * dxs);
  const float idyy = dxx * dss - dxs * dxs;
  const float idys = dxy * dxs - dxx * dys;
  const float idss = dxx * dyy - dxy * dxy;
  float pdx = idet * (idxx * dx + idxy * dy + idxs * ds);
  float pdy = idet * (idxy * dx + idyy * dy + idys * ds);
  float pds = idet * (idxs * dx + idys * dy + idss * ds);
  if (pdx < -0.5f || pdx > 0.5f || pdy < -0.5f || pdy > 0.5f || pds < -0.5f || pds > 0.5f) {
    pdx = dx / dxx;
    pdy = dy / dyy;
    pds = ds / dss;
  }
  const float dval = 0.5f * (dx * pdx + dy * pdy + ds * pds);
  float scm = scmul[0];
#pragma unroll
  for (int j = 1; j < NUM_SCALES; j++) scm = (s == j) ? scmul[j] : scm;
  const float sc = scm * det_exp2(pds * factor);
  if (!(sc >= lowest_scale)) return false;
  out.xpos = x + pdx;
  out.ypos = y + pdy;
  out.scale = sc;
  out.sharpness = val + dval;
  out.edgeness = edge;
  return true;
}

// 3x3x3 DoG neighbourhood of candidate (x, y, s) recomputed from the octave base image.
__device__ __forceinline__ void dog_from_base(const float *img, int w, int h, int pitch, const LaplaceTaps &taps,
                                              int x, int y, int s, float (&d)[3][3][3])
{
  float tk[4][5];                    // taps of blur scales s .. s+3 (static selects, no dynamic register indexing)
#pragma unroll
  for (int bs = 0; bs < 4; bs++)
#pragma unroll
    for (int j = 0; j < 5; j++) {
      float t = taps.k[bs][j];
#pragma unroll
      for (int ss = 1; ss < NUM_SCALES; ss++) t = (s == ss) ? taps.k[ss + bs][j] : t;
      tk[bs][j] = t;
    }
  dog_patch(img, w, h, pitch, tk, x, y, d);
}

template <bool FROM_BASE>
__global__ __launch_bounds__(64) void refine_kernel(const float *__restrict__ src, long long src_frame_stride,
                                                    LaplaceTaps taps, RefineParams P,
                                                    unsigned *__restrict__ counters,
                                                    const unsigned *__restrict__ cand, SiftPointD *__restrict__ pts)
{
  const int frame = blockIdx.y;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  const unsigned *list = cand + (size_t)frame * P.cand_cap;
  SiftPointD *out = pts + (size_t)frame * P.max_pts;
  const float *img = src + (long long)frame * src_frame_stride;
  const int o = P.octave;
  // counter protocol of cudaSiftD.cu:1297-1300: octave o starts where octave o-1 ended
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned start = cnt[2 * o - 1];
    atomicMax(&cnt[2 * o + 0], start);
    atomicMax(&cnt[2 * o + 1], start);
  }
  const unsigned ncand = min(cnt[CNT_CAND + o], P.cand_cap);
  const size_t plane = (size_t)P.height * P.pitch;
  for (unsigned ci = blockIdx.x * blockDim.x + threadIdx.x; ci < ncand; ci += gridDim.x * blockDim.x) {
    const unsigned code = list[ci];
    if (code == 0xffffffffu) continue;        // struck by the reference's 32-per-block cap (options.reference_cap)
    const int x = code & 0x3fff, y = (code >> 14) & 0x3fff, s = code >> 28;
    float d[3][3][3];       // [plane s..s+2][dy][dx]
    if (FROM_BASE) {
      dog_from_base(img, P.width, P.height, P.pitch, taps, x, y, s, d);
    } else {
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++)
            d[p][dy][dx] = img[(size_t)(s + p) * plane + (size_t)(y + dy - 1) * P.pitch + (x + dx - 1)];
    }
    Refined r;
    if (!refine_math(d, x, y, s, P.thresh, P.edge_limit, P.factor, P.lowest_scale, P.scmul, r)) continue;
    atomicMax(&cnt[2 * o + 0], cnt[2 * o - 1]);
    const unsigned idx = atomicAdd(&cnt[2 * o + 0], 1u);
    if (idx >= (unsigned)P.max_pts) { atomicAdd(&cnt[CNT_PTOVF], 1u); continue; }
    SiftPointD *p = &out[idx];
    p->xpos = r.xpos;
    p->ypos = r.ypos;
    p->scale = r.scale;
    p->sharpness = r.sharpness;
    p->edgeness = r.edgeness;
    p->subsampling = P.subsampling;
  }
}

#define REFCAP_W 30               // a block of FindPointsMultiNew: MINMAX_W x MINMAX_H pixels of one scale (cudaSiftD.h)
#define REFCAP_H 8
#define REFCAP_WORDS 8            // 240 bits per block
// Merged-octave refine: all candidates of all octaves of a frame in one launch; survivors go to the
// per-octave staging area as Detection records (orient_all_kernel / descr_all_kernel take it from there).
struct RefineAllParams {
  float thresh, edge_limit, factor;
  float scmul[NUM_SCALES];
  int max_pts;
  unsigned cand_stride;
  // options.reference_cap on the fused path (r06): byte counters of TRUE extrema per (octave, scale, 30 x 8 block), four to
  // a word; a block that reaches a 33rd extremum raises the frame's CNT_CANDOVF and the callers redo that frame on the
  // dense kernels, which apply the cap in the reference's order (launch_refcap) — frames in which no block reaches it
  // (every natural image) ARE the reference's result as they stand.  cap_words = 0: off.
  unsigned cap_words;                               // words per frame
  unsigned cap_limit;                               // 32 (MEMWID)
  unsigned cap_off[MISIFT_MAX_OCTAVES + 1];         // first counter of octave o
  int cap_tx[MISIFT_MAX_OCTAVES + 1], cap_ty[MISIFT_MAX_OCTAVES + 1];
};
// Work decomposition (rocprof: the one-lane-per-candidate version needed 326 VGPRs -> 1 wave/SIMD, and each
// of its 121 gathers touched 64 different cache lines): SIXTEEN lanes share one candidate, four candidates
// per wavefront.  Lane c of a group owns column x-5+c of the 11x11 patch (11 row loads per lane, each load
// instruction covers 4 x 11 contiguous pixels), blurs it vertically for the 4 scales and 3 rows, fetches the
// horizontal neighbours with DPP row shifts inside its 16-lane row, and lanes 4..6 end up with the DoG values
// of columns x-1..x+1; lane 5 collects the 27 values and runs the tests/refinement.  ~60 VGPRs, every value
// computed with the same expression as dog_patch() (bit-identical).
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define ROW_SHR(n) (0x110 + (n))      // lane i reads lane i-n of its 16-lane row
#define ROW_SHL(n) (0x100 + (n))      // lane i reads lane i+n of its 16-lane row

__global__ __launch_bounds__(256) void refine_all_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                         AllTaps taps, RefineAllParams R,
                                                         unsigned *__restrict__ counters,
                                                         const unsigned *__restrict__ cand,
                                                         Detection *__restrict__ det, unsigned *__restrict__ capcnt)
{
  const int frame = blockIdx.y;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * R.max_pts;
  const int lane = threadIdx.x & 63, c = lane & 15;
  const unsigned group = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + (unsigned)(lane >> 4);
  const unsigned ngroups = gridDim.x * 16u;
  // candidates of all octaves are flattened (finest first) so no group idles through octaves it has no work in
  // the per-octave candidate counts are final (the scan is done): read them once, into scalar registers
  unsigned ncand[MISIFT_MAX_OCTAVES + 1];
  unsigned total = 0;
#pragma unroll
  for (int k = 1; k <= MISIFT_MAX_OCTAVES; k++) {
    ncand[k] = k <= P.noct ? __builtin_amdgcn_readfirstlane(min(cnt[CNT_CAND + k], P.o[k].cand_cap)) : 0u;
    total += ncand[k];
  }
  __shared__ unsigned s_n[MISIFT_MAX_OCTAVES + 1], s_base[MISIFT_MAX_OCTAVES + 1];
  if (threadIdx.x <= MISIFT_MAX_OCTAVES) s_n[threadIdx.x] = 0;
  __syncthreads();
  const unsigned rounds = (total + ngroups - 1) / ngroups;       // workgroup-uniform trip count (DPP needs all lanes, the
  for (unsigned it = 0; it < rounds; it++) {                     //  slot hand-out below all wavefronts)
    const unsigned fi = it * ngroups + group;
    const bool live = fi < total;
    int o = P.noct;
    unsigned ci = live ? fi : 0u;
    bool found = false;
#pragma unroll
    for (int k = MISIFT_MAX_OCTAVES; k >= 1; k--) {
      if (!found && k <= P.noct) {
        if (ci < ncand[k]) { o = k; found = true; }
        else ci -= ncand[k];
      }
    }
    const int lw = P.o[o].w, lh = P.o[o].h, lp = P.o[o].p;
    const float *img = scratch + (long long)frame * P.frame_stride + P.o[o].img_off;
    const unsigned code = live ? cand[(size_t)frame * R.cand_stride + P.o[o].cand_off + ci] : 0u;
    const int x = code & 0x3fff, y = (code >> 14) & 0x3fff, s = code >> 28;
    // ---- this lane's column of the patch: rows y-5 .. y+5 (clamped), column x-5+c (clamped; lanes 11..15 idle along)
    const float *col = img + clampi(x + min(c, 10) - 5, 0, lw - 1);
    float r[11];
#pragma unroll
    for (int j = 0; j < 11; j++) r[j] = col[(size_t)clampi(y + j - 5, 0, lh - 1) * lp];
    // pair sums of the vertical taps are shared by the four scales
    float ps[3][4];
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int k = 1; k <= 4; k++) ps[dy][k - 1] = r[dy + 4 - k] + r[dy + 4 + k];
    float d[3][3];                     // [plane s..s+2][dy] at this lane's column (meaningful for c = 4, 5, 6)
    float prev[3];
#pragma unroll
    for (int bs = 0; bs < 4; bs++) {
      const float *tk = taps.t[o].k[s + bs];
      const float k0 = tk[0], k1 = tk[1], k2 = tk[2], k3 = tk[3], k4 = tk[4];
#pragma unroll
      for (int dy = 0; dy < 3; dy++) {
        const float v = conv9(k0, k1, k2, k3, k4, r[dy + 4], ps[dy][0], ps[dy][1], ps[dy][2], ps[dy][3]);
        const float cur = conv9(k0, k1, k2, k3, k4, v,
                                row_dpp<ROW_SHR(1)>(v) + row_dpp<ROW_SHL(1)>(v),
                                row_dpp<ROW_SHR(2)>(v) + row_dpp<ROW_SHL(2)>(v),
                                row_dpp<ROW_SHR(3)>(v) + row_dpp<ROW_SHL(3)>(v),
                                row_dpp<ROW_SHR(4)>(v) + row_dpp<ROW_SHL(4)>(v));
        if (bs > 0) d[bs - 1][dy] = cur - prev[dy];
        prev[dy] = cur;
      }
    }
    // ---- lane 5 of the group collects columns x-1 (lane 4) and x+1 (lane 6)
    float dd[3][3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int dy = 0; dy < 3; dy++) {
        dd[p][dy][0] = row_dpp<ROW_SHR(1)>(d[p][dy]);
        dd[p][dy][1] = d[p][dy];
        dd[p][dy][2] = row_dpp<ROW_SHL(1)>(d[p][dy]);
      }
    Refined rr;
    bool extremum = false;
    const bool ok = live && c == 5 &&
                    refine_math(dd, x, y, s, R.thresh, R.edge_limit, R.factor, P.o[o].lowest_scale, R.scmul, rr, &extremum);
    if (R.cap_words && live && c == 5 && extremum) {
      const unsigned idx = R.cap_off[o] + ((unsigned)s * R.cap_ty[o] + (unsigned)(y / REFCAP_H)) * R.cap_tx[o] + (unsigned)(x / REFCAP_W);
      const unsigned sh = 8u * (idx & 3u);
      const unsigned old = atomicAdd(&capcnt[(size_t)frame * R.cap_words + (idx >> 2)], 1u << sh);      // (<= 240 per block: no carry)
      if (((old >> sh) & 0xffu) == R.cap_limit) atomicAdd(&cnt[CNT_CANDOVF], 1u);      // the 33rd: this frame needs the cap applied
    }
    // Staging slots are handed out per WORKGROUP and octave: the survivors of a round take a rank from an LDS counter
    // and one thread per octave asks the frame's counter for that many slots.  (One atomicAdd-with-return per survivor
    // on the same word serialises at the memory side: ~2000 of them were 17 us of a single frame's 22 us refine, r04.)
    unsigned rank = 0;
    if (ok) rank = atomicAdd(&s_n[o], 1u);
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= MISIFT_MAX_OCTAVES) {
      const unsigned n = s_n[threadIdx.x];
      if (n) {
        s_base[threadIdx.x] = atomicAdd(&cnt[CNT_DET + threadIdx.x], n);
        s_n[threadIdx.x] = 0;
      }
    }
    __syncthreads();
    if (!ok) continue;
    const unsigned idx = s_base[o] + rank;
    if (idx >= (unsigned)R.max_pts) { atomicAdd(&cnt[CNT_PTOVF], 1u); continue; }
    Detection *pd = &fdet[(size_t)(o - 1) * R.max_pts + idx];
    pd->xpos = rr.xpos;
    pd->ypos = rr.ypos;
    pd->scale = rr.scale;
    pd->sharpness = rr.sharpness;
    pd->edgeness = rr.edgeness;
  }
}

// ------------------------------------------------------------- host wrappers
static inline bool is_aligned16(const void *p, int pitch) { return (((uintptr_t)p) & 15) == 0 && (pitch & 3) == 0; }

static inline dim3 grid_for(const StripGeom &g)
{
  const long long nitems = (long long)g.nframes * g.nstrips * g.nsegs;
  return dim3((unsigned)((nitems + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
}

int launch_laplace(misift_ctx *ctx, const float *base, const StripGeom &g, float *dog,
                   long long dog_frame_stride, const LaplaceTaps &taps)
{
  const int al = is_aligned16(base, g.pitch) && is_aligned16(dog, g.pitch) && (g.frame_stride & 3) == 0 &&
                 (dog_frame_stride & 3) == 0;
  LaunchScope ls(ctx, "laplace");
  if (al && (g.width & 3) == 0)
    hipLaunchKernelGGL(laplace_kernel<true>, grid_for(g), dim3(256), 0, ctx->stream, base, g, dog, dog_frame_stride,
                       taps, al);
  else
    hipLaunchKernelGGL(laplace_kernel<false>, grid_for(g), dim3(256), 0, ctx->stream, base, g, dog,
                       dog_frame_stride, taps, al);
  return ls.finish();
}

int launch_detect(misift_ctx *ctx, const float *dog, const StripGeom &g, long long dog_frame_stride,
                  float thresh, int octave)
{
  const int al = is_aligned16(dog, g.pitch) && (dog_frame_stride & 3) == 0;
  LaunchScope ls(ctx, "detect");
  hipLaunchKernelGGL(detect_kernel, grid_for(g), dim3(256), 0, ctx->stream, dog, g, dog_frame_stride, thresh,
                     octave, ctx->d_counters, ctx->d_cand, (unsigned)ctx->cand_cap, al);
  return ls.finish();
}

// ---- options.reference_cap: the 32-extrema-per-block cap of FindPointsMultiNew (cudaSiftD.cu:1369-1377).  A block of the
// reference is 30 columns x 8 rows of one scale; it numbers its extrema by an exclusive prefix sum over its threads
// (columns), each thread listing its own by row, and only the first MEMWID = 32 are handed on.  Here: every true extremum
// of the dense detect_kernel sets ITS bit in a 240-bit mask of its block (bit = column * 8 + row: the reference's order),
// then every one whose rank — the number of set bits below its own — is 32 or more is struck from the candidate list.
__device__ __forceinline__ void refcap_locate(unsigned code, int tiles_x, int tiles_y, size_t *word0, unsigned *bit)
{
  const int x = code & 0x3fff, y = (code >> 14) & 0x3fff, s = code >> 28;
  const int tx = x / REFCAP_W, ty = y / REFCAP_H;
  *word0 = (((size_t)s * tiles_y + ty) * tiles_x + tx) * REFCAP_WORDS;
  *bit = (unsigned)((x - tx * REFCAP_W) * REFCAP_H + (y - ty * REFCAP_H));
}
__global__ __launch_bounds__(256) void refcap_mark_kernel(const unsigned *__restrict__ counters, const unsigned *__restrict__ cand,
                                                          unsigned cand_cap, int octave, int tiles_x, int tiles_y,
                                                          unsigned *__restrict__ mask, size_t mask_words_per_frame)
{
  const int frame = blockIdx.y;
  const unsigned n = min(counters[(size_t)frame * CNT_STRIDE + CNT_CAND + octave], cand_cap);
  const unsigned *list = cand + (size_t)frame * cand_cap;
  unsigned *m = mask + (size_t)frame * mask_words_per_frame;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    size_t w0; unsigned bit;
    refcap_locate(list[i], tiles_x, tiles_y, &w0, &bit);
    atomicOr(&m[w0 + (bit >> 5)], 1u << (bit & 31u));
  }
}
__global__ __launch_bounds__(256) void refcap_drop_kernel(const unsigned *__restrict__ counters, unsigned *__restrict__ cand,
                                                          unsigned cand_cap, int octave, int tiles_x, int tiles_y,
                                                          const unsigned *__restrict__ mask, size_t mask_words_per_frame)
{
  const int frame = blockIdx.y;
  const unsigned n = min(counters[(size_t)frame * CNT_STRIDE + CNT_CAND + octave], cand_cap);
  unsigned *list = cand + (size_t)frame * cand_cap;
  const unsigned *m = mask + (size_t)frame * mask_words_per_frame;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    size_t w0; unsigned bit;
    refcap_locate(list[i], tiles_x, tiles_y, &w0, &bit);
    unsigned rank = 0;
    for (unsigned w = 0; w < (bit >> 5); w++) rank += __popc(m[w0 + w]);
    rank += __popc(m[w0 + (bit >> 5)] & ((1u << (bit & 31u)) - 1u));
    if (rank >= 32u) list[i] = 0xffffffffu;            // struck: refine_kernel skips it (no pixel has this code)
  }
}

int launch_refcap(misift_ctx *ctx, int w, int h, int nframes, int octave)
{
  const int tiles_x = (w + REFCAP_W - 1) / REFCAP_W, tiles_y = (h + REFCAP_H - 1) / REFCAP_H;
  const size_t words = (size_t)NUM_SCALES * tiles_x * tiles_y * REFCAP_WORDS;
  if (sizeof(unsigned) * words * nframes > ctx->refcap_bytes) {
    if (ctx->d_refcap) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(misift_dev_free(ctx->d_refcap)); }
    ctx->d_refcap = nullptr; ctx->refcap_bytes = 0;
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_refcap, sizeof(unsigned) * words * nframes, "refcap_masks"));
    ctx->refcap_bytes = sizeof(unsigned) * words * nframes;
  }
  HIP_TRY(hipMemsetAsync(ctx->d_refcap, 0, sizeof(unsigned) * words * nframes, ctx->stream));
  LaunchScope ls(ctx, "refcap");
  hipLaunchKernelGGL(refcap_mark_kernel, dim3(64, nframes), dim3(256), 0, ctx->stream, ctx->d_counters, ctx->d_cand,
                     (unsigned)ctx->cand_cap, octave, tiles_x, tiles_y, ctx->d_refcap, words);
  hipLaunchKernelGGL(refcap_drop_kernel, dim3(64, nframes), dim3(256), 0, ctx->stream, ctx->d_counters, ctx->d_cand,
                     (unsigned)ctx->cand_cap, octave, tiles_x, tiles_y, ctx->d_refcap, words);
  return ls.finish();
}

int launch_dog_scan(misift_ctx *ctx, const float *base, const StripGeom &g, const LaplaceTaps &taps,
                      float thresh, int octave)
{
  const int al0 = is_aligned16(base, g.pitch) && (g.frame_stride & 3) == 0;
  LaunchScope ls(ctx, "dog_scan");
  // <FAST, 2>: two wavefronts per SIMD (<= 256 VGPRs) measured 1.29x faster than one
  if (al0 && (g.width & 3) == 0)
    hipLaunchKernelGGL((dog_scan_kernel<true, 2>), grid_for(g), dim3(256), 0, ctx->stream, base, g, taps, thresh,
                       octave, ctx->d_counters, ctx->d_cand, (unsigned)ctx->cand_cap, al0);
  else
    hipLaunchKernelGGL((dog_scan_kernel<false, 2>), grid_for(g), dim3(256), 0, ctx->stream, base, g, taps, thresh,
                       octave, ctx->d_counters, ctx->d_cand, (unsigned)ctx->cand_cap, al0);
  return ls.finish();
}

int launch_refine(misift_ctx *ctx, const float *dog, long long dog_frame_stride, const float *base,
                  long long base_frame_stride, const LaplaceTaps *taps, int w, int h, int pitch, int nframes,
                  float thresh, float edge_limit, float factor, float lowest_scale, float subsampling, int octave,
                  SiftPointD *pts, int max_pts)
{
  RefineParams P;
  P.width = w; P.height = h; P.pitch = pitch; P.nframes = nframes;
  P.edge_limit = edge_limit; P.factor = factor; P.lowest_scale = lowest_scale; P.subsampling = subsampling;
  P.thresh = thresh;
  for (int s = 0; s < NUM_SCALES; s++) P.scmul[s] = powf(2.0f, (float)s / NUM_SCALES);
  P.octave = octave; P.max_pts = max_pts; P.cand_cap = (unsigned)ctx->cand_cap;
  LaplaceTaps t;
  if (taps) t = *taps; else memset(&t, 0, sizeof(t));
  const dim3 grid(128, nframes);
  LaunchScope ls(ctx, "refine");
  if (dog)
    hipLaunchKernelGGL(refine_kernel<false>, grid, dim3(64), 0, ctx->stream, dog, dog_frame_stride, t, P,
                       ctx->d_counters, ctx->d_cand, pts);
  else
    hipLaunchKernelGGL(refine_kernel<true>, grid, dim3(64), 0, ctx->stream, base, base_frame_stride, t, P,
                       ctx->d_counters, ctx->d_cand, pts);
  return ls.finish();
}

// ---- merged-octave launchers -------------------------------------------------------------------
static AllTaps pack_taps(const LaplaceTaps *taps, int noct)
{
  AllTaps a;
  memset(&a, 0, sizeof(a));
  for (int o = 1; o <= noct; o++) a.t[o] = taps[o];
  return a;
}

// options.reference_cap on the fused path: layout of the per-block extremum counters (bytes, four to a word) and their buffer
struct CapLayout { unsigned words; unsigned off[MISIFT_MAX_OCTAVES + 1]; int tx[MISIFT_MAX_OCTAVES + 1], ty[MISIFT_MAX_OCTAVES + 1]; };
static CapLayout cap_layout(const PyramidInfo &P)
{
  CapLayout L;
  memset(&L, 0, sizeof(L));
  unsigned n = 0;
  for (int o = 1; o <= P.noct; o++) {
    L.off[o] = n;
    L.tx[o] = (P.o[o].w + REFCAP_W - 1) / REFCAP_W;
    L.ty[o] = (P.o[o].h + REFCAP_H - 1) / REFCAP_H;
    n += (unsigned)NUM_SCALES * L.tx[o] * L.ty[o];
  }
  L.words = (n + 3u) / 4u;
  return L;
}
static int ensure_capcnt(misift_ctx *ctx, const PyramidInfo &P, const CapLayout &L)
{
  const size_t bytes = sizeof(unsigned) * (size_t)L.words * P.nframes;
  if (bytes > ctx->capcnt_bytes) {
    if (ctx->d_capcnt) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(misift_dev_free(ctx->d_capcnt)); }
    ctx->d_capcnt = nullptr; ctx->capcnt_bytes = 0;
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_capcnt, bytes, "refcap_counts"));
    ctx->capcnt_bytes = bytes;
    ctx->alloc_gen++;
  }
  return MISIFT_OK;
}

// lev_begin..lev_end: pyramid levels to scan, 0 = finest (all levels: 0, P.noct)
int launch_dog_scan_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, const LaplaceTaps *taps,
                        float thresh, int lev_begin, int lev_end, const ChainGeom *chain, const float *k5)
{
  ScanAllGeom G;
  memset(&G, 0, sizeof(G));
  G.nlev = lev_end - lev_begin; G.nframes = P.nframes; G.frame_stride = P.frame_stride;
  bool fast = (is_aligned16(scratch, 4) && (P.frame_stride & 3) == 0), ragged = false;
  long long items = 0;
  unsigned cand_stride = 0;
  for (int o = 1; o <= P.noct; o++) cand_stride += P.o[o].cand_cap;
  G.cand_stride = cand_stride;
  // total wavefronts aimed at: ~16 per CU; every level gets segments sized for its share
  for (int lev = lev_begin; lev < lev_end; lev++) {
    const int o = P.noct - lev;                           // finest first
    const OctaveInfo &L = P.o[o];
    ScanOct &S = G.o[lev - lev_begin];
    S.w = L.w; S.h = L.h; S.p = L.p; S.octave = o; S.img_off = L.img_off;
    S.cand_off = L.cand_off; S.cand_cap = L.cand_cap;
    const int nquads = (L.w + 3) / 4;
    S.nstrips = (nquads + (OUT_LANES - 2) - 1) / (OUT_LANES - 2);
    if (S.nstrips < 1) S.nstrips = 1;
    const long long target = (long long)ctx->num_cus * ctx->scan_waves_per_cu;
    long long want = (target + (long long)P.nframes * S.nstrips - 1) / ((long long)P.nframes * S.nstrips);
    if (want < 1) want = 1;
    int seg = (int)((L.h + want - 1) / want);
#if SCAN_RING || SCAN_UNROLL9
    if (P.nframes <= ctx->small_frames) {
      // a frame or two: the launch is latency-bound (rows per wavefront x ~1 us), so short segments on every SIMD —
      // shorter still on the levels behind the embedded ScaleDown chain, whose rows are the tail of the launch's
      // critical path (chain -> coarse rows)
      const int floor_rows = (chain && lev - lev_begin >= 2) ? ctx->scan_rows_small_coarse : ctx->scan_rows_small;
      if (seg < floor_rows) seg = floor_rows;
      if (seg > RING_ROWS) seg = (seg + RING_ROWS - 1) / RING_ROWS * RING_ROWS;
    } else {
      seg = (seg + RING_ROWS - 1) / RING_ROWS * RING_ROWS;     // whole turns of the nine-row ring
      if (seg < 2 * RING_ROWS) seg = 2 * RING_ROWS;
    }
    if (seg > 14 * RING_ROWS) seg = 14 * RING_ROWS;
#else
    seg = (seg + 7) / 8 * 8;
    if (seg < 16) seg = 16;
    if (seg > 128) seg = 128;
#endif
    S.seg_rows = seg;
    S.nsegs = (L.h + seg - 1) / seg;
    S.item_begin = items;
    items += (long long)P.nframes * S.nstrips * S.nsegs;
    if ((L.p & 3) != 0 || (L.img_off & 3) != 0) fast = false;
    if ((L.w & 3) != 0) ragged = true;          // some level's width is not a multiple of 4 (e.g. 1000 -> 250 -> 125)
  }
  G.total_items = items;
  if (ctx->opt.reference_cap && lev_begin == 0) {        // the launch that holds the finest level clears refine_all's counters
    const CapLayout L = cap_layout(P);
    const int rc = ensure_capcnt(ctx, P, L);
    if (rc) return rc;
    G.cap_clear = ctx->d_capcnt;
    G.cap_clear_words = L.words * (unsigned)P.nframes;
  }
  const AllTaps at = pack_taps(taps, P.noct);
  ChainGeom C;
  Taps5 t5;
  memset(&C, 0, sizeof(C));
  memset(&t5, 0, sizeof(t5));
  int nchain = 0;
  if (chain) {
    // the chain's outputs are the levels below its source: scan items of those levels wait for it
    C = *chain;
    for (int j = 0; j < 5; j++) t5.k[j] = k5[j];
    nchain = C.tiles_x * C.tiles_y * P.nframes;
    G.wait_lev = 0;                                        // (source not among the scanned levels: everything waits)
    G.wait_ticks = ctx->chain_wait_ticks;
    for (int k = 0; k < G.nlev; k++)
      if (G.o[k].img_off == C.lv[0].off) G.wait_lev = k + 1;
  }
  const dim3 grid((unsigned)((items + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK) + (unsigned)nchain);
  LaunchScope ls(ctx, "dog_scan");
#define SCAN_LAUNCH(F, CH) hipLaunchKernelGGL((dog_scan_all_kernel<F, CH>), grid, dim3(256), (size_t)ctx->lds_pad_scan, ctx->stream, scratch, G, at, \
                                              thresh, ctx->d_counters, ctx->d_cand, C, t5, nchain)
  if (chain) {
    if (fast && !ragged) SCAN_LAUNCH(1, true);
    else if (fast) SCAN_LAUNCH(2, true);
    else SCAN_LAUNCH(0, true);
  } else {
    if (fast && !ragged) SCAN_LAUNCH(1, false);
    else if (fast) SCAN_LAUNCH(2, false);        // pyramid levels are ours (pitch % 128 == 0): ragged widths keep the dwordx4 row loads (r03)
    else SCAN_LAUNCH(0, false);
  }
#undef SCAN_LAUNCH
  return ls.finish();
}

int launch_refine_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, const LaplaceTaps *taps,
                      float thresh, float edge_limit, float factor, int max_pts)
{
  RefineAllParams R;
  R.thresh = thresh; R.edge_limit = edge_limit; R.factor = factor; R.max_pts = max_pts;
  for (int s = 0; s < NUM_SCALES; s++) R.scmul[s] = powf(2.0f, (float)s / NUM_SCALES);
  unsigned cand_stride = 0;
  for (int o = 1; o <= P.noct; o++) cand_stride += P.o[o].cand_cap;
  R.cand_stride = cand_stride;
  R.cap_words = 0;
  if (ctx->opt.reference_cap) {                    // (cleared by the scan launch that ran before: launch_dog_scan_all)
    const CapLayout L = cap_layout(P);
    const int rc = ensure_capcnt(ctx, P, L);
    if (rc) return rc;
    for (int o = 1; o <= P.noct; o++) { R.cap_off[o] = L.off[o]; R.cap_tx[o] = L.tx[o]; R.cap_ty[o] = L.ty[o]; }
    R.cap_words = L.words;
    R.cap_limit = (unsigned)ctx->refcap_limit;
  }
  const AllTaps at = pack_taps(taps, P.noct);
  LaunchScope ls(ctx, "refine");
  // 16 candidates per workgroup and round: a batch keeps 64 workgroups per frame busy for many rounds; a single frame
  // wants its ~10 k candidates done in one (every round is a dependent chain of 11 loads)
  const int gx = P.nframes <= ctx->small_frames ? 1024 / P.nframes : 64;
  hipLaunchKernelGGL(refine_all_kernel, dim3(gx, P.nframes), dim3(256), 0, ctx->stream, scratch, P, at, R,
                     ctx->d_counters, ctx->d_cand, ctx->d_det, ctx->d_capcnt);
  return ls.finish();
}

// ---- test-only: the device det_exp2 on caller-supplied inputs (misift_test_elementary; tests compare the bits with the
// oracle's and the values with float64 libm)
__global__ void test_exp2_kernel(const float *__restrict__ x, float *__restrict__ out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = det_exp2(x[i]);
}
int launch_test_exp2(misift_ctx *ctx, const float *x, float *out, int n)
{
  hipLaunchKernelGGL(test_exp2_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, x, out, n);
  HIP_TRY(hipGetLastError());
  return MISIFT_OK;
}

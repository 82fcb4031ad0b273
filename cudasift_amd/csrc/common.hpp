// common.hpp — shared definitions of libmisift.so (gfx950 only).
//
// Streaming-kernel vocabulary used throughout:
//   quad     : 4 horizontally adjacent pixels held by one lane as a float4
//   strip    : the 64 quads (256 px) one wavefront owns; lanes 0 and 63 are
//              halo lanes for the radius-4 filters, so a strip emits 62 quads
//   segment  : the run of image rows one wavefront walks down, keeping its
//              filter windows in VGPRs (no LDS staging of image tiles needed:
//              R = 4 px is exactly one quad, so neighbours are a DPP shift away)
//   item     : (frame, strip, segment) — one wavefront's unit of work
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "misift.h"

#define NUM_SCALES 5
#define NUM_BLURS  8          // NUM_SCALES + 3 (cudaSiftD.h:33 LAPLACE_S)
#define NUM_DOG    7
#define CNT_STRIDE 64         // uints per frame in the counter block
#define CNT_MAXPTS 17         // slot holding maxPts (reference d_MaxNumPoints)
#define CNT_CAND   20         // CNT_CAND + octave : candidate count of that octave
#define CNT_CANDOVF 28        // candidates dropped because the list was full
#define CNT_PTOVF  29         // points dropped because maxPts was reached
#define CNT_CHAINTMO 30       // frame 0's block only: workgroups whose bounded wait for the embedded chain expired (dog_scan_all_kernel)
#define CNT_DET    32         // CNT_DET + octave : detections of that octave (merged-octave pipeline)
#define CNT_DUP    40         // CNT_DUP + octave : second-orientation duplicates of that octave
#define CNT_SPARE_BLOCKS 9    // counter blocks behind the last frame's: flags and ticket words of the call (dog_scan_all_kernel)
#define CNT_BIG    48         // keypoints deferred to descr_big_kernel (descriptor window larger than descr_all's LDS tile)
#define CNT_TICKET 49         // frame 0's block only: workgroups of the last kernel that have finished (host export)
#define CNT_ORIDONE 50        // CNT_ORIDONE + octave : orientations finished (orient_descr_fused_kernel: the octaves coarser than the finest)
#define CNT_FUSETMO 31        // frame 0's block only: workgroups of the fused kernel whose bounded wait for the coarser octaves' orientations expired

struct alignas(16) SiftPointD {   // device view of the 576-byte record
  float xpos, ypos, scale, sharpness, edgeness, orientation, score, ambiguity;
  int   match;
  float match_xpos, match_ypos, match_error, subsampling;
  float empty[3];
  float data[128];
};
static_assert(sizeof(SiftPointD) == MISIFT_POINT_BYTES, "record size");

struct Taps5 { float k[5]; };                    // k[0] = centre tap
struct LaplaceTaps { float k[NUM_BLURS][5]; };   // per blur scale, k[.][0] = centre

// Merged-octave pipeline: one pyramid level as seen by the per-keypoint kernels.  Index = reference
// octave number (1 = coarsest ... noct = finest); img_off = float offset of the level inside a frame's arena.
struct OctaveInfo {
  int w, h, p;
  long long img_off;
  float subsampling, lowest_scale;
  unsigned cand_off, cand_cap;     // this octave's slice of the frame's candidate list
};
struct PyramidInfo {
  int noct, nframes;
  float out_scale;                 // 0.5 when the frames were up-sampled first (RescalePositions, cudaSiftH.cu:130), else 1
  int fix_numpts;                  // options.fix_numpts: the finest octave's second orientations lie INSIDE numPts (and are rescaled)
  float patch_reach;               // descr_all: keypoints whose samples reach further than this many texels go to descr_big
                                   // (17.9 = what the 40 x 40 LDS window covers; MISIFT_TEST_PATCH_REACH lowers it so that tests
                                   //  can send ordinary keypoints down the rare path)
  long long frame_stride;          // floats between frames' arenas
  OctaveInfo o[MISIFT_MAX_OCTAVES + 1];
};
// A detection waiting in the staging area (32 bytes): refine fills the first five fields, orient the rest.
struct alignas(16) Detection {
  float xpos, ypos, scale, sharpness, edgeness, ori1, ori2;
  int dupslot;                     // slot among this octave's duplicates, -1 = no second orientation
};

// Geometry of one batched streaming launch.
struct StripGeom {
  int width, height, pitch;        // source image
  int nstrips, nsegs, seg_rows;    // decomposition
  int nframes;
  long long frame_stride;          // floats between frames of the source
  int noremap;                     // 1 = keep the hardware's block order (no XCD remap)
};

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Lane i receives lane i-1's value (lane 0 receives 0: bound_ctrl): DPP wave_shr:1.
__device__ __forceinline__ float lane_from_left(float x)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
// Lane i receives lane i+1's value (lane 63 receives 0): DPP wave_shl:1.
__device__ __forceinline__ float lane_from_right(float x)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float4 quad_from_left(float4 v)
{
  return make_float4(lane_from_left(v.x), lane_from_left(v.y), lane_from_left(v.z), lane_from_left(v.w));
}
__device__ __forceinline__ float4 quad_from_right(float4 v)
{
  return make_float4(lane_from_right(v.x), lane_from_right(v.y), lane_from_right(v.z), lane_from_right(v.w));
}

// The dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md).  Remap so
// that the blocks sharing an XCD (and its private 4 MiB L2) are consecutive
// logical blocks, i.e. spatial neighbours.  Bijective for any nb; speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb)
{
  const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u, i = b >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + i;
}

// The same dot product where the reference writes it as ONE expression k4*c + k3*p1 + ... (LowPassBlock,
// cudaSiftD.cu:2001-2005, :2022-2026): a contracting compiler fuses the LEFT product of `a*b + c*d` and rounds the
// right one (oracle/sift_oracle.c conv9_expr; bit-identical to the reference source built with -ffp-contract=fast).
__device__ __forceinline__ float conv9_expr(const float k0, const float k1, const float k2, const float k3,
                                            const float k4, float c, float p1, float p2, float p3, float p4)
{
  float s = __builtin_fmaf(k0, c, k1 * p1);
  s = __builtin_fmaf(k2, p2, s);
  s = __builtin_fmaf(k3, p3, s);
  s = __builtin_fmaf(k4, p4, s);
  return s;
}

// Symmetric 9-tap dot product: centre tap first, then outward (explicit fmaf chain;
// arithmetic contract shared with oracle/sift_oracle.c conv9()).
__device__ __forceinline__ float conv9(const float k0, const float k1, const float k2, const float k3,
                                       const float k4, float c, float p1, float p2, float p3, float p4)
{
  float s = k0 * c;
  s = __builtin_fmaf(k1, p1, s);
  s = __builtin_fmaf(k2, p2, s);
  s = __builtin_fmaf(k3, p3, s);
  s = __builtin_fmaf(k4, p4, s);
  return s;
}

// Load the quad of pixels [4q, 4q+3] of one image row with clamp-to-edge columns.
// `aligned` rows (pitch % 4 == 0, base 16-byte aligned) use one dwordx4 load in the interior.
__device__ __forceinline__ float4 load_quad(const float *row, int q, int width, bool aligned)
{
  const int x = 4 * q;
  if (aligned && x >= 0 && x + 3 < width) return *reinterpret_cast<const float4 *>(row + x);
  const int w1 = width - 1;
  return make_float4(row[clampi(x, 0, w1)], row[clampi(x + 1, 0, w1)], row[clampi(x + 2, 0, w1)],
                     row[clampi(x + 3, 0, w1)]);
}

// Fast-path column addressing for images whose width is a multiple of 4 and whose rows are 16-byte
// aligned (every pyramid level of the standard sizes): the lane's quad index is clamped once, every
// row is ONE unconditional dwordx4 load, and the clamp-to-edge semantics for the halo quads left /
// right of the image are restored with selects (splat of column 0 / column width-1) — no divergent
// branches in the streaming loops.
// Wave votes on a bool: one v_cmp + s_and (+ s_cmp) — HIP's __any(int) goes through v_cndmask 0/1 + v_cmp_ne first.
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

struct QuadCol {
  int off;      // float offset of the clamped quad within a row
  int edge;     // -1: quad lies left of the image, +1: right of it, 0: inside;
                // ragged widths (width % 4 != 0, MODE 2): 2 = the quad that holds the last column, 3 = right of it
  int rem;      // width % 4 (MODE 2)
  int wave_has_edge;   // wave-uniform: some lane of this wavefront holds a quad outside / on the ragged edge of the image.
                       // Only the first and the last strip of a row do; everywhere else the eight selects of the
                       // clamp are skipped by a scalar branch (r03: they were 9 of the 130 VALU of a lowpass_down row)
};
__device__ __forceinline__ QuadCol make_quadcol(int q, int width)
{
  const int nq = width >> 2;                  // full quads
  const int rem = width & 3;
  QuadCol c;
  c.rem = rem;
  if (rem == 0) {
    c.edge = q < 0 ? -1 : (q > nq - 1 ? 1 : 0);
    c.off = 4 * clampi(q, 0, nq - 1);
  } else {                                    // the partial quad nq exists; quads right of it re-load it
    c.edge = q < 0 ? -1 : (q == nq ? 2 : (q > nq ? 3 : 0));
    c.off = 4 * clampi(q, 0, nq);
  }
  c.wave_has_edge = __builtin_amdgcn_readfirstlane(__any(c.edge != 0) ? 1 : 0);
  return c;
}
// clamp-to-edge of a quad loaded at a clamped position.  MODE 1: width % 4 == 0.  MODE 2: any width — the partial quad
// is read as a whole dwordx4 (the up-to-3 floats past the last column lie inside the row pitch: the launchers check
// pitch >= roundup4(width)) and its invalid elements, like every quad right of it, take the value of column width-1.
template <int MODE>
__device__ __forceinline__ float4 clamp_quad(float4 v, const QuadCol &c)
{
  if (c.edge < 0) v = make_float4(v.x, v.x, v.x, v.x);
  if (MODE == 2) {
    if (c.edge == 1) v = make_float4(v.w, v.w, v.w, v.w);
    if (c.edge >= 2) {
      const float e = c.rem == 1 ? v.x : (c.rem == 2 ? v.y : v.z);
      if (c.edge == 3 || c.rem < 2) v.y = e;
      if (c.edge == 3 || c.rem < 3) v.z = e;
      v.w = e;
      if (c.edge == 3) v.x = e;
    }
  } else {
    if (c.edge > 0) v = make_float4(v.w, v.w, v.w, v.w);
  }
  return v;
}
// MODE: 0 = generic (scalar edge loads, any alignment), 1 = fast (bool true), 2 = fast with ragged widths
template <int MODE>
__device__ __forceinline__ float4 load_quad_t(const float *row, int q, int width, bool aligned, const QuadCol &c)
{
  if (MODE) {
    float4 v = *reinterpret_cast<const float4 *>(row + c.off);
    if (c.wave_has_edge) v = clamp_quad<MODE>(v, c);
    return v;
  }
  return load_quad(row, q, width, aligned);
}

// 8-bit source rows (misift_extract_batch_u8: camera frames uploaded as bytes, §8f-2): one dword load per
// lane, v_cvt_f32_ubyte0..3 — exact, so the result equals an fp32 upload of the same pixel values.
__device__ __forceinline__ float4 load_quad(const unsigned char *row, int q, int width, bool aligned)
{
  const int x = 4 * q;
  if (aligned && x >= 0 && x + 3 < width) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(row + x);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  }
  const int w1 = width - 1;
  return make_float4((float)row[clampi(x, 0, w1)], (float)row[clampi(x + 1, 0, w1)], (float)row[clampi(x + 2, 0, w1)],
                     (float)row[clampi(x + 3, 0, w1)]);
}
template <int MODE>
__device__ __forceinline__ float4 load_quad_t(const unsigned char *row, int q, int width, bool aligned, const QuadCol &c)
{
  if (MODE) {
    const uchar4 u = *reinterpret_cast<const uchar4 *>(row + c.off);
    float4 v = make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
    if (c.wave_has_edge) v = clamp_quad<MODE>(v, c);
    return v;
  }
  return load_quad(row, q, width, aligned);
}

struct __attribute__((packed, aligned(4))) Pair2 { float a, b; };

// tex2D<float>() of a pitch2D texture with clamp addressing and linear filtering
// (cudaSiftH.cu:196-205).  frac8: round the weights to 8 fractional bits like the
// CUDA texture unit.  Same operation sequence as oracle tex2d().
// INTERIOR: the caller guarantees 1 <= x - 0.5 and x + 0.5 <= w - 2 (same for y), i.e. all four texels lie
// inside the image: no clamping, no edge selects (same values, ~20 VALU instructions less per fetch).
template <bool INTERIOR = false>
__device__ __forceinline__ float tex2d(const float *img, int w, int h, int pitch, float x, float y, bool frac8)
{
  float xb = x - 0.5f, yb = y - 0.5f;
  float fx = 0.0f, fy = 0.0f, a, b;
  int ix = 0, iy = 0;
  if (INTERIOR) {
    // coordinates >= 1: v_fract_f32 is xb - floor(xb) exactly, and the floor feeds only the integer conversion
    // (v_cvt_flr_i32_f32): two instructions per coordinate instead of floor + subtract + convert
    a = __builtin_amdgcn_fractf(xb); b = __builtin_amdgcn_fractf(yb);
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(xb));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(yb));
  } else {
    fx = floorf(xb); fy = floorf(yb);
    a = xb - fx; b = yb - fy;
  }
  if (frac8) {
    // 256 x the oracle's weights rintf(a * 256) / 256: adding and subtracting 1.5 * 2^23 rounds to the nearest integer,
    // ties to even, in two full-rate instructions (floor-based rounding: three, one of them quarter rate).  The blend
    // below runs on the 256 x weights and is scaled by 2^-16 at the end: powers of two, so every rounding is the same.
    a = __builtin_fmaf(a, 256.0f, 12582912.0f) - 12582912.0f;
    b = __builtin_fmaf(b, 256.0f, 12582912.0f) - 12582912.0f;
  }
  float t00, t10, t01, t11;
  // Addresses are 32-bit BYTE offsets from the level's base pointer (wave-uniform: an SGPR pair), so each texel-pair
  // load is `global_load_dwordx2 v, voffset, s[base]` after ONE v_mad_u32_u24 + shift — the 64-bit form the compiler
  // derives from size_t indexing costs two v_mad_i64_i32 and five 64-bit shifts/adds per fetch (r02: the per-keypoint
  // kernels are VALU-issue-bound, and this was a third of their sampling loop).  A pyramid level is < 2^31 bytes.
  const char *base = reinterpret_cast<const char *>(img);
  if (INTERIOR) {
    const unsigned off = (__umul24((unsigned)iy, (unsigned)pitch) + (unsigned)ix) * 4u;      // v_mad_u32_u24: rows, pitch < 2^24
    const Pair2 r0 = *reinterpret_cast<const Pair2 *>(base + off);
    const Pair2 r1 = *reinterpret_cast<const Pair2 *>(base + (off + (unsigned)pitch * 4u));
    t00 = r0.a; t10 = r0.b; t01 = r1.a; t11 = r1.b;
  } else {
    fx = fminf(fmaxf(fx, -2.0f), (float)w);
    fy = fminf(fmaxf(fy, -2.0f), (float)h);
    ix = (int)fx; iy = (int)fy;
    const int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
    const int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
    // The two texels of a row are adjacent except at the clamped image edges: fetch them with ONE 8-byte load at
    // column xl (4-byte aligned is enough on gfx950) and pick.
    const int xl = clampi(ix, 0, w > 1 ? w - 2 : 0);        // (a 1-px-wide level: both texels are column 0)
    const Pair2 r0 = *reinterpret_cast<const Pair2 *>(base + (__umul24((unsigned)y0, (unsigned)pitch) + (unsigned)xl) * 4u);
    const Pair2 r1 = *reinterpret_cast<const Pair2 *>(base + (__umul24((unsigned)y1, (unsigned)pitch) + (unsigned)xl) * 4u);
    const bool lo0 = x0 == xl, hi1 = x1 == xl + 1;
    t00 = lo0 ? r0.a : r0.b; t10 = hi1 ? r0.b : r0.a;
    t01 = lo0 ? r1.a : r1.b; t11 = hi1 ? r1.b : r1.a;
  }
  const float one = frac8 ? 256.0f : 1.0f;
  const float ia = one - a, ib = one - b;
  float v = (ia * ib) * t00;
  v = __builtin_fmaf(a * ib, t10, v);
  v = __builtin_fmaf(ia * b, t01, v);
  v = __builtin_fmaf(a * b, t11, v);
  if (frac8) v *= 1.0f / 65536.0f;
  return v;
}

#endif  // __HIPCC__

// ---------------------------------------------------------------- host side
struct ProfEntry {
  char name[32];
  float total_ms;
  int calls;
};

// A folded single call (launch_descr_all): what descr_big_kernel needs should the host find a deferred keypoint afterwards.
struct PendingBig {
  int valid;
  const float *scratch;
  PyramidInfo P;
  const Detection *det;
  SiftPointD *pts;
  int max_pts;
  const int *pack_offsets;
  SiftPointD *pack_dst;
  unsigned big_stride;
};

struct misift_ctx {
  int device;
  hipStream_t stream;
  misift_options opt;
  unsigned int *d_counters;     // [cap_frames][CNT_STRIDE]
  unsigned int *h_counters;     // pinned mirror
  int cap_frames;
  unsigned int *d_cand;         // candidate lists [cap_frames][cand_cap]
  size_t cand_cap;              // entries per frame
  Detection *d_det;             // staging [cap_det_frames][MISIFT_MAX_OCTAVES][det_max_pts] (refine's append order)
  Detection *d_det_sorted;      // the same records binned by 32x32-px tile (what orient_all / descr_all consume)
  int descr_occ;                // descr_all_kernel variant: registers held to 3 or 4 waves/SIMD (MISIFT_DESCR_OCC)
  int tile_descr, tile_orient;  // 1 = LDS-staged window in descr_all / orient_all, 0 = bilinear fetches from global memory
                                // (MISIFT_TILE_DESCR / MISIFT_TILE_ORIENT; MISIFT_TILE sets both)
  int bin_detections;           // 1 = run bin_detections_kernel between refine_all and orient_all (MISIFT_BIN=0 disables)
  int cap_det_frames, det_max_pts;
  float *d_own_scratch;         // scratch allocated on behalf of the caller (NULL tempMemory)
  size_t own_scratch_floats;
  // developer knobs: bytes of dynamic LDS added to a launch, i.e. a cap on the workgroups of that kernel a CU holds at once
  // (160 KB per CU: > 53.4 KB -> 2, > 40 KB -> 3), leaving wave slots and registers to the kernels of the OTHER batches in
  // flight (MISIFT_LDS_PAD_LPD / _SCAN / _ORIENT / _DESCR; 0 = none)
  int lds_pad_lpd, lds_pad_scan, lds_pad_orient, lds_pad_descr;
  float patch_reach;            // PyramidInfo.patch_reach of this context's calls
  int fold_descr_tail;          // 1 = single calls: descr_all's last workgroup exports the counters, descr_big only on demand (MISIFT_FOLD_TAIL)
  int descr_big_fallbacks;      // folded calls that needed descr_big after all (misift_ctx_descr_big_fallbacks)
  PendingBig pending_big;
  unsigned *d_refcap;           // options.reference_cap: 240-bit extremum masks of the reference's 30 x 8 blocks (launch_refcap)
  size_t refcap_bytes;
  unsigned *d_capcnt;           // ... on the fused path: byte counters of true extrema per block (launch_refine_all)
  size_t capcnt_bytes;
  int refcap_limit;             // extrema a block may hold before its frame is redone with the cap applied (32; test knob)
  void *d_match_tmp;            // matcher partial results
  size_t match_tmp_bytes;
  int num_cus;
  // packed-output mode (misift_extract_batch_packed_async): descr_all writes the valid records of all frames
  // straight into one contiguous array; set only around that entry point's enqueue
  // pyramid tail (coarse ScaleDowns + their scan) on a second, high-priority stream beside the scan of the fine levels
  int split_tail, in_capture;
  hipStream_t stream2;
  hipEvent_t ev_fork, ev_join;
  int *pack_counts, *pack_offsets;
  SiftPointD *pack_dst;
  int orient_blocks_per_cu;
  int point_blocks_per_cu;                     // grid sizing of the per-keypoint kernels
  int strip_waves_per_cu, scan_waves_per_cu;   // segment sizing targets of the streaming kernels
  // small batches / the single-call path (r04): bound by dependent dispatches, so fewer and wider launches
  int chain_max_frames;         // <= this many frames: coarse ScaleDowns as chained launches (MISIFT_CHAIN_FRAMES)
  int chain_embed;              // 1 = that chain runs inside the scan launch when one chain covers all levels (MISIFT_CHAIN_EMBED)
  unsigned chain_wait_ticks;    // bound of the in-launch wait for that chain, 100 MHz ticks (MISIFT_CHAIN_WAIT_US; default 100 ms)
  int chain_fallbacks;          // calls re-run with a stand-alone chain launch because the bound expired (misift_ctx_chain_fallbacks)
  int fuse_orient;              // 1 = single calls: orientations and descriptors in ONE launch (orient_descr_fused_kernel; MISIFT_FUSE_ORIENT)
  unsigned fuse_wait_ticks;     // bound of its in-launch wait for the coarser octaves' orientations, 100 MHz ticks (MISIFT_FUSE_WAIT_US)
  int fuse_fallbacks;           // calls re-run with the two separate launches because that bound expired (misift_ctx_fuse_fallbacks)
  int bin_min_frames;           // >= this many frames: bin_detections runs (MISIFT_BIN_MIN_FRAMES)
  int small_frames;             // <= this many frames: short scan segments, wide refine / per-keypoint grids (MISIFT_SMALL_FRAMES)
  int lowpass_tile;             // 1 = such batches: the LDS-tiled prefilter + first ScaleDown (MISIFT_LOWPASS_TILE)
  int strip_rows_small;         // rows per prefilter / ScaleDown segment for such batches, even (MISIFT_STRIP_ROWS_SMALL)
  int scan_rows_small_coarse;   // ... and for the levels behind the embedded ScaleDown chain (MISIFT_SCAN_ROWS_SMALL_COARSE)
  int scan_rows_small;          // rows per scan segment for such batches (MISIFT_SCAN_ROWS_SMALL)
  int cur_binned;               // this call's per-keypoint kernels read d_det_sorted (set by misift_extract_enqueue)
  // Batches whose frames differ in keypoint count (r04, MISIFT_BALANCE=1): the workgroups of orient_all / descr_all are
  // dealt out in proportion to the frames' counts through a block -> (frame, sub-block, sub-blocks) table that
  // frame_shares_kernel writes behind refine_all; off: every frame gets the same number of workgroups
  int balance_frames;           // 1 = batches of more than small_frames frames use the table
  int cur_balanced;             // this call's block tables are built: by the extra workgroup of bin_detections, or by frame_shares_kernel (orient_all uses the first, descr_all the second)
  int4 *d_block_map;            // [map_t_orient | map_t_descr] entries
  int block_map_cap, map_t_orient, map_t_descr;
  int want_export, exported;    // host export of the counters by the last kernel: asked for by misift_extract_sync / done
  unsigned export_seq;          // sequence number the exporting kernel stores behind the counters
  int host_spin;                // 1 = poll the exported flag instead of hipStreamSynchronize (MISIFT_HOST_SPIN=0 disables)
  unsigned *d_flags, *h_flags;  // 64 words each: device ticket words / pinned host flags of synchronous calls (matcher)
  int want_match_flag, match_flagged;
  unsigned match_seq;
  int alloc_gen;                // bumped whenever a context-owned device buffer is reallocated (invalidates captured graphs)
  hipEvent_t ev0, ev1;
  // per-kernel profiling (HIP events on the context stream)
  bool profile;
  ProfEntry prof[32];
  int nprof;
  hipEvent_t pev[2];
};

void misift_set_error(const char *fmt, ...);
// Every device allocation of the library (its own buffers and misift_malloc's) goes through these two.  Normally they ARE
// hipMalloc / hipFree.  In guard mode (misift_test_set_guard, MISIFT_GUARD=1: the test build of SURVEY section 5's
// "guard-paged scratch arenas") an allocation gets MISIFT_GUARD_BYTES of a byte pattern in front of and behind the payload
// and the payload is pre-filled with 0xFF (NaN as float, -1 as int); misift_test_check_guards() verifies every band.
#define MISIFT_GUARD_BYTES 65536
hipError_t misift_dev_alloc(void **out, size_t bytes, const char *tag);
hipError_t misift_dev_free(void *ptr);
void misift_warn_hw_queues(const char *who);   // once per process: GPU_MAX_HW_QUEUES unset or < 8 (stderr + misift_last_error)
int misift_ensure_frames(misift_ctx *ctx, int nframes, size_t cand_cap);
// the stream the most recent batch of `ctx` ran on (its own, or a pipeline's with batches in flight)
hipStream_t misift_ctx_result_stream(misift_ctx *ctx);
int misift_ensure_tmp(misift_ctx *ctx, size_t bytes);
int launch_refcap(misift_ctx *ctx, int w, int h, int nframes, int octave);
int launch_descr_big_pending(misift_ctx *ctx);
bool misift_tiny_call(int width, int height, int num_octaves);
int misift_extract_enqueue(misift_ctx *ctx, const void *d_imgs, int src_u8, int nframes, long long frame_stride,
                           int width, int height, int pitch, int num_octaves, float init_blur, float thresh,
                           float lowest_scale, int scale_up, float *d_scratch, SiftPointD *pts, int max_pts);
// synchronous extraction of a batch incl. the exact dense re-run when a candidate list overflowed
int misift_extract_sync(misift_ctx *ctx, const void *d_imgs, int src_u8, int nframes, long long frame_stride, int width,
                        int height, int pitch, int num_octaves, float init_blur, float thresh, float lowest_scale,
                        int scale_up, float *d_scratch, SiftPointD *pts, int max_pts, int *num_pts_out);
// counts_out[f] = numPts of frame f (clamped to max_pts), or -1 when a candidate list of that frame overflowed;
// offsets_out (optional, nframes+1 entries): exclusive prefix sum of the non-negative counts
int launch_export_counts(misift_ctx *ctx, int nframes, int num_octaves, int max_pts, int *counts_out,
                         int *offsets_out);
int launch_pack_records(misift_ctx *ctx, const SiftPointD *pts, int max_pts, int nframes, const int *offsets,
                        SiftPointD *packed);

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      misift_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MISIFT_EHIP;                                                                    \
    }                                                                                        \
  } while (0)

// Launch bookkeeping: optional per-kernel event timing + launch error check.
struct LaunchScope {
  misift_ctx *ctx;
  const char *name;
  LaunchScope(misift_ctx *c, const char *n);
  int finish();   // returns MISIFT_OK / MISIFT_EHIP
};

// kernel launch wrappers (defined in the .hip files)
// zero_cnt (both prefilter launchers): the frames' counter blocks to clear, or nullptr
int launch_lowpass(misift_ctx *ctx, const void *src, int src_u8, const StripGeom &g, float *dst, int dpitch,
                   long long dst_frame_stride, const float k9[9], unsigned *zero_cnt);
int launch_lowpass_down(misift_ctx *ctx, const void *src, int src_u8, const StripGeom &g, float *dst, int dpitch,
                        long long dst_frame_stride, const float k9[9], float *dst2, int dpitch2,
                        long long dst2_frame_stride, const float k5[5], int *done, unsigned *zero_cnt);
int launch_lowpass_down_tile(misift_ctx *ctx, const void *src, int src_u8, int width, int height, int spitch,
                             long long src_frame_stride, int nframes, float *dst, int dpitch, long long dst_frame_stride,
                             const float k9[9], float *dst2, int dpitch2, long long dst2_frame_stride, const float k5[5],
                             unsigned *zero_cnt);
int launch_scaledown_chain(misift_ctx *ctx, float *scratch, long long frame_stride, int nframes, const int (*dims)[3],
                           const long long *offs, int nlev, const float k5[5]);
int launch_scaledown(misift_ctx *ctx, const float *src, const StripGeom &g, float *dst, int dpitch,
                     long long dst_frame_stride, const float k5[5]);
int launch_scaleup(misift_ctx *ctx, const void *src, int src_u8, int w, int h, int spitch, long long src_frame_stride,
                   int nframes, float *dst, int dpitch, long long dst_frame_stride);
int launch_laplace(misift_ctx *ctx, const float *base, const StripGeom &g, float *dog,
                   long long dog_frame_stride, const LaplaceTaps &taps);
int launch_detect(misift_ctx *ctx, const float *dog, const StripGeom &g, long long dog_frame_stride,
                  float thresh, int octave);
int launch_dog_scan(misift_ctx *ctx, const float *base, const StripGeom &g, const LaplaceTaps &taps,
                      float thresh, int octave);
int launch_refine(misift_ctx *ctx, const float *dog, long long dog_frame_stride, const float *base,
                  long long base_frame_stride, const LaplaceTaps *taps, int w, int h, int pitch, int nframes,
                  float thresh, float edge_limit, float factor, float lowest_scale, float subsampling, int octave,
                  SiftPointD *pts, int max_pts);
int launch_orient(misift_ctx *ctx, const float *base, long long base_frame_stride, int w, int h, int pitch,
                  int nframes, int octave, SiftPointD *pts, int max_pts);
int launch_descr(misift_ctx *ctx, const float *base, long long base_frame_stride, int w, int h, int pitch,
                 int nframes, float subsampling, int octave, SiftPointD *pts, int max_pts);
int launch_rescale(misift_ctx *ctx, SiftPointD *pts, int npts, float scale);
int launch_rescale_batch(misift_ctx *ctx, SiftPointD *pts, int max_pts, int nframes, int num_octaves, float scale);
int launch_sort_segments(misift_ctx *ctx, SiftPointD *pts, int max_pts, int nframes, int num_octaves);
int launch_bin_detections(misift_ctx *ctx, const PyramidInfo &P, int max_pts);
int launch_renumber_dups(misift_ctx *ctx, const PyramidInfo &P, int max_pts);
int launch_orient_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts);
int launch_orient_descr_fused(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts);
int launch_descr_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts,
                     const int *pack_offsets, SiftPointD *pack_dst);
// counts / offsets from the STAGED per-octave counters (valid once orient_all has run, before descr_all)
int launch_export_counts_staged(misift_ctx *ctx, int nframes, int num_octaves, int max_pts, int *counts_out,
                                int *offsets_out);
struct ScanAll;
struct ChainGeom;
// chain (optional): the ScaleDown chain that produces the coarse levels runs in the first workgroups of the same launch
int launch_dog_scan_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, const LaplaceTaps *taps,
                        float thresh, int lev_begin, int lev_end, const ChainGeom *chain = nullptr,
                        const float *k5 = nullptr);
int make_chain_geom(ChainGeom *G, long long frame_stride, const int (*dims)[3], const long long *offs, int nlev, int tile);
int launch_refine_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, const LaplaceTaps *taps,
                      float thresh, float edge_limit, float factor, int max_pts);
int launch_match(misift_ctx *ctx, SiftPointD *pts1, int row_begin, int row_count, const SiftPointD *pts2, int n2);
int launch_match_split(misift_ctx *ctx, SiftPointD *pts1, int row_begin, int row_count, const SiftPointD *pts2, int n2,
                       const SiftPointD *pts2_own, int own_t0, int own_t1, hipEvent_t rest_ready, int phase,
                       int packed2 = 0);     // packed2: set 2 is an array of MISIFT_MATCH_COLUMN_BYTES match columns, not records
enum { MATCH_PHASE_ALL = 0, MATCH_PHASE_OWN = 1, MATCH_PHASE_REST = 2 };   // both launches + merge / the own-shard launch / the rest + merge
int launch_test_exp2(misift_ctx *ctx, const float *x, float *out, int n);
int launch_test_points_fn(misift_ctx *ctx, int fn, const float *x, const float *y, float *out, float *out2, int n);
int launch_selftest(misift_ctx *ctx);

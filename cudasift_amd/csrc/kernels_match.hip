// kernels_match.hip — brute-force descriptor matching on the fp32 matrix cores of gfx950.
//
//   match_kernel + match_merge_kernel replace CleanMatches (reference matching.cu:289-293)
//   and FindMaxCorr10 (matching.cu:301-397), host MatchSiftData (matching.cu:1090-1206).
//
// The only dense contraction of the pipeline: S = D1 (n1 x 128) * D2^T (128 x n2), then a
// per-row (max, second max, argmax).  Each wavefront keeps its 32 rows of D1 for the whole
// K = 128 in VGPRs (64 registers) and sweeps 64-column super-tiles of D2 staged through
// double-buffered LDS as TWO interleaved accumulator chains, one v_mfma_f32_32x32x2_f32 per k-pair and chain with
// k ascending — so every
// score is bit-identical to the reference's sequential fp32 FMA chain (matching.cu:343-346;
// MI355X f32 MFMA == k-ordered fmaf chain).  The running top-2 is kept per lane, i.e. per
// column residue (p2 mod 32); residues 4c..4c+3 form the reference's "class" c = (p2 mod 32)/4
// (thread row iy of FindMaxCorr10), so the reference's lossy 8-class runner-up merge
// (matching.cu:375-390) can be reproduced exactly — or replaced by the exact second best.
// Columns are split into chunks across workgroups to fill 256 CUs; a small merge kernel
// combines the per-chunk class triples and writes score/match/ambiguity/match_xpos/ypos.
#include <stdlib.h>
#include <string.h>
#include "common.hpp"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef MT_WG_WAVES
#define MT_WG_WAVES 4              // wavefronts per workgroup (experiment: 8 = 256 rows share a staged tile, one workgroup per CU)
#endif
#define MT_ROWS_PER_BLOCK (32 * MT_WG_WAVES)
#define MT_STAGE (32 / MT_WG_WAVES)   // float4 loads per thread and super-tile
// LDS image of a 32-column tile of D2: column-major, each column split into its even-k and odd-k halves
// ([col][half][64]) so that lane (col, half) — which feeds k = 2t + half to MFMA t — reads its 64 operands
// as 16 contiguous ds_read_b128.  Column stride 132 floats: a b128 lane group (16 columns) then starts on
// 16 distinct multiples of 4 banks => conflict-free.
#define MT_BSTRIDE 132
#define MT_TILE 32

struct MatchGeom {
  int row_begin, row_count;      // rows of set 1 handled by this launch
  int n1_total;
  int n2, ncols;                 // set-2 size, columns that take part (32*floor(n2/32) or n2)
  int ntiles, nchunks, tiles_per_chunk;   // super-tiles / chunks of THIS launch
  // A launch sweeps a run of `ntiles` VIRTUAL super-tiles: virtual tile v is actual tile tile_base + v, plus hole_len
  // from v >= hole_begin on (the sharded matcher sweeps its own shard's tiles while the rest of set 2 is still on the
  // wire, then everything around them: multigpu.hip).  Chunk c of this launch is chunk chunk_base + c of nchunks_total.
  int tile_base, hole_begin, hole_len;
  int chunk_base, nchunks_total;
  // set-2 element layout, in floats: SiftPoint records (stride 144, descriptor at 16, xpos/ypos at 0) or the packed
  // match columns the sharded matcher ships (MISIFT_MATCH_COLUMN_BYTES = 528: stride 132, descriptor at 0, xy at 128)
  int stride2, data_off2, xy_off2;
};
__device__ __forceinline__ int tile_col0(const MatchGeom &G, int v)       // first column of virtual super-tile v
{
  return (G.tile_base + v + (v >= G.hole_begin ? G.hole_len : 0)) * 64;
}

// partial results: [row][class][chunk][3] : max, second, index — one contiguous run of chunks per (row, class), which is
// what a thread of match_merge_kernel reads
#define MT_PART_WORDS 24

__device__ __forceinline__ void top2_update(float sc, int p2, float &mx, float &sec, int &ix)
{
  // reference update rule (matching.cu:352-360): `if (sc > mx) {sec = mx; mx = sc; ix = p2;} else if (sc > sec)
  // sec = sc;` — strict '>' so the earliest column wins ties.  With sec <= mx always, the new (sec, mx) is the
  // top two of {sec, mx, sc}: sec' = median of the three (one v_med3_f32), mx' = sc or mx.  4 VALU per score
  // (compare, median, two selects) instead of 8 with fmaxf (which also costs canonicalising v_max x,x pairs).
  const bool gt = sc > mx;
  sec = __builtin_amdgcn_fmed3f(sec, mx, sc);
  ix = gt ? p2 : ix;
  mx = gt ? sc : mx;
}

// exact merge of two top-2 summaries of disjoint column sets (ties -> smaller column index,
// which is what one ascending scan over the union would produce)
// (selects, not branches: as an if/else every merge became two exec-mask regions behind its own operand wait, and the
//  32 merges at the end of match_kernel ran as one serial chain — 4.4 us of a 24 us launch, tools/match_stamps.py)
__device__ __forceinline__ void top2_merge(float &mx, float &sec, int &ix, float omx, float osec, int oix)
{
  const bool take = (omx > mx) | ((omx == mx) & (oix >= 0) & ((ix < 0) | (oix < ix)));
  const float sec_take = fmaxf(mx, osec), sec_keep = fmaxf(sec, omx);
  sec = take ? sec_take : sec_keep;
  mx = take ? omx : mx;
  ix = take ? oix : ix;
}
// lane ^ 1 / lane ^ 2 within a quad: one DPP move each (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E), no LDS round trip
template <int CTRL> __device__ __forceinline__ int quad_xchg(int v)
{
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float quad_xchg(float v)
{
  return __builtin_bit_cast(float, quad_xchg<CTRL>(__builtin_bit_cast(int, v)));
}

// One workgroup = 4 wavefronts = 128 rows of set 1; it sweeps a chunk of 64-column super-tiles of set 2.
// Per super-tile every wavefront runs TWO independent accumulator chains (columns 0-31 and 32-63) of
// 64 dependent v_mfma_f32_32x32x2_f32 each, interleaved, so the 64-cycle dependent-issue latency of one
// chain is covered by the other (and by the second wavefront resident on the SIMD).
#define MT_SUPER 64
#ifndef MT_A_SWAP
#define MT_A_SWAP 1
#endif
#ifndef MT_STAMPS
#define MT_STAMPS 0              // developer build (tools/variants.sh -DMT_STAMPS=1): 100 MHz time stamps, tools/match_stamps.py
#endif
#if MT_STAMPS
__device__ unsigned g_mt_stamp[16];
#define MT_STAMP_MAX(slot) do { if (threadIdx.x == 0) atomicMax(&g_mt_stamp[slot], (unsigned)wall_clock64()); } while (0)
#define MT_STAMP_MIN(slot) do { if (threadIdx.x == 0) atomicMin(&g_mt_stamp[slot], (unsigned)wall_clock64()); } while (0)
extern "C" int misift_debug_match_stamps(unsigned *out16)
{
  unsigned init[16];
  for (int i = 0; i < 16; i++) init[i] = (i == 0 || i == 8) ? 0xffffffffu : 0u;
  HIP_TRY(hipDeviceSynchronize());
  if (out16) HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mt_stamp), sizeof(init)));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_mt_stamp), init, sizeof(init)));
  return MISIFT_OK;
}
#else
#define MT_STAMP_MAX(slot) do { } while (0)
#define MT_STAMP_MIN(slot) do { } while (0)
#endif
__global__ __launch_bounds__(64 * MT_WG_WAVES, 8 / MT_WG_WAVES) void match_kernel(const SiftPointD *__restrict__ pts1,
                                                       const float *__restrict__ set2, MatchGeom G,
                                                       float *__restrict__ partial)
{
  __shared__ float Bs[2][MT_SUPER * MT_BSTRIDE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, col = lane & 31;
  const unsigned item = xcd_remap(blockIdx.x, gridDim.x);
  // chunk-major: the workgroups resident on one XCD work on the same chunk of set 2 (it stays in that L2)
  const int nrb = (G.row_count + MT_ROWS_PER_BLOCK - 1) / MT_ROWS_PER_BLOCK;
  const int chunk = item / nrb, rb = item % nrb;
  const int st0 = chunk * G.tiles_per_chunk;                       // super-tile range of this chunk
  const int st1 = min(st0 + G.tiles_per_chunk, G.ntiles);
  MT_STAMP_MIN(0);                 // first workgroup starts
  MT_STAMP_MAX(1);                 // last workgroup starts

  // ---- A fragment: row (lane&31) of this wave, k = 2t + half, t = 0..63
  const int row_local = rb * MT_ROWS_PER_BLOCK + wave * 32 + col;          // within [0,row_count)
  const int row_ld = G.row_begin + min(row_local, G.row_count - 1);
  float a[64];
  {
    const float4 *src = reinterpret_cast<const float4 *>(pts1[row_ld].data);
#if MT_A_SWAP
    // The two lanes of a row fetch adjacent float4s (k = 8i..8i+3 | 8i+4..8i+7) and trade the halves they do not need
    // with v_permlane32_swap (lanes 32-63 of the first operand <-> lanes 0-31 of the second): 16 loads per lane
    // instead of 32 of which half of every float4 was thrown away — the row fetch is the head of a small launch's
    // critical path and bound by the texture addresser (32 cache lines per instruction: rows are 576 bytes apart).
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float4 v = src[2 * i + half];
      const auto xy = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), false, false);
      const auto zw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w), false, false);
      a[4 * i + 0] = __builtin_bit_cast(float, (unsigned)xy[0]);       // k = 8i     | 8i + 1
      a[4 * i + 1] = __builtin_bit_cast(float, (unsigned)zw[0]);       // k = 8i + 2 | 8i + 3
      a[4 * i + 2] = __builtin_bit_cast(float, (unsigned)xy[1]);       // k = 8i + 4 | 8i + 5
      a[4 * i + 3] = __builtin_bit_cast(float, (unsigned)zw[1]);       // k = 8i + 6 | 8i + 7
    }
#else
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const float4 v = src[j];
      a[2 * j] = half ? v.y : v.x;
      a[2 * j + 1] = half ? v.w : v.z;
    }
#endif
  }
  // ---- per-lane running top-2 for 16 rows (accumulator register r <-> row (r&3)+8*(r>>2)+4*half)
  float mx[16], sec[16];
  int ix[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { mx[r] = 0.0f; sec[r] = 0.0f; ix[r] = -1; }

  // ---- B staging: thread -> (column scol + 8j, float4 index f4), j = 0..7
  const int scol = tid >> 5, f4 = tid & 31;
  float4 stage[MT_STAGE];
#ifndef MT_LEAN_STAGING
#define MT_LEAN_STAGING 1
#endif
#ifndef MT_TOP2_FILTER
#define MT_TOP2_FILTER 0
#endif
#if MT_LEAN_STAGING
  // r06: the staging costs the SIMD VALU time the matrix pipe does not get back (SQ counters: MFMA busy + VALU issuing ~ 1 of
  // the launch's SIMD-cycles).  (a) one 32-bit byte offset per thread and load (clamp + v_mad_u32_u24) against a wave-uniform
  // (SGPR) tile base instead of clamp + 64-bit multiply-add + 64-bit shift-add per load; (b) the even-k / odd-k halves of a staged float4 go to LDS as ds_write2_b32
  // x,z | y,w — two separate registers each — instead of ds_write2_b64 of register PAIRS the compiler has to assemble with
  // three v_mov per float4.  24 + 8 of the ~196 non-MFMA VALU instructions per super-tile and wavefront.
  const unsigned vconst = (unsigned)G.data_off2 * 4u + (unsigned)f4 * 16u;
  auto gload = [&](int st) {
    const int c0 = tile_col0(G, st);                                         // wave-uniform
    const char *sb = reinterpret_cast<const char *>(set2) + (size_t)c0 * (size_t)G.stride2 * 4u;
    const int last = G.n2 - 1 - c0;                                          // (scalar) the clamp matters in a partial last super-tile only
#pragma unroll
    for (int j = 0; j < MT_STAGE; j++) {
      const unsigned rel = (unsigned)min(scol + 2 * MT_WG_WAVES * j, last);
      stage[j] = *reinterpret_cast<const float4 *>(sb + (__umul24(rel, (unsigned)G.stride2 * 4u) + vconst));
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < MT_STAGE; j++) {
      float *d = &Bs[buf][(scol + 2 * MT_WG_WAVES * j) * MT_BSTRIDE + 2 * f4];       // k = 4*f4 .. 4*f4+3
      const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) float *)d;
      asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(a), "v"(stage[j].x), "v"(stage[j].z) : "memory");            // even k -> half 0
      asm volatile("ds_write2_b32 %0, %1, %2 offset0:64 offset1:65" :: "v"(a), "v"(stage[j].y), "v"(stage[j].w) : "memory"); // odd k -> half 1
    }
  };
  // the compiler's wait-count bookkeeping does not see the DS stores issued from inline asm: before a barrier that publishes
  // them, wait for them by hand
#define MT_LDS_STORES_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
  auto gload = [&](int st) {
#pragma unroll
    for (int j = 0; j < MT_STAGE; j++) {
      const int p2 = min(tile_col0(G, st) + scol + 2 * MT_WG_WAVES * j, G.n2 - 1);
      stage[j] = reinterpret_cast<const float4 *>(set2 + (size_t)p2 * G.stride2 + G.data_off2)[f4];
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < MT_STAGE; j++) {
      float *d = &Bs[buf][(scol + 2 * MT_WG_WAVES * j) * MT_BSTRIDE + 2 * f4];       // k = 4*f4 .. 4*f4+3
      *reinterpret_cast<float2 *>(d) = make_float2(stage[j].x, stage[j].z);        // even k -> half 0
      *reinterpret_cast<float2 *>(d + 64) = make_float2(stage[j].y, stage[j].w);   // odd k  -> half 1
    }
  };
#define MT_LDS_STORES_DONE() do { } while (0)
#endif

#ifndef MT_PIPE_EPILOGUE
#define MT_PIPE_EPILOGUE 1
#endif
#ifndef MT_EARLY_STORE
#define MT_EARLY_STORE 1
#endif
#ifndef MT_EARLY_STORE_AT
#define MT_EARLY_STORE_AT 10
#endif
// timing-only experiments (wrong results): what the barrier / the LDS store / the global loads cost (tools/variants.sh)
#ifndef MT_EXP_NOBARRIER
#define MT_EXP_NOBARRIER 0
#endif
#ifndef MT_EXP_NOSTORE
#define MT_EXP_NOSTORE 0
#endif
#ifndef MT_EXP_NOGLOAD
#define MT_EXP_NOGLOAD 0
#endif
#if (MT_EXP_NOBARRIER || MT_EXP_NOSTORE || MT_EXP_NOGLOAD) && !defined(MISIFT_TIMING_ONLY_BUILD)
#error "MT_EXP_* are timing-only experiments that compute WRONG results: build them with -DMISIFT_TIMING_ONLY_BUILD (tools/variants.sh), never into libmisift.so"
#endif
  if (st0 < st1) {
    gload(st0);
#if MT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MT_STAMP_MAX(2);               // rows of set 1 and the first super-tile have arrived
#endif
    lstore(0);
  }
  MT_LDS_STORES_DONE();
  __syncthreads();
  MT_STAMP_MAX(3);                 // first super-tile staged
#if MT_PIPE_EPILOGUE
  // Software pipeline over the super-tiles (r03): the top-2 update of tile t-1 (128 VALU instructions on its 32 finished
  // accumulator registers) is issued BETWEEN the MFMAs of tile t instead of after them, so a wavefront's matrix pipe
  // never waits for its own epilogue.  Two accumulator sets alternate (the loop body is instantiated for both, no
  // register copies): 32 more VGPRs, still 2 wavefronts per SIMD.
  auto tile = [&](const int st, floatx16 &acc0, floatx16 &acc1, const floatx16 &prev0, const floatx16 &prev1,
                  const bool have_prev) __attribute__((always_inline)) {
    const int buf = (st - st0) & 1;
#if !MT_EXP_NOGLOAD
    gload(min(st + 1, st1 - 1));
#endif
    const float4 *b0 = reinterpret_cast<const float4 *>(&Bs[buf][col * MT_BSTRIDE + half * 64]);
    const float4 *b1 = reinterpret_cast<const float4 *>(&Bs[buf][(col + 32) * MT_BSTRIDE + half * 64]);
    acc0 = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc1 = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int pc0 = tile_col0(G, st - 1) + col, pc1 = pc0 + 32;          // columns of the previous tile
    const bool do0 = have_prev && pc0 < G.ncols, do1 = have_prev && pc1 < G.ncols;
    float4 p0 = b0[0], p1 = b1[0], q0, q1;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      q0 = b0[i + 1]; q1 = b1[i + 1];
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 0], p0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 0], p1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 1], p0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 1], p1.y, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // previous tile, ascending column order within the residue class: its columns 0-31 (chain 0) during the first
      // four slots of this tile's k-loop, columns 32-63 (chain 1) during the last four; four rows per slot
      {
        const int t4 = 4 * ((i >> 1) & 3);
#if MT_TOP2_FILTER
        // r06: a score changes a lane's top two only if it beats the running SECOND best — after a few hundred columns that
        // is rare (2/n per lane), so the three instructions behind the compare run only when some lane of the wavefront needs
        // them (wave-uniform branch): on 100 000 columns ~17 % of the row updates.  Exact: a score that is not above `sec`
        // (or is NaN) leaves (mx, sec, ix) untouched in top2_update as well.
        if (i < 8) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float sc = prev0[t4 + r];
            if (__builtin_amdgcn_ballot_w64(do0 && sc > sec[t4 + r]) != 0ull) {
              if (do0) top2_update(sc, pc0, mx[t4 + r], sec[t4 + r], ix[t4 + r]);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float sc = prev1[t4 + r];
            if (__builtin_amdgcn_ballot_w64(do1 && sc > sec[t4 + r]) != 0ull) {
              if (do1) top2_update(sc, pc1, mx[t4 + r], sec[t4 + r], ix[t4 + r]);
            }
          }
        }
#else
        if (i < 8) {
          if (do0) {
#pragma unroll
            for (int r = 0; r < 4; r++) top2_update(prev0[t4 + r], pc0, mx[t4 + r], sec[t4 + r], ix[t4 + r]);
          }
        } else {
          if (do1) {
#pragma unroll
            for (int r = 0; r < 4; r++) top2_update(prev1[t4 + r], pc1, mx[t4 + r], sec[t4 + r], ix[t4 + r]);
          }
        }
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 2], p0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 2], p1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 3], p0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 3], p1.w, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < 16) { p0 = b0[i + 2]; p1 = b1[i + 2]; }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 4], q0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 4], q1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 5], q0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 5], q1.y, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#if MT_EARLY_STORE
      // the next tile's operands go to the other LDS buffer in the MIDDLE of this tile's MFMA stream (the loads were
      // issued at its start; that buffer's readers all passed the previous barrier): nothing but the barrier itself is
      // left between the last MFMA of this tile and the first operand read of the next
#if !MT_EXP_NOSTORE
      if (i == MT_EARLY_STORE_AT) lstore(buf ^ 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
#endif
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 6], q0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 6], q1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 7], q0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 7], q1.w, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if !MT_EARLY_STORE
    lstore(buf ^ 1);
#endif
#if !MT_EXP_NOBARRIER
    MT_LDS_STORES_DONE();
    __syncthreads();
#endif
  };
  {
    floatx16 A0, A1, B0, B1;
    int st = st0;
    bool have = false;
    for (; st + 1 < st1; st += 2) {
      tile(st, A0, A1, B0, B1, have);
      tile(st + 1, B0, B1, A0, A1, true);
      have = true;
    }
    if (st < st1) {                       // an odd tile left: it finishes B, then its own results are in A
      tile(st, A0, A1, B0, B1, have);
      const int c0 = tile_col0(G, st) + col, c1 = c0 + 32;
      if (c0 < G.ncols) {
#pragma unroll
        for (int r = 0; r < 16; r++) top2_update(A0[r], c0, mx[r], sec[r], ix[r]);
      }
      if (c1 < G.ncols) {
#pragma unroll
        for (int r = 0; r < 16; r++) top2_update(A1[r], c1, mx[r], sec[r], ix[r]);
      }
    } else if (have) {                    // the last tile of an even count sits in B
      const int c0 = tile_col0(G, st1 - 1) + col, c1 = c0 + 32;
      if (c0 < G.ncols) {
#pragma unroll
        for (int r = 0; r < 16; r++) top2_update(B0[r], c0, mx[r], sec[r], ix[r]);
      }
      if (c1 < G.ncols) {
#pragma unroll
        for (int r = 0; r < 16; r++) top2_update(B1[r], c1, mx[r], sec[r], ix[r]);
      }
    }
  }
#else
  for (int st = st0; st < st1; st++) {
    const int buf = (st - st0) & 1;
    gload(min(st + 1, st1 - 1));     // unconditional (the last iteration re-fetches its own tile): no phi copies of the 32 staging registers
    const float4 *b0 = reinterpret_cast<const float4 *>(&Bs[buf][col * MT_BSTRIDE + half * 64]);
    const float4 *b1 = reinterpret_cast<const float4 *>(&Bs[buf][(col + 32) * MT_BSTRIDE + half * 64]);
    floatx16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    floatx16 acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // software pipeline: the ds_read_b128 pair of the next 4 k-pairs is in flight while 8 MFMAs run
    float4 p0 = b0[0], p1 = b1[0], q0, q1;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      q0 = b0[i + 1]; q1 = b1[i + 1];
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch ahead of the MFMAs that hide it
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 0], p0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 0], p1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 1], p0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 1], p1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 2], p0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 2], p1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 3], p0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 3], p1.w, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < 16) { p0 = b0[i + 2]; p1 = b1[i + 2]; }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 4], q0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 4], q1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 5], q0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 5], q1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 6], q0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 6], q1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 7], q0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * i + 7], q1.w, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ascending column order within the residue class: columns 0-31 of the super-tile first
    const int c0 = tile_col0(G, st) + col, c1 = c0 + 32;
    if (c0 < G.ncols) {
#pragma unroll
      for (int r = 0; r < 16; r++) top2_update(acc0[r], c0, mx[r], sec[r], ix[r]);
    }
    if (c1 < G.ncols) {
#pragma unroll
      for (int r = 0; r < 16; r++) top2_update(acc1[r], c1, mx[r], sec[r], ix[r]);
    }
    lstore(buf ^ 1);
    MT_LDS_STORES_DONE();
    __syncthreads();
  }

#endif
  MT_STAMP_MAX(4);                 // sweep done
  // ---- reduce the 4 residues of a class (lanes 4c..4c+3 of the same half): exact merge
#pragma unroll
  for (int r = 0; r < 16; r++)
    top2_merge(mx[r], sec[r], ix[r], quad_xchg<0xB1>(mx[r]), quad_xchg<0xB1>(sec[r]), quad_xchg<0xB1>(ix[r]));
#pragma unroll
  for (int r = 0; r < 16; r++)
    top2_merge(mx[r], sec[r], ix[r], quad_xchg<0x4E>(mx[r]), quad_xchg<0x4E>(sec[r]), quad_xchg<0x4E>(ix[r]));
  if ((lane & 3) == 0) {
    const int cls = col >> 2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int rl = rb * MT_ROWS_PER_BLOCK + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (rl < G.row_count) {
        float *p = partial + (((size_t)rl * 8 + cls) * G.nchunks_total + G.chunk_base + chunk) * 3;
        p[0] = mx[r];
        p[1] = sec[r];
        reinterpret_cast<int *>(p)[2] = ix[r];
      }
    }
  }
  MT_STAMP_MAX(5);                 // partial results stored
}

// Eight threads per row, one per class: each merges its class over the chunks (a contiguous run of 12-byte triples; r02
// had one thread per row walk a strided [chunk][24] table — 0.34 ms for 12 500 rows x 126 chunks, 13 % of that sweep),
// then the eight classes are combined in every lane of the group exactly as before and lane 0 writes the row.
__global__ __launch_bounds__(256) void match_merge_kernel(SiftPointD *__restrict__ pts1,
                                                          const float *__restrict__ set2, MatchGeom G,
                                                          const float *__restrict__ partial, int exact_top2,
                                                          unsigned *__restrict__ ticket, unsigned *__restrict__ host_flag,
                                                          unsigned host_seq)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int rl = t >> 3;
  const bool live = rl < G.row_count;
  MT_STAMP_MIN(8);
  MT_STAMP_MAX(9);
  float m = 0.0f, sd = 0.0f;
  int ixm = -1;
  if (live) {
    const float *p = partial + (size_t)t * G.nchunks_total * 3;
    int ch = 0;
    for (; ch + 4 <= G.nchunks_total; ch += 4) {                  // four triples in flight
      float a[12];
#pragma unroll
      for (int k = 0; k < 12; k++) a[k] = p[3 * ch + k];
#pragma unroll
      for (int k = 0; k < 4; k++) top2_merge(m, sd, ixm, a[3 * k], a[3 * k + 1], __float_as_int(a[3 * k + 2]));
    }
    for (; ch < G.nchunks_total; ch++) top2_merge(m, sd, ixm, p[3 * ch], p[3 * ch + 1], __float_as_int(p[3 * ch + 2]));
  }
  float cmax[8], csec[8];
  int cidx[8];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    cmax[c] = __shfl(m, c, 8);
    csec[c] = __shfl(sd, c, 8);
    cidx[c] = __shfl(ixm, c, 8);
  }
  if (live && (t & 7) == 0) {
    float max_score, sec_score;
    int index;
    if (exact_top2) {
      max_score = cmax[0]; sec_score = csec[0]; index = cidx[0];
  #pragma unroll
      for (int c = 1; c < 8; c++) top2_merge(max_score, sec_score, index, cmax[c], csec[c], cidx[c]);
    } else {
      // the reference's final merge, literally (matching.cu:375-390)
      max_score = cmax[0]; sec_score = csec[0]; index = cidx[0];
  #pragma unroll
      for (int y = 0; y < 8; y++)
        if (index != cidx[y]) {
          if (cmax[y] > max_score) {
            sec_score = fmaxf(max_score, sec_score);
            max_score = cmax[y];
            index = cidx[y];
          } else if (cmax[y] > sec_score)
            sec_score = cmax[y];
        }
    }
    SiftPointD *o = &pts1[G.row_begin + rl];
    o->score = max_score;
    o->match = index;
    const float *m2 = set2 + (size_t)(index >= 0 ? index : 0) * G.stride2 + G.xy_off2;
    o->match_xpos = index >= 0 ? m2[0] : 0.0f;              // never reads sift2[-1] (Appendix B #9)
    o->match_ypos = index >= 0 ? m2[1] : 0.0f;
    o->ambiguity = sec_score / (max_score + 1e-6f);
  }
  // Synchronous callers (misift_match): the workgroup that draws the last ticket stores the call's sequence number in
  // pinned host memory, which the host polls instead of synchronising the stream (r04 single-call budget: ~5 us of the
  // 39 us a 2000 x 2000 MatchSiftData took).  The ticket word is left at zero for the next call.
  MT_STAMP_MAX(10);                // rows written
  if (!host_flag) return;
  __shared__ unsigned s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // vmcnt(0): this wavefront's row stores are acknowledged BEFORE the ticket is
                                             // drawn (the workgroup-scope fence alone waits for lgkmcnt only; advisor r04)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host_flag, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    MT_STAMP_MAX(11);              // flag on its way to the host
  }
}

// Column chunks for a launch of `ntiles` super-tiles over nrb row blocks.  Measured (r03, tools/match_chunks.py,
// profiles/r03_match_chunks.json): a CU works its workgroups off at a fixed rate, two resident at a time, so the launch
// takes ceil(workgroups / CUs) "CU rounds" of one chunk each plus ~3/4 of a super-tile of prologue/epilogue per workgroup,
// and what hurts is a chunk count that leaves the last round nearly empty (12 500 rows: 42 chunks = 16.08 rounds is the
// worst of 14 counts tried, 26 chunks = 9.95 rounds the best) — r02's fixed "24 workgroups per slot" gave 2-3 tiles per
// chunk at 16 384 x 16 384 (0.78 ms; 10 chunks: 0.61 ms).  So: the count that minimises rounds x (tiles per chunk + 3/4).
static int match_plan_param(const char *name, int dflt)
{
  const char *e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : dflt;
}
static void plan_chunks_cus(int ncu, int nrb, int ntiles, int &nchunks, int &tiles_per_chunk)
{
  static const int forced = match_plan_param("MISIFT_MATCH_CHUNKS", 0);       // experiments only
  nchunks = 1; tiles_per_chunk = ntiles > 0 ? ntiles : 1;
  if (ntiles <= 0 || nrb <= 0) return;
  if (ncu <= 0) ncu = 256;
  const int cmax = ntiles < 256 ? ntiles : 256;
  double best = 0.0;
  for (int c = 1; c <= cmax; c++) {
    const int tpc = (ntiles + c - 1) / c;
    if ((ntiles + tpc - 1) / tpc != c) continue;                    // the same cut as a smaller count
    const long long m = ((long long)nrb * c + ncu - 1) / ncu;      // workgroups of the most loaded CU; two run side by side,
    const double rounds = 2.0 * (m / 2) + 1.35 * (m & 1);           // one alone does not keep the matrix pipe full
    const double cost = rounds * (tpc + 0.75);
    if (c == 1 || cost < best * 0.999) { best = cost; nchunks = c; tiles_per_chunk = tpc; }
  }
  if (forced) {
    nchunks = forced < ntiles ? forced : ntiles;
    tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
    nchunks = (ntiles + tiles_per_chunk - 1) / tiles_per_chunk;
  }
}

static void plan_chunks(const misift_ctx *ctx, int nrb, int ntiles, int &nchunks, int &tiles_per_chunk)
{
  plan_chunks_cus(ctx->num_cus, nrb, ntiles, nchunks, tiles_per_chunk);
}
// host-only test hook (no device needed): the chunk plan for n1 rows x n2 columns on a chip of `num_cus` CUs
extern "C" int misift_test_match_plan(int num_cus, int n1, int n2, int *nchunks, int *tiles_per_chunk, int *ntiles)
{
  if (!nchunks || !tiles_per_chunk || !ntiles || n1 < 0 || n2 < 0) return MISIFT_EINVAL;
  const int ncols = MT_TILE * (n2 / MT_TILE);
  *ntiles = (ncols + MT_SUPER - 1) / MT_SUPER;
  plan_chunks_cus(num_cus, (n1 + MT_ROWS_PER_BLOCK - 1) / MT_ROWS_PER_BLOCK, *ntiles, *nchunks, *tiles_per_chunk);
  return MISIFT_OK;
}

// The sweep in up to two launches: first the super-tiles [own_t0, own_t1) read through `pts2_own` (a pointer such that
// pts2_own[column] is valid for exactly those tiles' columns: the rank's own shard of set 2, wherever it lies), then —
// behind `rest_ready` on the context stream — every other super-tile from pts2; one merge over the chunks of both.  The
// per-class top-2 summaries and their exact merge do not depend on how the columns are cut (top2_merge: ties go to
// the smaller column), so the result is the single sweep's, bit for bit.  own_t0 == own_t1: the plain single launch.
// `phase` lets the caller post its exchange between the two launches (the chunk plan is a pure function of the arguments).
int launch_match_split(misift_ctx *ctx, SiftPointD *pts1, int row_begin, int row_count, const SiftPointD *pts2, int n2,
                       const SiftPointD *pts2_own, int own_t0, int own_t1, hipEvent_t rest_ready, int phase, int packed2)
{
  if (row_count <= 0 || n2 <= 0) return MISIFT_OK;
  MatchGeom G;
  memset(&G, 0, sizeof(G));                    // (phases that skip a launch still pass G by value to the merge)
  G.row_begin = row_begin; G.row_count = row_count; G.n1_total = row_begin + row_count;
  G.n2 = n2;
  if (packed2) { G.stride2 = MISIFT_MATCH_COLUMN_BYTES / 4; G.data_off2 = 0; G.xy_off2 = 128; }
  else { G.stride2 = MISIFT_POINT_BYTES / 4; G.data_off2 = 16; G.xy_off2 = 0; }
  const float *f2 = reinterpret_cast<const float *>(pts2), *f2_own = reinterpret_cast<const float *>(pts2_own);
  G.ncols = ctx->opt.match_full ? n2 : MT_TILE * (n2 / MT_TILE);
  const int ntiles_all = (G.ncols + MT_SUPER - 1) / MT_SUPER;                 // 64-column super-tiles
  const int nrb = (row_count + MT_ROWS_PER_BLOCK - 1) / MT_ROWS_PER_BLOCK;
  if (own_t1 > ntiles_all) own_t1 = ntiles_all;
  if (own_t0 < 0) own_t0 = 0;
  const int n_own = own_t1 > own_t0 ? own_t1 - own_t0 : 0, n_rest = ntiles_all - n_own;
  int ch_own = 0, tpc_own = 1, ch_rest = 0, tpc_rest = 1;
  if (n_own) plan_chunks(ctx, nrb, n_own, ch_own, tpc_own);
  if (n_rest || !n_own) plan_chunks(ctx, nrb, n_rest, ch_rest, tpc_rest);
  G.nchunks_total = ch_own + ch_rest;
  const size_t need = (size_t)row_count * G.nchunks_total * MT_PART_WORDS * sizeof(float);
  {
    int rc = misift_ensure_tmp(ctx, need);
    if (rc) return rc;
  }
  float *partial = reinterpret_cast<float *>(ctx->d_match_tmp);
  if (n_own && phase != MATCH_PHASE_REST) {
    G.ntiles = n_own; G.nchunks = ch_own; G.tiles_per_chunk = tpc_own;
    G.tile_base = own_t0; G.hole_begin = 0x7fffffff; G.hole_len = 0; G.chunk_base = 0;
    LaunchScope ls(ctx, "match_mfma");
    hipLaunchKernelGGL(match_kernel, dim3(nrb * ch_own), dim3(64 * MT_WG_WAVES), 0, ctx->stream, pts1, f2_own, G, partial);
    int rc = ls.finish();
    if (rc) return rc;
  }
  if (phase == MATCH_PHASE_OWN) return MISIFT_OK;
  if (rest_ready) HIP_TRY(hipStreamWaitEvent(ctx->stream, rest_ready, 0));
  if (ch_rest) {
    G.ntiles = n_rest; G.nchunks = ch_rest; G.tiles_per_chunk = tpc_rest;
    G.tile_base = 0; G.hole_begin = n_own ? own_t0 : 0x7fffffff; G.hole_len = n_own; G.chunk_base = ch_own;
    LaunchScope ls(ctx, "match_mfma");
    hipLaunchKernelGGL(match_kernel, dim3(nrb * ch_rest), dim3(64 * MT_WG_WAVES), 0, ctx->stream, pts1, f2, G, partial);
    int rc = ls.finish();
    if (rc) return rc;
  }
  {
    LaunchScope ls(ctx, "match_merge");
    unsigned *host_flag = nullptr;
    if (ctx->want_match_flag && ctx->h_flags && ctx->d_flags) {
      host_flag = ctx->h_flags;
      ctx->match_seq++;
      ctx->match_flagged = 1;
    }
    hipLaunchKernelGGL(match_merge_kernel, dim3((row_count * 8 + 255) / 256), dim3(256), 0, ctx->stream, pts1, f2,
                       G, partial, ctx->opt.match_exact_top2, ctx->d_flags, host_flag, ctx->match_seq);
    return ls.finish();
  }
}

int launch_match(misift_ctx *ctx, SiftPointD *pts1, int row_begin, int row_count, const SiftPointD *pts2, int n2)
{
  return launch_match_split(ctx, pts1, row_begin, row_count, pts2, n2, nullptr, 0, 0, nullptr, MATCH_PHASE_ALL, 0);
}

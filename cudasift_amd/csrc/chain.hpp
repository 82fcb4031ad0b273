// chain.hpp — the ScaleDown chain of small batches: shared by its own kernel (kernels_pyramid.hip) and by the
// merged-octave scan, whose first workgroups run it in the same launch (kernels_dog.hip).
#pragma once
#include "common.hpp"

// ------------------------------------------------- ScaleDown chain (small batches)
// Up to three consecutive ScaleDowns in ONE launch: a workgroup owns a T x T tile of the LAST level of the chain and
// computes the cone of pixels under it on every level in LDS (level k region = 2 * level k+1 region + 3 per axis), so
// the coarse pyramid of a frame costs one dependent dispatch instead of three (a single 1080p frame: 37 us -> one
// launch; r04 single-call budget, profiles/r04_single_call_*).  Every output pixel is the scaledown_kernel expression
// on the same operands (horizontal pass on clamped source rows, then vertical: bit-identical); pixels in the overlap of
// neighbouring cones are computed by both workgroups and stored by the one whose tile they lie under.  The redundant
// arithmetic (1.4-1.8x per level, one pixel per lane instead of DPP quads) makes it the wrong tool for a 64-frame
// batch, where the extraction is bound by VALU issue: those keep the three streamed launches beside the fine scan.
#define CHAIN_MAX_LEVELS 3
struct ChainLevel { int w, h, p; long long off; };       // off: float offset inside a frame's arena
struct ChainGeom {
  int K, T, tiles_x, tiles_y;
  long long frame_stride;
  ChainLevel lv[CHAIN_MAX_LEVELS + 1];                     // lv[0] = source, lv[1..K] = outputs
};

// floats of LDS a workgroup of the chain needs: n0^2 + n0*n1 + n1^2 with n(k-1) = 2 n(k) + 3, n(K) = T
static inline int chain_lds_floats(int K, int T)
{
  const int n1 = ((T + 3) << (K - 1)) - 3, n0 = ((T + 3) << K) - 3;
  return n0 * n0 + n0 * n1 + n1 * n1;
}
#define CHAIN_LDS_FLOATS_MAX (85 * 85 + 85 * 41 + 41 * 41)     // standalone kernel: T << K = 64
#define CHAIN_LDS_FLOATS_EMBED (53 * 53 + 53 * 25 + 25 * 25)   // inside the scan launch:  T << K = 32

#ifdef __HIPCC__
// One workgroup (any multiple of 64 threads): tile `tile` of the chain's last level in frame `frame_index`.
__device__ __forceinline__ void scaledown_chain_block(float *__restrict__ scratch, const ChainGeom &G, const Taps5 &t,
                                                      int tile, int frame_index, float *lds)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int tx = tile % G.tiles_x, ty = tile / G.tiles_x;
  float *frame = scratch + (long long)frame_index * G.frame_stride;
  const float k0 = t.k[0], k1 = t.k[1], k2 = t.k[2];
  // region of level k: [ax(k), ax(k) + n(k)) x [ay(k), ay(k) + n(k)); a(k-1) = 2 a(k) - 2 and n(k-1) = 2 n(k) + 3 in closed
  // form (a - 2 and n + 3 double per level), so nothing is indexed dynamically
  auto ax = [&](int k) -> int { return (tx * G.T - 2) * (1 << (G.K - k)) + 2; };
  auto ay = [&](int k) -> int { return (ty * G.T - 2) * (1 << (G.K - k)) + 2; };
  auto n = [&](int k) -> int { return ((G.T + 3) << (G.K - k)) - 3; };
  // LDS (chain_lds_floats(G) floats): region of level k-1 / level k alternately in s_a and s_b, the horizontal pass
  // (rows of level k-1, columns of level k) in s_h
  float *const s_a = lds, *const s_h = lds + n(0) * n(0), *const s_b = s_h + n(0) * n(1);
  {                                                         // stage the source region (clamp-to-edge)
    const ChainLevel &S = G.lv[0];
    const float *src = frame + S.off;
    const int n0 = n(0), ax0 = ax(0), ay0 = ay(0);
    // eight rows per trip, all loads issued before the first LDS store: a rolled load -> store loop is one memory
    // round trip per row and wavefront (14 of them in a 256-thread workgroup: the chain took 20 us inside the scan)
    const int c0 = clampi(ax0 + lane, 0, S.w - 1), c1 = clampi(ax0 + lane + 64, 0, S.w - 1);     // n0 <= 85: two columns per lane
    for (int j0 = wave; j0 < n0; j0 += 8 * nwaves) {
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u * nwaves;
        const float *row = src + (size_t)clampi(ay0 + min(j, n0 - 1), 0, S.h - 1) * S.p;
        v0[u] = row[c0];
        v1[u] = row[c1];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u * nwaves;
        if (j < n0) {
          if (lane < n0) s_a[j * n0 + lane] = v0[u];
          if (lane + 64 < n0) s_a[j * n0 + lane + 64] = v1[u];
        }
      }
    }
  }
  __syncthreads();
  float *cur = s_a, *nxt = s_b;
  for (int k = 1; k <= G.K; k++) {
    const ChainLevel &L = G.lv[k];
    const int ns = n(k - 1), nd = n(k), axk = ax(k), ayk = ay(k), axs = ax(k - 1), ays = ay(k - 1);
    // Both passes walk their (row, column) pairs FLAT over the workgroup's threads: the regions are 4..85 wide, a
    // column-per-lane loop would leave most of a wavefront idle (25 of 64 lanes on the first level of an embedded chain).
    // row = idx / nd by a float multiply: (idx + 0.5) / nd is at least 0.5 / nd away from an integer, far above the
    // rounding error for idx < 2^12.
    const float inv_nd = 1.0f / (float)nd;
    // horizontal: every row of the source region, columns of this level (an out-of-image column holds the value of
    // the clamped one, like the region it is read from)
    for (int idx = threadIdx.x; idx < ns * nd; idx += blockDim.x) {
      const int j = (int)(((float)idx + 0.5f) * inv_nd), i = idx - j * nd;
      const int cx = clampi(axk + i, 0, L.w - 1);
      const float *r = cur + j * ns + (2 * cx - axs);      // source column 2 * cx
      const float s = __builtin_fmaf(k0, r[-2] + r[2], k1 * (r[-1] + r[1]));
      s_h[idx] = __builtin_fmaf(k2, r[0], s);
    }
    __syncthreads();
    // vertical + store of the pixels under this workgroup's tile
    const int sh = G.K - k;                                 // own block of level k: tile * T << sh
    const int ox0 = (tx * G.T) << sh, oy0 = (ty * G.T) << sh;
    const int ox1 = tx == G.tiles_x - 1 ? L.w : min(((tx + 1) * G.T) << sh, L.w);
    const int oy1 = ty == G.tiles_y - 1 ? L.h : min(((ty + 1) * G.T) << sh, L.h);
    float *dst = frame + L.off;
    for (int idx = threadIdx.x; idx < nd * nd; idx += blockDim.x) {
      const int j = (int)(((float)idx + 0.5f) * inv_nd), i = idx - j * nd;
      const int y = ayk + j, cy = clampi(y, 0, L.h - 1);
      const float *c = s_h + (2 * cy - ays) * nd + i;      // source row 2 * cy
      float v = __builtin_fmaf(k2, c[0], k0 * (c[-2 * nd] + c[2 * nd]));
      v = __builtin_fmaf(k1, c[-nd] + c[nd], v);
      nxt[idx] = v;
      const int x = axk + i;
      // (agent-scope store = write-through: inside the scan launch the readers are workgroups on other XCDs, and a
      //  release fence per workgroup would write the whole L2 back each time)
      if (x >= ox0 && x < ox1 && y >= oy0 && y < oy1)
        __hip_atomic_store(&dst[(size_t)y * L.p + x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    float *tmp = cur; cur = nxt; nxt = tmp;
  }
}
#endif

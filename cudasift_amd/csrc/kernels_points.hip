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

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// ------------------------------------------------------------- orientation
struct OrientResult { float ori1, ori2; bool has2; };      // meaningful in lane 0 only

// Orientation of one keypoint by one wavefront (reference cudaSiftD.cu:984-1037).  hist[64], gauss[16] and
// smp[128], tgrid[169] are wave-private LDS slices; smp[121..127] must hold bin -1 (never matches).
__device__ __forceinline__ OrientResult orient_core(const float *img, int w, int h, int pitch, bool q8, float xpos,
                                                    float ypos, float scale, float *hist, float *gauss,
                                                    float2 *smp, float *tgrid, int lane)
{
  const float i2sigma2 = -1.0f / (2.0f * 1.5f * 1.5f * scale * scale);
  if (lane < 11) gauss[lane] = expf(i2sigma2 * (lane - 5) * (lane - 5));
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
        int bin = (int)(16.0f * atan2f(dy, dx) / 3.1416f + 16.5f);
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
        int bin = (int)(16.0f * atan2f(dy, dx) / 3.1416f + 16.5f);
        if (bin > 31) bin = 0;
        const float grad = sqrtf(dx * dx + dy * dy);
        smp[tx] = make_float2((float)bin, grad * gauss[xd] * gauss[yd]);
      }
    }
  }
  wave_sync();
  // privatized histogram (no LDS atomics): lane (b, half) sums the samples of its half that fall in bin b
  {
    const float fb = (float)(lane & 31);
    const float2 *sp = smp + (lane >> 5) * 64;
    float acc = 0.0f;
#pragma unroll 16
    for (int j = 0; j < 64; j++) {
      const float2 e = sp[j];
      acc += (e.x == fb) ? e.y : 0.0f;
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
  float maxval1 = pk;
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) maxval1 = fmaxf(maxval1, __shfl_xor(maxval1, m, 64));
  maxval1 = __shfl(maxval1, 0, 64);
  const unsigned long long b1 = __ballot(lane < 32 && pk == maxval1 && maxval1 > 0.0f);
  const int i1 = b1 ? __ffsll((long long)b1) - 1 : -1;
  const float pk2 = lane == i1 ? 0.0f : pk;
  float maxval2 = pk2;
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) maxval2 = fmaxf(maxval2, __shfl_xor(maxval2, m, 64));
  maxval2 = __shfl(maxval2, 0, 64);
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
  const bool q8 = frac8 != 0;

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
__device__ __forceinline__ float fast_atan2(float y, float x)
{
  const float absx = fabsf(x), absy = fabsf(y);
  const float mx = fmaxf(absx, absy), mn = fminf(absx, absy);
  if (mx == 0.0f) return 0.0f;
  const float a = mn / mx;
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
  if (lane < 16) gauss[lane] = expf(-(lane - 7.5f) * (lane - 7.5f) / 128.0f);
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
  const float sina = sinf(theta);
  const float cosa = cosf(theta);
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
  const bool q8 = frac8 != 0;
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

__global__ __launch_bounds__(256) void orient_all_kernel(const float *__restrict__ scratch, PyramidInfo P,
                                                         unsigned *__restrict__ counters,
                                                         Detection *__restrict__ det, int max_pts, int frac8)
{
  __shared__ float s_hist[WAVES_PER_BLOCK][64];
  __shared__ float s_gauss[WAVES_PER_BLOCK][16];
  __shared__ float2 s_smp[WAVES_PER_BLOCK][128];
  __shared__ float s_tgrid[WAVES_PER_BLOCK][176];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  unsigned *cnt = counters + (size_t)frame * CNT_STRIDE;
  Detection *fdet = det + (size_t)frame * MISIFT_MAX_OCTAVES * max_pts;
  const bool q8 = frac8 != 0;
  if (lane >= 57) s_smp[wave][64 + lane] = make_float2(-1.0f, 0.0f);
  const FrameCounts fc = load_frame_counts(cnt, P.noct, max_pts, false);
  const int stride = gridDim.x * WAVES_PER_BLOCK;
  int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + wave);
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
    more = flat_to_octave(fc, P.noct, idx, o, i);
    if (more) nxt = *reinterpret_cast<const float4 *>(&fdet[(size_t)(o - 1) * max_pts + i]);
    const OrientResult r = orient_core(img, L.w, L.h, L.p, q8, cur.x, cur.y, cur.z, s_hist[wave],
                                       s_gauss[wave], s_smp[wave], s_tgrid[wave], lane);
    if (lane == 0) {
      d->ori1 = r.ori1;
      d->ori2 = r.ori2;
      d->dupslot = r.has2 ? (int)atomicAdd(&cnt[CNT_DUP + co], 1u) : -1;
    }
  }
}

#ifndef DESCR_OCC
#define DESCR_OCC 4
#endif
__global__ __launch_bounds__(256, DESCR_OCC) void descr_all_kernel(const float *__restrict__ scratch, PyramidInfo P,
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
  const bool q8 = frac8 != 0;
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
          p->xpos = d.xpos * L.subsampling;
          p->ypos = d.ypos * L.subsampling;
          p->scale = d.scale * L.subsampling;
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

// ------------------------------------------------------------- host wrappers
static inline int points_grid_x(misift_ctx *ctx, int nframes, int blocks_per_cu = 0)
{
  if (blocks_per_cu <= 0) blocks_per_cu = ctx->point_blocks_per_cu;
  // enough wavefronts to cover a few thousand keypoints per frame while keeping
  // the total near a few waves per SIMD when many frames are batched
  int per_frame = (ctx->num_cus * blocks_per_cu + nframes - 1) / nframes;
  if (per_frame < 8) per_frame = 8;
  if (per_frame > 512) per_frame = 512;
  return per_frame;
}

int launch_orient(misift_ctx *ctx, const float *base, long long base_frame_stride, int w, int h, int pitch,
                  int nframes, int octave, SiftPointD *pts, int max_pts)
{
  LaunchScope ls(ctx, "orient");
  hipLaunchKernelGGL(orient_kernel, dim3(points_grid_x(ctx, nframes), nframes), dim3(256), 0, ctx->stream, base,
                     base_frame_stride, w, h, pitch, octave, ctx->d_counters, pts, max_pts,
                     ctx->opt.texfrac_bits == 8 ? 1 : 0);
  return ls.finish();
}

int launch_descr(misift_ctx *ctx, const float *base, long long base_frame_stride, int w, int h, int pitch,
                 int nframes, float subsampling, int octave, SiftPointD *pts, int max_pts)
{
  LaunchScope ls(ctx, "descr");
  hipLaunchKernelGGL(descr_kernel, dim3(points_grid_x(ctx, nframes), nframes), dim3(256), 0, ctx->stream, base,
                     base_frame_stride, w, h, pitch, subsampling, octave, ctx->d_counters, pts, max_pts,
                     ctx->opt.texfrac_bits == 8 ? 1 : 0);
  return ls.finish();
}

int launch_orient_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts)
{
  (void)pts;
  LaunchScope ls(ctx, "orient_all");
  hipLaunchKernelGGL(orient_all_kernel, dim3(points_grid_x(ctx, P.nframes, ctx->orient_blocks_per_cu), P.nframes), dim3(256), 0, ctx->stream,
                     scratch, P, ctx->d_counters, ctx->d_det, max_pts, ctx->opt.texfrac_bits == 8 ? 1 : 0);
  return ls.finish();
}

int launch_descr_all(misift_ctx *ctx, const float *scratch, const PyramidInfo &P, SiftPointD *pts, int max_pts,
                     const int *pack_offsets, SiftPointD *pack_dst)
{
  LaunchScope ls(ctx, "descr_all");
  hipLaunchKernelGGL(descr_all_kernel, dim3(points_grid_x(ctx, P.nframes), P.nframes), dim3(256), 0, ctx->stream,
                     scratch, P, ctx->d_counters, ctx->d_det, pts, max_pts, ctx->opt.texfrac_bits == 8 ? 1 : 0,
                     pack_offsets, pack_dst);
  return ls.finish();
}

int launch_rescale(misift_ctx *ctx, SiftPointD *pts, int npts, float scale)
{
  if (npts <= 0) return MISIFT_OK;
  LaunchScope ls(ctx, "rescale");
  hipLaunchKernelGGL(rescale_kernel, dim3((npts + 255) / 256), dim3(256), 0, ctx->stream, pts, npts, scale);
  return ls.finish();
}

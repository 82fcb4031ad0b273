// homography.hip — RANSAC homography over stored matches, on the GPU (SURVEY §8f "next" row 1).
//
// Replaces FindHomography (reference matching.cu:1000-1087) and its kernels
// ComputeHomographies (:907-948, 8x8 solve through InvertMatrix<8> :821-905) and
// TestHomographies (:953-996).  Arithmetic follows oracle/sift_oracle.c orc_find_homography
// operation by operation (explicit fmaf where the reference's expressions contract, exact
// round-toward-zero products where it uses __fmul_rz), so H and the inlier count are
// bit-identical to the oracle for the same libc rand() state.
//
// MI355X shape (latency-bound, a few k points x ~10 k hypotheses):
//   gather   one workgroup: AoS SiftPoint -> SoA coords + ORDERED list of valid points
//            (ballot/popcount compaction keeps index order, which the rand()%numValid
//            sampling depends on); numValid goes back to the host (4 bytes).
//   host     draws the 4 sample indices per hypothesis with libc rand() in the reference's
//            call order (matching.cu:1041-1053) and uploads them.
//   solve    one lane per hypothesis: 8x8 Crout LU with implicit row scaling, inverse by
//            8 unit-vector solves, h = inv(A) b.
//   count    one wavefront per hypothesis sweeps all points (coalesced SoA reads, L2
//            resident), ballot+popcount accumulate -> count[hyp].
//   pick     one workgroup: first hypothesis with the largest count; 8 floats + count D2H.
//
// Deviation: the reference counts inliers over numPts rounded up to 16
// and so reads up to 15 uninitialised coordinates; here exactly numPts points are tested.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.hpp"

namespace {

constexpr int OFF_XPOS = 0, OFF_YPOS = 1, OFF_SCORE = 6, OFF_AMBIG = 7, OFF_MXPOS = 9, OFF_MYPOS = 10;
constexpr int PT_WORDS = (int)(sizeof(SiftPointD) / sizeof(float));

__device__ __forceinline__ float mul_rz(float a, float b)
{
  // exact round-toward-zero product: RN product, then step one ulp toward zero when the exact
  // residual says RN rounded away from zero
  float p = a * b;
  const float e = fmaf(a, b, -p);
  const bool away = (p > 0.0f && e < 0.0f) || (p < 0.0f && e > 0.0f);
  return away ? __uint_as_float(__float_as_uint(p) - 1u) : p;
}

// ---- gather: SoA coordinates + ordered compaction of the valid points -------------------------
__global__ __launch_bounds__(1024) void homo_gather_kernel(const float *__restrict__ pts, int npts, int stride,
                                                             float min_score, float max_ambiguity,
                                                             float *__restrict__ coord, int *__restrict__ valid,
                                                             int *__restrict__ num_valid)
{
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < npts; i0 += 1024) {
    const int i = i0 + tid;
    bool ok = false;
    if (i < npts) {
      const float *p = pts + (size_t)i * PT_WORDS;
      coord[0 * stride + i] = p[OFF_XPOS];
      coord[1 * stride + i] = p[OFF_YPOS];
      coord[2 * stride + i] = p[OFF_MXPOS];
      coord[3 * stride + i] = p[OFF_MYPOS];
      ok = p[OFF_SCORE] > min_score && p[OFF_AMBIG] < max_ambiguity;      // matching.cu:1035
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; w++) off += wave_cnt[w];
    if (ok) valid[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) {
      int s = 0;
      for (int w = 0; w < 16; w++) s += wave_cnt[w];
      base_s += s;
    }
    __syncthreads();
  }
  if (tid == 0) *num_valid = base_s;
}

// ---- solve: one lane per hypothesis -------------------------------------------------------------
// Crout LU with implicit scaling of an 8x8 system, in place; perm[] records the row swaps.
__device__ void lu8(float (&m)[8][8], int (&perm)[8])
{
  float rowscale[8];
  for (int r = 0; r < 8; r++) {
    float big = 0.0f;
    for (int c = 0; c < 8; c++) big = fmaxf(big, fabsf(m[r][c]));
    rowscale[r] = big > 0.0f ? 1.0f / big : 1e16f;
  }
  int piv = 0;
  for (int c = 0; c < 8; c++) {
    for (int r = 0; r < c; r++) {
      float s = m[r][c];
      for (int k = 0; k < r; k++) s = fmaf(-m[r][k], m[k][c], s);
      m[r][c] = s;
    }
    float best = 0.0f;
    for (int r = c; r < 8; r++) {
      float s = m[r][c];
      for (int k = 0; k < c; k++) s = fmaf(-m[r][k], m[k][c], s);
      m[r][c] = s;
      const float merit = rowscale[r] * fabsf(s);
      if (merit >= best) { best = merit; piv = r; }
    }
    if (piv != c) {
      for (int k = 0; k < 8; k++) { const float t = m[piv][k]; m[piv][k] = m[c][k]; m[c][k] = t; }
      rowscale[piv] = rowscale[c];
    }
    perm[c] = piv;
    if (m[c][c] == 0.0f) m[c][c] = 1e-16f;
    if (c != 7) {
      const float inv = 1.0f / m[c][c];
      for (int r = c + 1; r < 8; r++) m[r][c] *= inv;
    }
  }
}

// column `col` of the inverse from the LU factors (forward + back substitution of e_col)
__device__ void lu8_unit_solve(const float (&m)[8][8], const int (&perm)[8], int col, float (&x)[8])
{
  for (int k = 0; k < 8; k++) x[k] = 0.0f;
  x[col] = 1.0f;
  int first = -1;
  for (int r = 0; r < 8; r++) {
    const int p = perm[r];
    float s = x[p];
    x[p] = x[r];
    if (first != -1) {
      for (int k = first; k < r; k++) s = fmaf(-m[r][k], x[k], s);
    } else if (s != 0.0f) {
      first = r;
    }
    x[r] = s;
  }
  for (int r = 7; r >= 0; r--) {
    float s = x[r];
    for (int k = r + 1; k < 8; k++) s = fmaf(-m[r][k], x[k], s);
    x[r] = s / m[r][r];
  }
}

__global__ __launch_bounds__(64) void homo_solve_kernel(const float *__restrict__ coord, int stride,
                                                          const int *__restrict__ valid, const int *__restrict__ sample,
                                                          int num_loops, float *__restrict__ homo)
{
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= num_loops) return;
  float m[8][8], inv[8][8], rhs[8], x[8];
  int perm[8];
  for (int i = 0; i < 4; i++) {
    const int pt = valid[sample[i * num_loops + idx]];
    const float x1 = coord[0 * stride + pt], y1 = coord[1 * stride + pt];
    const float x2 = coord[2 * stride + pt], y2 = coord[3 * stride + pt];
    float *r1 = m[2 * i], *r2 = m[2 * i + 1];
    r1[0] = x1; r1[1] = y1; r1[2] = 1.0f; r1[3] = 0.0f; r1[4] = 0.0f; r1[5] = 0.0f;
    r1[6] = -x2 * x1; r1[7] = -x2 * y1;
    r2[0] = 0.0f; r2[1] = 0.0f; r2[2] = 0.0f; r2[3] = x1; r2[4] = y1; r2[5] = 1.0f;
    r2[6] = -y2 * x1; r2[7] = -y2 * y1;
    rhs[2 * i] = x2;
    rhs[2 * i + 1] = y2;
  }
  lu8(m, perm);
  for (int c = 0; c < 8; c++) {
    lu8_unit_solve(m, perm, c, x);
    for (int r = 0; r < 8; r++) inv[r][c] = x[r];
  }
  for (int r = 0; r < 8; r++) {
    float s = 0.0f;
    for (int k = 0; k < 8; k++) s = fmaf(inv[r][k], rhs[k], s);
    homo[r * num_loops + idx] = s;
  }
}

// ---- count: one wavefront per hypothesis ---------------------------------------------------------
__global__ __launch_bounds__(256) void homo_count_kernel(const float *__restrict__ coord, int stride, int npts,
                                                           const float *__restrict__ homo, int num_loops,
                                                           float thresh2, int *__restrict__ counts)
{
  const int lane = threadIdx.x & 63;
  const int hyp = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (hyp >= num_loops) return;
  float a[8];
  for (int k = 0; k < 8; k++) a[k] = homo[k * num_loops + hyp];
  int cnt = 0;
  for (int i = lane; i < npts; i += 64) {
    const float x1 = coord[0 * stride + i], y1 = coord[1 * stride + i];
    const float x2 = coord[2 * stride + i], y2 = coord[3 * stride + i];
    const float nomx = mul_rz(a[0], x1) + mul_rz(a[1], y1) + a[2];
    const float nomy = mul_rz(a[3], x1) + mul_rz(a[4], y1) + a[5];
    const float deno = mul_rz(a[6], x1) + mul_rz(a[7], y1) + 1.0f;
    const float errx = mul_rz(x2, deno) - nomx;
    const float erry = mul_rz(y2, deno) - nomy;
    const float err2 = mul_rz(errx, errx) + mul_rz(erry, erry);
    cnt += err2 < mul_rz(thresh2, mul_rz(deno, deno)) ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) counts[hyp] = cnt;
}

// ---- pick: first hypothesis with the largest count (strict '>' scan of matching.cu:1063-1068) ----
__global__ __launch_bounds__(1024) void homo_pick_kernel(const int *__restrict__ counts, const float *__restrict__ homo,
                                                           int num_loops, float *__restrict__ result)
{
  __shared__ unsigned long long best_s[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // key = count in the high word, (INT_MAX - index) in the low word: max key = largest count, smallest index
  unsigned long long best = 0;
  for (int i = tid; i < num_loops; i += 1024) {
    const unsigned long long key = ((unsigned long long)(unsigned)counts[i] << 32) | (unsigned)(0x7fffffff - i);
    best = key > best ? key : best;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off, 64);
    best = o > best ? o : best;
  }
  if (lane == 0) best_s[wave] = best;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; w++) best = best_s[w] > best ? best_s[w] : best;
    const int idx = 0x7fffffff - (int)(unsigned)(best & 0xffffffffull);
    for (int k = 0; k < 8; k++) result[k] = homo[k * num_loops + idx];
    reinterpret_cast<int *>(result)[8] = (int)(best >> 32);
    reinterpret_cast<int *>(result)[9] = idx;
  }
}

}  // namespace

extern "C" int misift_find_homography(misift_ctx *ctx, const void *d_pts, int npts, float *homography, int *num_matches,
                                      int num_loops, float min_score, float max_ambiguity, float thresh)
{
  if (!ctx || !homography || !num_matches || num_loops < 1) {
    misift_set_error("misift_find_homography: invalid argument");
    return MISIFT_EINVAL;
  }
  *num_matches = 0;
  homography[0] = homography[4] = homography[8] = 1.0f;
  homography[1] = homography[2] = homography[3] = 0.0f;
  homography[5] = homography[6] = homography[7] = 0.0f;
  if (!d_pts || npts < 8) return MISIFT_OK;                        // matching.cu:1008, :1016-1017
  num_loops = (num_loops + 15) / 16 * 16;
  const int stride = (npts + 15) / 16 * 16;
  // temp layout: coord[4*stride] | valid[stride] | sample[4*loops] | homo[8*loops] | counts[loops] | result[16]
  const size_t words = (size_t)5 * stride + (size_t)13 * num_loops + 16 + 16;
  int rc = misift_ensure_tmp(ctx, words * sizeof(float));
  if (rc) return rc;
  float *coord = reinterpret_cast<float *>(ctx->d_match_tmp);
  int *valid = reinterpret_cast<int *>(coord + (size_t)4 * stride);
  int *sample = valid + stride;
  float *homo = reinterpret_cast<float *>(sample + (size_t)4 * num_loops);
  int *counts = reinterpret_cast<int *>(homo + (size_t)8 * num_loops);
  float *result = reinterpret_cast<float *>(counts + num_loops);
  int *d_num_valid = reinterpret_cast<int *>(result + 16);

  {
    LaunchScope ls(ctx, "homo_gather");
    hipLaunchKernelGGL(homo_gather_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const float *)d_pts, npts, stride,
                       min_score, max_ambiguity, coord, valid, d_num_valid);
    rc = ls.finish();
    if (rc) return rc;
  }
  int numValid = 0;
  HIP_TRY(hipMemcpyAsync(&numValid, d_num_valid, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (numValid < 8) return MISIFT_OK;

  std::vector<int> h_sample((size_t)4 * num_loops);
  for (int i = 0; i < num_loops; i++) {                            // draw order of matching.cu:1041-1053
    int p1 = rand() % numValid;
    int p2 = rand() % numValid;
    int p3 = rand() % numValid;
    int p4 = rand() % numValid;
    while (p2 == p1) p2 = rand() % numValid;
    while (p3 == p1 || p3 == p2) p3 = rand() % numValid;
    while (p4 == p1 || p4 == p2 || p4 == p3) p4 = rand() % numValid;
    h_sample[i + 0 * (size_t)num_loops] = p1;
    h_sample[i + 1 * (size_t)num_loops] = p2;
    h_sample[i + 2 * (size_t)num_loops] = p3;
    h_sample[i + 3 * (size_t)num_loops] = p4;
  }
  HIP_TRY(hipMemcpyAsync(sample, h_sample.data(), h_sample.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  {
    LaunchScope ls(ctx, "homo_solve");
    hipLaunchKernelGGL(homo_solve_kernel, dim3((num_loops + 63) / 64), dim3(64), 0, ctx->stream, coord, stride, valid,
                       sample, num_loops, homo);
    rc = ls.finish();
    if (rc) return rc;
  }
  {
    LaunchScope ls(ctx, "homo_count");
    hipLaunchKernelGGL(homo_count_kernel, dim3((num_loops + 3) / 4), dim3(256), 0, ctx->stream, coord, stride, npts,
                       homo, num_loops, thresh * thresh, counts);
    rc = ls.finish();
    if (rc) return rc;
  }
  {
    LaunchScope ls(ctx, "homo_pick");
    hipLaunchKernelGGL(homo_pick_kernel, dim3(1), dim3(1024), 0, ctx->stream, counts, homo, num_loops, result);
    rc = ls.finish();
    if (rc) return rc;
  }
  float h_result[10];
  HIP_TRY(hipMemcpyAsync(h_result, result, sizeof(h_result), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));                      // also keeps h_sample alive until the upload is done
  memcpy(homography, h_result, 8 * sizeof(float));
  homography[8] = 1.0f;
  memcpy(num_matches, &h_result[8], sizeof(int));
  return MISIFT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// ImproveHomography on the device (reference geomFuncs.cpp:6-72, a HOST function there: it walks
// SiftData.h_data).  Same arithmetic as oracle orc_improve_homography — which is pinned bit for bit against the
// reference's own geomFuncs.cpp — and the same summation ORDER: the 36 distinct entries of the symmetric 8x8 normal
// matrix and the 8 entries of the right-hand side each belong to one lane, and every lane walks the points in index
// order, so each double-precision sum is accumulated exactly as the reference's sequential loop accumulates it.  The
// problem is tiny (a few thousand points, 5 loops); what the device version buys is that the records never leave HBM.
namespace {

struct ImproveArgs {
  const SiftPointD *pts;
  int npts, num_loops;
  float min_score, max_ambiguity, limit;
  double a0[8];
};

__device__ __forceinline__ double pick8(int k, double v0, double v1, double v2, double v3, double v4, double v5, double v6,
                                        double v7)
{
  double r = v0;
  r = k == 1 ? v1 : r; r = k == 2 ? v2 : r; r = k == 3 ? v3 : r; r = k == 4 ? v4 : r;
  r = k == 5 ? v5 : r; r = k == 6 ? v6 : r; r = k == 7 ? v7 : r;
  return r;
}

__global__ __launch_bounds__(64) void improve_homography_kernel(ImproveArgs P, SiftPointD *__restrict__ pts_rw,
                                                                float *__restrict__ result)
{
  __shared__ double s_M[64], s_X[8], s_A[8];
  const int lane = threadIdx.x;
  // lane -> accumulator: lanes 0..35 = M[r][c] for r <= c (row-major upper triangle), lanes 36..43 = X[r]
  int r = 0, c = 0;
  bool is_m = lane < 36, is_x = lane >= 36 && lane < 44;
  if (is_m) {
    int k = lane;
    for (r = 0; r < 8; r++) {
      const int len = 8 - r;
      if (k < len) { c = r + k; break; }
      k -= len;
    }
  } else if (is_x) {
    r = lane - 36;
  }
  if (lane < 8) s_A[lane] = P.a0[lane];
  __syncthreads();
  for (int loop = 0; loop < P.num_loops; loop++) {
    const double A0 = s_A[0], A1 = s_A[1], A2 = s_A[2], A3 = s_A[3], A4 = s_A[4], A5 = s_A[5], A6 = s_A[6], A7 = s_A[7];
    double acc = 0.0;
    for (int i = 0; i < P.npts; i++) {
      const SiftPointD &pt = P.pts[i];
      const float x = pt.xpos, y = pt.ypos, mx = pt.match_xpos, my = pt.match_ypos;
      if (pt.score < P.min_score || pt.ambiguity > P.max_ambiguity) continue;
      const float den = A6 * x + A7 * y + 1.0f;
      const float dx = (A0 * x + A1 * y + A2) / den - mx;
      const float dy = (A3 * x + A4 * y + A5) / den - my;
      const float err = dx * dx + dy * dy;
      const float wei = (err < P.limit ? 1.0f : 0.0f);
      const double xd = x, yd = y;
      const double p6 = -x * mx, p7 = -y * mx, q6 = -x * my, q7 = -y * my;      // float products, then widened
      // Y1 = (x, y, 1, 0, 0, 0, p6, p7), Y2 = (0, 0, 0, x, y, 1, q6, q7)
      const double y1r = pick8(r, xd, yd, 1.0, 0.0, 0.0, 0.0, p6, p7), y2r = pick8(r, 0.0, 0.0, 0.0, xd, yd, 1.0, q6, q7);
      if (is_m) {
        const double y1c = pick8(c, xd, yd, 1.0, 0.0, 0.0, 0.0, p6, p7), y2c = pick8(c, 0.0, 0.0, 0.0, xd, yd, 1.0, q6, q7);
        acc += (y1c * y1r * wei);
        acc += (y2c * y2r * wei);
      } else if (is_x) {
        acc += y1r * mx * wei;
        acc += y2r * my * wei;
      }
    }
    if (is_m) { s_M[r * 8 + c] = acc; s_M[c * 8 + r] = acc; }
    if (is_x) s_X[r] = acc;
    __syncthreads();
    if (lane == 0) {                                    // cv::solve(M, X, A, DECOMP_CHOLESKY), geomFuncs.cpp:55
      double L[64], B[8];
      for (int k = 0; k < 64; k++) L[k] = s_M[k];
      for (int k = 0; k < 8; k++) B[k] = s_X[k];
      bool ok = true;
      for (int i = 0; i < 8 && ok; i++)
        for (int j = 0; j <= i; j++) {
          double s = L[i * 8 + j];
          for (int k = 0; k < j; k++) s -= L[i * 8 + k] * L[j * 8 + k];
          if (i == j) {
            if (!(s > 0)) { ok = false; break; }
            L[i * 8 + i] = sqrt(s);
          } else {
            L[i * 8 + j] = s / L[j * 8 + j];
          }
        }
      if (ok) {
        for (int i = 0; i < 8; i++) {
          double s = B[i];
          for (int k = 0; k < i; k++) s -= L[i * 8 + k] * B[k];
          B[i] = s / L[i * 8 + i];
        }
        for (int i = 7; i >= 0; i--) {
          double s = B[i];
          for (int k = i + 1; k < 8; k++) s -= L[k * 8 + i] * B[k];
          B[i] = s / L[i * 8 + i];
        }
        for (int k = 0; k < 8; k++) s_A[k] = B[k];
      } else {
        for (int k = 0; k < 8; k++) s_A[k] = 0.0;       // cv::solve zeroes the solution when the factorisation fails
      }
    }
    __syncthreads();
  }
  const double A0 = s_A[0], A1 = s_A[1], A2 = s_A[2], A3 = s_A[3], A4 = s_A[4], A5 = s_A[5], A6 = s_A[6], A7 = s_A[7];
  int numfit = 0;
  for (int i = lane; i < P.npts; i += 64) {
    SiftPointD &pt = pts_rw[i];
    const float x = pt.xpos, y = pt.ypos;
    const float den = A6 * x + A7 * y + 1.0;
    const float dx = (A0 * x + A1 * y + A2) / den - pt.match_xpos;
    const float dy = (A3 * x + A4 * y + A5) / den - pt.match_ypos;
    const float err = dx * dx + dy * dy;
    if (err < P.limit) numfit++;
    pt.match_error = sqrtf(err);
  }
  for (int m = 32; m > 0; m >>= 1) numfit += __shfl_xor(numfit, m, 64);
  if (lane < 8) result[lane] = (float)s_A[lane];
  if (lane == 0) {
    result[8] = 1.0f;
    reinterpret_cast<int *>(result)[9] = numfit;
  }
}

}  // namespace

extern "C" int misift_improve_homography(misift_ctx *ctx, void *d_pts, int npts, float *homography, int num_loops,
                                         float min_score, float max_ambiguity, float thresh, int *num_fit)
{
  if (!ctx || !homography || !num_fit || num_loops < 0 || npts < 0) {
    misift_set_error("misift_improve_homography: invalid argument");
    return MISIFT_EINVAL;
  }
  *num_fit = 0;
  if (!d_pts) return MISIFT_OK;                                    // geomFuncs.cpp:11-12
  int rc = misift_ensure_tmp(ctx, 64 * sizeof(float));
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  ImproveArgs P;
  P.pts = (const SiftPointD *)d_pts;
  P.npts = npts; P.num_loops = num_loops;
  P.min_score = min_score; P.max_ambiguity = max_ambiguity; P.limit = thresh * thresh;
  for (int i = 0; i < 8; i++) P.a0[i] = homography[i] / homography[8];      // float division (geomFuncs.cpp:20-21)
  float *result = reinterpret_cast<float *>(ctx->d_match_tmp);
  {
    LaunchScope ls(ctx, "improve_homography");
    hipLaunchKernelGGL(improve_homography_kernel, dim3(1), dim3(64), 0, ctx->stream, P, (SiftPointD *)d_pts, result);
    rc = ls.finish();
    if (rc) return rc;
  }
  float h[10];
  HIP_TRY(hipMemcpyAsync(h, result, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  memcpy(homography, h, 9 * sizeof(float));
  memcpy(num_fit, &h[9], sizeof(int));
  return MISIFT_OK;
}

// homography.hip — RANSAC homography over stored matches (SURVEY §8f "next" row 1).
//
// Replaces FindHomography (reference matching.cu:1000-1087) with its kernels
// ComputeHomographies (:907-948) and TestHomographies (:953-996).  Not on the
// roofline-graded hot path (latency-bound, a few k points); first cut runs the
// hypothesis generation and inlier counting on the host after one strided D2H of
// the six fields it needs.  Sampling uses libc rand() in the reference's call order
// (matching.cu:1041-1053) so hypothesis sets are reproducible the same way.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.hpp"

// Solve the 8x8 system A h = b (partial pivoting, fp32 like the reference's LU).
static bool solve8(float A[8][8], float b[8], float h[8])
{
  int perm[8];
  for (int i = 0; i < 8; i++) perm[i] = i;
  for (int c = 0; c < 8; c++) {
    int piv = c;
    float big = fabsf(A[c][c]);
    for (int r = c + 1; r < 8; r++)
      if (fabsf(A[r][c]) > big) { big = fabsf(A[r][c]); piv = r; }
    if (big == 0.0f) return false;
    if (piv != c) {
      for (int k = 0; k < 8; k++) { float t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
      float t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
    const float inv = 1.0f / A[c][c];
    for (int r = c + 1; r < 8; r++) {
      const float f = A[r][c] * inv;
      if (f == 0.0f) continue;
      for (int k = c; k < 8; k++) A[r][k] -= f * A[c][k];
      b[r] -= f * b[c];
    }
  }
  for (int r = 7; r >= 0; r--) {
    float s = b[r];
    for (int k = r + 1; k < 8; k++) s -= A[r][k] * h[k];
    h[r] = s / A[r][r];
  }
  return true;
}

extern "C" int misift_find_homography(misift_ctx *ctx, const void *d_pts, int npts, float *homography, int *num_matches,
                                      int num_loops, float min_score, float max_ambiguity, float thresh)
{
  if (!ctx || !homography || !num_matches) {
    misift_set_error("misift_find_homography: invalid argument");
    return MISIFT_EINVAL;
  }
  *num_matches = 0;
  homography[0] = homography[4] = homography[8] = 1.0f;
  homography[1] = homography[2] = homography[3] = 0.0f;
  homography[5] = homography[6] = homography[7] = 0.0f;
  if (!d_pts || npts < 8) return MISIFT_OK;                        // matching.cu:1008, :1016-1017
  num_loops = (num_loops + 15) / 16 * 16;
  // one strided read of xpos,ypos (offset 0), score,ambiguity (24) and match_xpos,match_ypos (36)
  std::vector<SiftPointD> h((size_t)npts);
  int rc = misift_download_fields(ctx, h.data(), d_pts, npts, 0, 2);
  if (rc) return rc;
  rc = misift_download_fields(ctx, h.data(), d_pts, npts, 24, 2);
  if (rc) return rc;
  rc = misift_download_fields(ctx, h.data(), d_pts, npts, 36, 2);
  if (rc) return rc;
  std::vector<int> valid;
  for (int i = 0; i < npts; i++)
    if (h[i].score > min_score && h[i].ambiguity < max_ambiguity) valid.push_back(i);
  const int numValid = (int)valid.size();
  if (numValid < 8) return MISIFT_OK;
  const float thresh2 = thresh * thresh;
  int bestCount = -1;
  float best[8] = {1, 0, 0, 0, 1, 0, 0, 0};
  for (int loop = 0; loop < num_loops; loop++) {
    int p1 = rand() % numValid;
    int p2 = rand() % numValid;
    int p3 = rand() % numValid;
    int p4 = rand() % numValid;
    while (p2 == p1) p2 = rand() % numValid;
    while (p3 == p1 || p3 == p2) p3 = rand() % numValid;
    while (p4 == p1 || p4 == p2 || p4 == p3) p4 = rand() % numValid;
    const int sel[4] = {valid[p1], valid[p2], valid[p3], valid[p4]};
    float A[8][8], b[8], hh[8];
    for (int i = 0; i < 4; i++) {
      const float x1 = h[sel[i]].xpos, y1 = h[sel[i]].ypos;
      const float x2 = h[sel[i]].match_xpos, y2 = h[sel[i]].match_ypos;
      float *r1 = A[2 * i], *r2 = A[2 * i + 1];
      r1[0] = x1; r1[1] = y1; r1[2] = 1.0f; r1[3] = r1[4] = r1[5] = 0.0f; r1[6] = -x2 * x1; r1[7] = -x2 * y1;
      r2[0] = r2[1] = r2[2] = 0.0f; r2[3] = x1; r2[4] = y1; r2[5] = 1.0f; r2[6] = -y2 * x1; r2[7] = -y2 * y1;
      b[2 * i] = x2;
      b[2 * i + 1] = y2;
    }
    if (!solve8(A, b, hh)) continue;
    int cnt = 0;
    for (int i = 0; i < npts; i++) {                              // TestHomographies, matching.cu:971-984
      const float x1 = h[i].xpos, y1 = h[i].ypos, x2 = h[i].match_xpos, y2 = h[i].match_ypos;
      const float nomx = hh[0] * x1 + hh[1] * y1 + hh[2];
      const float nomy = hh[3] * x1 + hh[4] * y1 + hh[5];
      const float deno = hh[6] * x1 + hh[7] * y1 + 1.0f;
      const float errx = x2 * deno - nomx;
      const float erry = y2 * deno - nomy;
      const float err2 = errx * errx + erry * erry;
      if (err2 < thresh2 * (deno * deno)) cnt++;
    }
    if (cnt > bestCount) {
      bestCount = cnt;
      memcpy(best, hh, sizeof(best));
    }
  }
  if (bestCount >= 0) {
    *num_matches = bestCount;
    memcpy(homography, best, sizeof(best));
    homography[8] = 1.0f;
  }
  return MISIFT_OK;
}

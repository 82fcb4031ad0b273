/* sift_oracle.c — CPU restatement of the CudaSift hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under oracle/ is part of the product: only tests/, the smoke check
 * in __graft_entry__.py and the `cpu_baseline` leg of bench.py may load this
 * library, and only as the checker / the timed CPU baseline.  The shipped path
 * (libmisift.so) never links, loads or falls back to it.
 *
 * What it restates (all citations are file:line in Celebrandil/CudaSift @ v1):
 *   taps            cudaSiftH.cu:316-323 (ScaleDown), :408-418 (LowPass), :439-458 (Laplace)
 *   LowPass         cudaSiftD.cu:1986-2037 (LowPassBlock; plainest form :1928-1950)
 *   ScaleDown       cudaSiftD.cu:84-168
 *   ScaleUp         cudaSiftD.cu:170-190
 *   LaplaceMulti    cudaSiftD.cu:1753-1793 (LaplaceMultiMem)
 *   FindPointsMulti cudaSiftD.cu:1292-1431 (FindPointsMultiNew)
 *   Orientations    cudaSiftD.cu:972-1057  (ComputeOrientationsCONST)
 *   Descriptors     cudaSiftD.cu:295-417   (FastAtan2 + ExtractSiftDescriptorsCONSTNew)
 *   ExtractSift     cudaSiftH.cu:72-232    (octave recursion, counter protocol, numPts rule)
 *   MatchSiftData   matching.cu:289-397, :1090-1206 (CleanMatches + FindMaxCorr10)
 *
 * PARITY PINNING (round 3: pinned).  The reference ships no golden vectors, known-answer tests or fixtures for
 * extraction and its CUDA cannot be built here — but its kernels and host code CAN be executed:
 * oracle/build_ref.sh compiles the reference's own cudaImage.cu, cudaSiftH.cu (+ cudaSiftD.cu) and matching.cu where
 * they lie against a CPU SIMT emulation of the CUDA subset they use (oracle/simt_emul.h) into
 * oracle/_ref/libcudasift_refemul_{fast,off}.so.  tests/test_refemul_cpu.py holds this restatement against that library
 * (all 17 counters and the keypoint set identical; separable filters, DoG planes, positions, match scores, homographies
 * the same bits in the contraction flavour; 1-2 ulp / 0.01 degrees where the reference calls hardware math functions),
 * and against tests/golden/refemul_golden.npz, vectors the emulated reference produced (they travel to the GPU box).
 * The MATCHER is additionally pinned to the reference's CPU routines MatchC1/MatchC3 (match.cu:57-130) and
 * ImproveHomography to the reference's own geomFuncs.cpp, both compiled by the same script.  The independent
 * cross-implementations of tests/test_oracle_cpu.py (scipy separable filters, scipy rank filters + numpy.linalg for
 * detection / refinement, float64 numpy for orientation and descriptor) stay as a second line.
 *
 * Arithmetic conventions (shared with the HIP kernels so the DoG pyramid and every discrete decision agree bit for
 * bit): IEEE fp32, compiled with -ffp-contract=off; the separable filters use the explicit fmaf chains written below —
 * conv9 where the reference accumulates a running sum (LaplaceMulti), conv9_expr where it writes one expression
 * (LowPass, ScaleDown: the left product fused, the right one rounded, as a contracting compiler does; bit-identical to
 * the reference source built with contraction); everything else is plain, uncontracted arithmetic — contract mode
 * ORC_CONTRACT_PLAIN, the one the HIP kernels implement and all parity tests use.
 *
 * ERROR BAR ON THAT CHOICE.  nvcc (-fmad=true by default) would also contract the
 * multiply-adds of the refinement, orientation and descriptor code.  Which ones
 * exactly is a compiler decision nobody can observe here, so a second mode,
 * ORC_CONTRACT_NVCC (orc_set_contract(1)), applies the LLVM/NVPTX DAG-combiner
 * rules to the reference expressions as written: fadd(fmul(a,b),c) -> fma(a,b,c)
 * taking the LEFT multiply first, fsub(fmul(a,b),c) -> fma(a,b,-c),
 * fsub(c,fmul(a,b)) -> fma(-a,b,c), multiplies reached through a phi are not
 * fused.  tests/test_oracle_cpu.py::test_contraction_sensitivity runs both modes
 * and bounds what the choice can change (profiles/r02_contraction_sensitivity.json,
 * DESIGN.md section 2).  Elementary functions: the reference calls CUDA's
 * atan2f / exp / __sinf / __cosf / __expf, none of which can be reproduced bit
 * for bit elsewhere; wherever a HARD decision or an 8-bit texture weight hangs on
 * one of them (orientation histogram: atan2f, expf; descriptor rotation: sin, cos;
 * descriptor Gaussian: expf) the oracle uses its own written-out fmaf-chain
 * versions det_atan2 / det_exp / det_sincos (accurate to ~1-2 ulp), and the HIP
 * kernels evaluate the identical chains: orientations agree bit for bit.  sqrtf
 * and the divisions are IEEE-exact on both sides; the keypoint scale uses det_exp2
 * (and powf(2, s/5) evaluated on the HOST by both sides).
 *
 * Deliberate deviations from the reference: #4 no 32
 * candidates-per-tile cap (overflows are counted in orc_stats), #7 FastAtan2(0,0)
 * -> 0 instead of NaN, #8 empty orientation histogram -> orientation 0,
 * #9/#10 selectable, #11 never writes past n1, #13 ScaleDown guarded, descriptor
 * votes landing at index >= 128 (angi==8 in the last cell) are dropped.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NUM_SCALES 5
#define LAPLACE_S  (NUM_SCALES + 3)

typedef struct {
  float xpos, ypos, scale, sharpness, edgeness, orientation, score, ambiguity;
  int32_t match;
  float match_xpos, match_ypos, match_error, subsampling;
  float empty[3];
  float data[128];
} SiftPoint;

typedef struct {
  long tile_overflows;   /* 30x8xscale tiles that held > 32 candidates (App. B #4) */
  long nan_guards;       /* descriptor samples with dx==dy==0 (App. B #7)          */
  long empty_hists;      /* orientation histograms without a peak (App. B #8)      */
  long oob_votes;        /* descriptor votes dropped at index >= 128                */
  long capacity_drops;   /* points dropped because maxPts was reached               */
} orc_stats_t;

static orc_stats_t g_stats;

/* contraction model for everything outside the separable filters (see header) */
#define ORC_CONTRACT_PLAIN 0
#define ORC_CONTRACT_NVCC  1
static int g_contract = ORC_CONTRACT_PLAIN;
static int g_refcap = 0;       /* 1 = keep the reference's 32-extrema-per-block cap (Appendix B #4); default: keep every extremum */
void orc_set_reference_cap(int on) { g_refcap = on ? 1 : 0; }
void orc_set_contract(int mode) { g_contract = mode ? ORC_CONTRACT_NVCC : ORC_CONTRACT_PLAIN; }
int orc_get_contract(void) { return g_contract; }
/* a*b + c, a*b - c*d, a*b + c*d + e*f as nvcc-style contraction would evaluate them */
static inline float mad(float a, float b, float c) { return g_contract ? fmaf(a, b, c) : a * b + c; }
static inline float mmsub(float a, float b, float c, float d) { return g_contract ? fmaf(a, b, -(c * d)) : a * b - c * d; }
static inline float dot3(float a, float b, float c, float d, float e, float f)
{
  return g_contract ? fmaf(e, f, fmaf(a, b, c * d)) : a * b + c * d + e * f;
}
void orc_stats_reset(void) { memset(&g_stats, 0, sizeof(g_stats)); }
void orc_stats_get(orc_stats_t *out) { *out = g_stats; }
int orc_sizeof_point(void) { return (int)sizeof(SiftPoint); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------ taps */

/* cudaSiftH.cu:408-418.  k[4] is the centre tap. */
void orc_lowpass_taps(float scale, float k[9])
{
  float kernelSum = 0.0f;
  float ivar2 = 1.0f / (2.0f * scale * scale);
  for (int j = -4; j <= 4; j++) {
    k[j + 4] = (float)expf((float)(-(double)j * j * ivar2));
    kernelSum += k[j + 4];
  }
  for (int j = -4; j <= 4; j++) k[j + 4] /= kernelSum;
}

/* cudaSiftH.cu:316-323.  k[2] is the centre tap. */
void orc_scaledown_taps(float variance, float k[5])
{
  float kernelSum = 0.0f;
  for (int j = 0; j < 5; j++) {
    k[j] = (float)expf((float)(-(double)(j - 2) * (j - 2) / 2.0 / variance));
    kernelSum += k[j];
  }
  for (int j = 0; j < 5; j++) k[j] /= kernelSum;
}

/* cudaSiftH.cu:439-458, recursive; table index [octave*12*16 + 16*scale + j],
 * j=0 is the centre tap.  `kernel` must hold 8*12*16 floats. */
void orc_laplace_taps_rec(int numOctaves, float initBlur, float *kernel)
{
  if (numOctaves > 1) {
    float totInitBlur = sqrtf(initBlur * initBlur + 0.5f * 0.5f) / 2.0f;
    orc_laplace_taps_rec(numOctaves - 1, totInitBlur, kernel);
  }
  float scale = powf(2.0f, -1.0f / NUM_SCALES);
  float diffScale = powf(2.0f, 1.0f / NUM_SCALES);
  for (int i = 0; i < NUM_SCALES + 3; i++) {
    float kernelSum = 0.0f;
    float var = scale * scale - initBlur * initBlur;
    for (int j = 0; j <= 4; j++) {
      kernel[numOctaves * 12 * 16 + 16 * i + j] = (float)expf((float)(-(double)j * j / 2.0 / var));
      kernelSum += (j == 0 ? 1 : 2) * kernel[numOctaves * 12 * 16 + 16 * i + j];
    }
    for (int j = 0; j <= 4; j++) kernel[numOctaves * 12 * 16 + 16 * i + j] /= kernelSum;
    scale *= diffScale;
  }
}

void orc_laplace_taps(int numOctaves, float *kernel)
{
  memset(kernel, 0, sizeof(float) * 8 * 12 * 16);
  orc_laplace_taps_rec(numOctaves, 0.0f, kernel);   /* cudaSiftH.cu:110 */
}

/* ---------------------------------------------------- per-thread work buffers
 * The big work buffers of an extraction (scratch arena 101 MB, Laplace intermediate 66 MB, filter rows 8 MB at 1080p)
 * are kept per thread and re-used from frame to frame.  malloc/free would serve them with mmap/munmap (they exceed the
 * 64 MB heaps of glibc's per-thread arenas whatever M_MMAP_THRESHOLD says), and with one frame per core every munmap
 * is a TLB shoot-down interrupt on all the cores the process runs on: that, not memory bandwidth, is what held the
 * frame-parallel CPU baseline to 0.2 frames/s per core on a 256-core host. */
#define ORC_TLS_SLOTS 4
static __thread void *tls_ptr[ORC_TLS_SLOTS];
static __thread size_t tls_cap[ORC_TLS_SLOTS];
static void *tls_buf(int slot, size_t bytes)
{
  if (tls_cap[slot] < bytes) {
    free(tls_ptr[slot]);
    tls_ptr[slot] = malloc(bytes);
    tls_cap[slot] = tls_ptr[slot] ? bytes : 0;
  }
  return tls_ptr[slot];
}
/* release the calling thread's buffers (the test-suite's sanitizer flavour calls it; worker threads keep theirs) */
void orc_release_thread_buffers(void)
{
  for (int i = 0; i < ORC_TLS_SLOTS; i++) { free(tls_ptr[i]); tls_ptr[i] = NULL; tls_cap[i] = 0; }
}

/* ---------------------------------------------------- separable filtering */

/* Symmetric 9-tap dot product, kc[0] = centre tap, p_j = (value at -j) + (value at +j).
 * This exact fmaf chain is the shared arithmetic contract with the HIP kernels. */
static inline float conv9(const float kc[5], float c, float p1, float p2, float p3, float p4)
{
  float s = kc[0] * c;
  s = fmaf(kc[1], p1, s);
  s = fmaf(kc[2], p2, s);
  s = fmaf(kc[3], p3, s);
  s = fmaf(kc[4], p4, s);
  return s;
}

/* The same dot product where the reference writes it as ONE expression `k4*c + k3*p1 + k2*p2 + k1*p3 + k0*p4`
 * (LowPassBlock, cudaSiftD.cu:2001-2005, :2022-2026) instead of a running sum: a contracting compiler (LLVM's DAG
 * combiner, which NVPTX uses under -fmad=true, and g++ -ffp-contract=fast alike) fuses the LEFT product of
 * `a*b + c*d` and rounds the right one, then fuses every later product into the running sum.  Pinned by the
 * reference's own source compiled that way (oracle/_ref/libcudasift_refemul_fast.so, tests/test_refemul_cpu.py). */
static inline float conv9_expr(const float kc[5], float c, float p1, float p2, float p3, float p4)
{
  float s = fmaf(kc[0], c, kc[1] * p1);
  s = fmaf(kc[2], p2, s);
  s = fmaf(kc[3], p3, s);
  s = fmaf(kc[4], p4, s);
  return s;
}

/* LowPass: horizontal pass first, then vertical; clamp-to-edge.
 * cudaSiftD.cu:1999-2005 (horizontal), :2022-2026 (vertical). */
void orc_lowpass(const float *src, int w, int h, int spitch, float *dst, int dpitch, float sigma)
{
  float k9[9], kc[5];
  orc_lowpass_taps(sigma, k9);
  for (int j = 0; j <= 4; j++) kc[j] = k9[4 - j];   /* kc[0]=k[4] centre … kc[4]=k[0] */
  float *tmp = (float *)tls_buf(0, sizeof(float) * (size_t)w * h);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    const float *r = src + (size_t)y * spitch;
    for (int x = 0; x < w; x++) {
#define R(d) r[clampi(x + (d), 0, w - 1)]
      tmp[(size_t)y * w + x] = conv9_expr(kc, R(0), R(1) + R(-1), R(2) + R(-2), R(3) + R(-3), R(4) + R(-4));
#undef R
    }
  }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
#define T(d) tmp[(size_t)clampi(y + (d), 0, h - 1) * w + x]
      dst[(size_t)y * dpitch + x] = conv9_expr(kc, T(0), T(-1) + T(1), T(-2) + T(2), T(-3) + T(3), T(-4) + T(4));
#undef T
    }
  }
}

/* ScaleDown: 5-tap Gaussian (variance 0.5) + 2x decimation, horizontal then vertical.
 * cudaSiftD.cu:123 (row filter) and :125 (column filter).  dst is (w/2, h/2). */
void orc_scaledown(const float *src, int w, int h, int spitch, float *dst, int dpitch)
{
  float k[5];
  orc_scaledown_taps(0.5f, k);
  const float k0 = k[0], k1 = k[1], k2 = k[2];
  int w2 = w / 2, h2 = h / 2;
  float *tmp = (float *)tls_buf(1, sizeof(float) * (size_t)(w2 > 0 ? w2 : 1) * h);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    const float *r = src + (size_t)y * spitch;
    for (int x = 0; x < w2; x++) {
#define R(m) r[clampi(2 * x + (m) - 2, 0, w - 1)]
      float s = fmaf(k0, R(0) + R(4), k1 * (R(1) + R(3)));   /* one expression: left product fused, see conv9_expr */
      s = fmaf(k2, R(2), s);
      tmp[(size_t)y * w2 + x] = s;
#undef R
    }
  }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h2; y++) {
    for (int x = 0; x < w2; x++) {
#define T(m) tmp[(size_t)clampi(2 * y + (m) - 2, 0, h - 1) * w2 + x]
      float s = fmaf(k2, T(2), k0 * (T(0) + T(4)));
      s = fmaf(k1, T(1) + T(3), s);
      dst[(size_t)y * dpitch + x] = s;
#undef T
    }
  }
}

/* ScaleUp: cudaSiftD.cu:170-190.  dst is (2w, 2h). */
void orc_scaleup(const float *src, int w, int h, int spitch, float *dst, int dpitch)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int xr = x + 1 < w ? x + 1 : w - 1, yd = y + 1 < h ? y + 1 : h - 1;
      float vul = src[(size_t)y * spitch + x], vur = src[(size_t)y * spitch + xr];
      float vdl = src[(size_t)yd * spitch + x], vdr = src[(size_t)yd * spitch + xr];
      dst[(size_t)(2 * y) * dpitch + 2 * x] = vul;
      dst[(size_t)(2 * y) * dpitch + 2 * x + 1] = 0.50f * (vul + vur);
      dst[(size_t)(2 * y + 1) * dpitch + 2 * x] = 0.50f * (vul + vdl);
      dst[(size_t)(2 * y + 1) * dpitch + 2 * x + 1] = 0.25f * (vul + vur + vdl + vdr);
    }
}

/* LaplaceMulti: 8 blurs of the same base (vertical pass first, then horizontal),
 * 7 differences.  cudaSiftD.cu:1762-1774 (vertical), :1777-1791 (horizontal + DoG).
 * taps: the 8*12*16 table; octave: reference octave index.  dog: 7 planes, plane
 * stride h*pitch, row stride pitch. */
void orc_laplace(const float *base, int w, int h, int pitch, const float *taps, int octave, float *dog)
{
  const float *kt = taps + octave * 12 * 16;
  float *vbuf = (float *)tls_buf(2, sizeof(float) * (size_t)LAPLACE_S * w * h);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
#define B(d) base[(size_t)clampi(y + (d), 0, h - 1) * pitch + x]
      float c = B(0), p1 = B(-1) + B(1), p2 = B(-2) + B(2), p3 = B(-3) + B(3), p4 = B(-4) + B(4);
#undef B
      for (int s = 0; s < LAPLACE_S; s++)
        vbuf[((size_t)s * h + y) * w + x] = conv9(kt + 16 * s, c, p1, p2, p3, p4);
    }
  }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float old = 0.0f;
      for (int s = 0; s < LAPLACE_S; s++) {
        const float *v = vbuf + ((size_t)s * h + y) * w;
#define V(d) v[clampi(x + (d), 0, w - 1)]
        float res = conv9(kt + 16 * s, V(0), V(-1) + V(1), V(-2) + V(2), V(-3) + V(3), V(-4) + V(4));
#undef V
        if (s > 0) dog[((size_t)(s - 1) * h + y) * pitch + x] = res - old;
        old = res;
      }
    }
  }
}

/* 2^x as a written-out fmaf chain (cephes exp2f: split off the nearest integer, degree-6 kernel on [-0.5, 0.5], scale
 * by 2^n; ~1.5 ulp), IDENTICAL in kernels_dog.hip.  The keypoint scale 2^(s/5) * exp2f(pds/5) (cudaSiftD.cu:1417)
 * feeds the orientation window's Gaussian and the descriptor's sample spacing, i.e. hard decisions downstream: with
 * libm exp2f here and ocml exp2f on the GPU the scales differed in the last bit for a few keypoints per frame, and
 * with them, occasionally, an orientation bin or an 8-bit texture weight. */
static inline float det_exp2(float x)
{
  if (x < -125.0f) return 0.0f;
  if (x > 126.0f) x = 126.0f;
  const float n = rintf(x);
  const float r = x - n;                                    /* exact */
  float p = fmaf(1.535336188319500e-4f, r, 1.339887440266574e-3f);
  p = fmaf(p, r, 9.618437357674640e-3f);
  p = fmaf(p, r, 5.550332471162809e-2f);
  p = fmaf(p, r, 2.402264791363012e-1f);
  p = fmaf(p, r, 6.931472028550421e-1f);
  p = fmaf(p, r, 1.0f);
  union { float f; int32_t i; } sc;
  sc.i = ((int32_t)n + 127) << 23;
  return p * sc.f;
}
float orc_det_exp2(float x) { return det_exp2(x); }

/* ------------------------------------------------------------ FindPoints */

/* FindPointsMultiNew, cudaSiftD.cu:1292-1431.  Appends records starting at
 * index *count (counter protocol handled by the caller); returns the number of
 * detections (before the capacity clamp).  Scan order: scale, row, column. */
typedef struct { float xpos, ypos, scale, sharpness, edgeness; } orc_det_t;
typedef struct { orc_det_t *d; int n, cap; } orc_detlist_t;

int orc_findpoints(const float *dog, int w, int h, int pitch, float thresh, float edgeLimit,
                   float factor, float lowestScale, float subsampling, SiftPoint *pts,
                   int start, int maxPts)
{
  const size_t size = (size_t)pitch * h;
  /* overflow statistics for the reference's 30x8 tiles (Appendix B #4) */
  int tilesx = (w + 29) / 30, tilesy = (h + 7) / 8;
  int *tilecnt = (int *)calloc((size_t)tilesx * tilesy * NUM_SCALES, sizeof(int));
  /* Rows are scanned in parallel (8-row blocks = one tile row, so tile counters are private to a block);
   * every block keeps its detections in scan order and the blocks are concatenated in (scale, row) order
   * afterwards, so the output order is the serial one: scale, row, column. */
  const int nblk = tilesy;
  orc_detlist_t *lists = (orc_detlist_t *)calloc((size_t)nblk * NUM_SCALES, sizeof(orc_detlist_t));
#pragma omp parallel for schedule(dynamic, 1)
  for (int job = 0; job < nblk * NUM_SCALES; job++) {
    const int s = job / nblk, yb = job % nblk;
    orc_detlist_t *L = &lists[job];
    const float *d0 = dog + size * s, *d1 = dog + size * (s + 1), *d2 = dog + size * (s + 2);
    /* pass 1: the 26-neighbour test of every pixel of this 8-row band (cudaSiftD.cu:1337-1366) */
    unsigned char *isext = (unsigned char *)calloc((size_t)8 * w, 1);
    for (int y = 8 * yb; y < 8 * yb + 8 && y < h; y++) {
      int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
      for (int x = 0; x < w; x++) {
        float v = d1[(size_t)y * pitch + x];
        if (!(fabsf(v) > thresh)) continue;
        int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
        const int xs[3] = {xm, x, xp}, ys[3] = {ym, y, yp};
        const float *pl[3] = {d0, d1, d2};
        float minv = INFINITY, maxv = -INFINITY;
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++)
            for (int c = 0; c < 3; c++) {
              if (a == 1 && b == 1 && c == 1) continue;    /* the 26 neighbours, clamped coords */
              float t = pl[a][(size_t)ys[b] * pitch + xs[c]];
              minv = fminf(minv, t);
              maxv = fmaxf(maxv, t);
            }
        if (!((v < fminf(-thresh, minv)) || (v > fmaxf(thresh, maxv)))) continue;
        isext[(size_t)(y - 8 * yb) * w + x] = 1;
        tilecnt[((size_t)s * tilesy + y / 8) * tilesx + x / 30]++;
      }
    }
    /* the reference's cap (cudaSiftD.cu:1369-1377, off unless orc_set_reference_cap(1)): a block of FindPointsMultiNew —
     * 30 columns x 8 rows of one scale — hands at most 32 extrema on to the refinement: the first 32 in the order of its
     * prefix sum over the threads, i.e. by column, then by row */
    if (g_refcap)
      for (int tx0 = 0; tx0 < w; tx0 += 30) {
        int kept = 0;
        for (int x = tx0; x < tx0 + 30 && x < w; x++)
          for (int yy = 0; yy < 8; yy++)
            if (isext[(size_t)yy * w + x] && ++kept > 32) isext[(size_t)yy * w + x] = 0;
      }
    /* pass 2: edge test, refinement, append — in scan order (row, column) */
    for (int y = 8 * yb; y < 8 * yb + 8 && y < h; y++) {
      for (int x = 0; x < w; x++) {
        if (!isext[(size_t)(y - 8 * yb) * w + x]) continue;
        /* refinement, cudaSiftD.cu:1383-1417: unclamped neighbour reads (x,y interior here) */
        const float *data1 = d1 + (size_t)y * pitch + x;
        float val = data1[0];
        float dxx = 2.0f * val - data1[-1] - data1[1];
        float dyy = 2.0f * val - data1[-pitch] - data1[pitch];
        float dxy = 0.25f * (data1[+pitch + 1] + data1[-pitch - 1] - data1[-pitch + 1] - data1[+pitch - 1]);
        float tra = dxx + dyy;
        float det = mmsub(dxx, dyy, dxy, dxy);
        if (!(tra * tra < edgeLimit * det)) continue;
        float edge = (tra * tra) / det;
        float dx = 0.5f * (data1[1] - data1[-1]);
        float dy = 0.5f * (data1[pitch] - data1[-pitch]);
        const float *data0 = d0 + (size_t)y * pitch + x;
        const float *data2 = d2 + (size_t)y * pitch + x;
        float ds = 0.5f * (data0[0] - data2[0]);
        float dss = 2.0f * val - data2[0] - data0[0];
        float dxs = 0.25f * (data2[1] + data0[-1] - data0[1] - data2[-1]);
        float dys = 0.25f * (data2[pitch] + data0[-pitch] - data2[-pitch] - data0[pitch]);
        float idxx = mmsub(dyy, dss, dys, dys);
        float idxy = mmsub(dys, dxs, dxy, dss);
        float idxs = mmsub(dxy, dys, dyy, dxs);
        float idet = 1.0f / dot3(idxx, dxx, idxy, dxy, idxs, dxs);
        float idyy = mmsub(dxx, dss, dxs, dxs);
        float idys = mmsub(dxy, dxs, dxx, dys);
        float idss = mmsub(dxx, dyy, dxy, dxy);
        float pdx = idet * dot3(idxx, dx, idxy, dy, idxs, ds);
        float pdy = idet * dot3(idxy, dx, idyy, dy, idys, ds);
        float pds = idet * dot3(idxs, dx, idys, dy, idss, ds);
        if (pdx < -0.5f || pdx > 0.5f || pdy < -0.5f || pdy > 0.5f || pds < -0.5f || pds > 0.5f) {
          pdx = dx / dxx;
          pdy = dy / dyy;
          pds = ds / dss;
        }
        float dsum = dot3(dx, pdx, dy, pdy, ds, pds);
        float dval = 0.5f * dsum;
        float sc = powf(2.0f, (float)s / NUM_SCALES) * det_exp2(pds * factor);
        if (!(sc >= lowestScale)) continue;
        if (L->n == L->cap) {
          L->cap = L->cap ? 2 * L->cap : 64;
          L->d = (orc_det_t *)realloc(L->d, sizeof(orc_det_t) * (size_t)L->cap);
        }
        orc_det_t *q = &L->d[L->n++];
        q->xpos = x + pdx;
        q->ypos = y + pdy;
        q->scale = sc;
        q->sharpness = g_contract ? fmaf(0.5f, dsum, val) : val + dval;
        q->edgeness = edge;
      }
    }
    free(isext);
  }
  int n = 0;
  for (int job = 0; job < nblk * NUM_SCALES; job++) {
    for (int i = 0; i < lists[job].n; i++) {
      int idx = start + n;
      n++;
      if (idx >= maxPts) {
#pragma omp atomic
        g_stats.capacity_drops++;
        continue;
      }
      const orc_det_t *q = &lists[job].d[i];
      SiftPoint *p = &pts[idx];
      p->xpos = q->xpos;
      p->ypos = q->ypos;
      p->scale = q->scale;
      p->sharpness = q->sharpness;
      p->edgeness = q->edgeness;
      p->subsampling = subsampling;
    }
    free(lists[job].d);
  }
  free(lists);
  for (size_t i = 0; i < (size_t)tilesx * tilesy * NUM_SCALES; i++)
    if (tilecnt[i] > 32) {
#pragma omp atomic
      g_stats.tile_overflows++;
    }
  free(tilecnt);
  return n;
}

/* ------------------------------------------------ texture-fetch emulation */

/* tex2D<float>(x, y) of a pitch2D texture: unnormalised coordinates, clamp
 * addressing, linear filtering (cudaSiftH.cu:196-205).  fracbits==8 rounds the
 * interpolation weights to 8 fractional bits like the CUDA texture unit;
 * anything else keeps full fp32 weights.  Shared arithmetic contract with HIP. */
static inline float tex2d(const float *img, int w, int h, int pitch, float x, float y, int fracbits)
{
  float xb = x - 0.5f, yb = y - 0.5f;
  float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (fracbits == 8) {
    /* 8 fractional bits, round to nearest, ties to even (CUDA documents the 1.8 fixed-point format of the filter
     * weights, not how the fraction is rounded into it; nearest-even is one add and one subtract on the GPU) */
    a = rintf(a * 256.0f) * (1.0f / 256.0f);
    b = rintf(b * 256.0f) * (1.0f / 256.0f);
  }
  /* clamp in float first so far-away coordinates cannot overflow the int cast */
  fx = fminf(fmaxf(fx, -2.0f), (float)w);
  fy = fminf(fmaxf(fy, -2.0f), (float)h);
  int ix = (int)fx, iy = (int)fy;
  int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
  int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
  float t00 = img[(size_t)y0 * pitch + x0], t10 = img[(size_t)y0 * pitch + x1];
  float t01 = img[(size_t)y1 * pitch + x0], t11 = img[(size_t)y1 * pitch + x1];
  float ia = 1.0f - a, ib = 1.0f - b;
  float v = (ia * ib) * t00;
  v = fmaf(a * ib, t10, v);
  v = fmaf(ia * b, t01, v);
  v = fmaf(a * b, t11, v);
  return v;
}

float orc_tex2d(const float *img, int w, int h, int pitch, float x, float y, int fracbits)
{
  return tex2d(img, w, h, pitch, x, y, fracbits);
}

/* ---------------------------------------------------- written-out elementary functions
 * The orientation histogram takes HARD decisions on atan2f (which of 32 bins) and on sums weighted by expf (which bins
 * are peaks), and the resulting angle positions every descriptor sample against the 8-bit texture-weight grid.  Two
 * different libm's (glibc here, ocml on the GPU) agree to an ulp, which is enough to flip such a decision once in a
 * few hundred keypoints.  The reference's own CUDA versions cannot be reproduced bit for bit anywhere else, so the
 * oracle states them explicitly instead — accurate to ~1-2 ulp like the CUDA libm ones — and kernels_points.hip
 * evaluates the IDENTICAL fmaf chains: orientations then agree bit for bit and descriptors to summation order. */
static inline float det_atan2(float y, float x)            /* cephes atanf kernel on [0,1] + octant fix-ups */
{
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = mx == 0.0f ? 0.0f : mn / mx;                   /* atan2f(0,0) = 0 */
  float base = 0.0f;
  if (a > 0.414213562f) { base = 0.785398163f; a = (a - 1.0f) / (a + 1.0f); }
  const float z = a * a;
  float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  float r = base + fmaf(p * z, a, a);
  r = ay > ax ? 1.57079637f - r : r;
  r = x < 0.0f ? 3.14159274f - r : r;
  return y < 0.0f ? -r : r;
}
static inline float det_exp(float x)                       /* cephes expf: x = n ln2 + r, degree-5 kernel, scale by 2^n */
{
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  const float n = rintf(x * 1.44269504f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float e = fmaf(p * r, r, r) + 1.0f;
  union { float f; int32_t i; } sc;
  sc.i = ((int32_t)n + 127) << 23;                          /* 2^n, n in [-126, 127] */
  return e * sc.f;
}
float orc_det_atan2(float y, float x) { return det_atan2(y, x); }
float orc_det_exp(float x) { return det_exp(x); }

/* ----------------------------------------------------------- orientation */

/* ComputeOrientationsCONST, cudaSiftD.cu:972-1057, for points [first,last).
 * Duplicates are appended at *dupCount (incremented; dropped when >= maxPts).
 * Returns nothing; orientation written in place. */
void orc_orientations(const float *img, int w, int h, int pitch, SiftPoint *pts, int first,
                      int last, unsigned int *dupCount, int maxPts, int fracbits)
{
  /* keypoints are independent: evaluate them in parallel, then append the second-orientation duplicates
   * serially in keypoint order (the order a serial run of the loop below would produce) */
  const int np = last > first ? last - first : 0;
  float *ori2 = (float *)malloc(sizeof(float) * (size_t)(np > 0 ? np : 1));
  unsigned char *has2 = (unsigned char *)calloc((size_t)(np > 0 ? np : 1), 1);
  long empties = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : empties)
  for (int bx = first; bx < last; bx++) {
    SiftPoint *p = &pts[bx];
    float hist[64];
    float gauss[11];
    float i2sigma2 = -1.0f / (2.0f * 1.5f * 1.5f * p->scale * p->scale);
    for (int t = 0; t < 11; t++) gauss[t] = det_exp(i2sigma2 * (t - 5) * (t - 5));
    for (int t = 0; t < 64; t++) hist[t] = 0.0f;
    /* The reference adds the 121 weights with shared-memory atomics (cudaSiftD.cu:1013), i.e. in no defined order; the
     * order fixed here — samples 0..63 and 64..120 summed separately, in index order, then the two partial sums added —
     * is the one the HIP kernel's two half-wavefront sums produce, so the histograms agree bit for bit. */
    float hist2[32];
    for (int t = 0; t < 32; t++) hist2[t] = 0.0f;
    float xp = p->xpos - 4.5f;
    float yp = p->ypos - 4.5f;
    for (int tx = 0; tx < 121; tx++) {
      int yd = tx / 11;
      int xd = tx - yd * 11;
      float xf = xp + xd;
      float yf = yp + yd;
      float dx = tex2d(img, w, h, pitch, xf + 1.0f, yf, fracbits) - tex2d(img, w, h, pitch, xf - 1.0f, yf, fracbits);
      float dy = tex2d(img, w, h, pitch, xf, yf + 1.0f, fracbits) - tex2d(img, w, h, pitch, xf, yf - 1.0f, fracbits);
      int bin = (int)(16.0f * det_atan2(dy, dx) / 3.1416f + 16.5f);
      if (bin > 31) bin = 0;
      float grad = sqrtf(mad(dx, dx, dy * dy));
      if (tx < 64) hist[bin] += grad * gauss[xd] * gauss[yd];
      else hist2[bin] += grad * gauss[xd] * gauss[yd];
    }
    for (int t = 0; t < 32; t++) hist[t] += hist2[t];
    for (int tx = 0; tx < 32; tx++) {
      int x1m = (tx >= 1 ? tx - 1 : tx + 31), x1p = (tx <= 30 ? tx + 1 : tx - 31);
      int x2m = (tx >= 2 ? tx - 2 : tx + 30), x2p = (tx <= 29 ? tx + 2 : tx - 30);
      hist[tx + 32] = mad(6.0f, hist[tx], 4.0f * (hist[x1m] + hist[x1p])) + (hist[x2m] + hist[x2p]);
    }
    for (int tx = 0; tx < 32; tx++) {
      int x1m = (tx >= 1 ? tx - 1 : tx + 31), x1p = (tx <= 30 ? tx + 1 : tx - 31);
      float v = hist[32 + tx];
      hist[tx] = (v > hist[32 + x1m] && v >= hist[32 + x1p] ? v : 0.0f);
    }
    float maxval1 = 0.0f, maxval2 = 0.0f;
    int i1 = -1, i2 = -1;
    for (int i = 0; i < 32; i++) {
      float v = hist[i];
      if (v > maxval1) {
        maxval2 = maxval1; maxval1 = v; i2 = i1; i1 = i;
      } else if (v > maxval2) {
        maxval2 = v; i2 = i;
      }
    }
    if (i1 < 0) {                       /* Appendix B #8 */
      empties++;
      p->orientation = 0.0f;
      continue;
    }
    {
      float val1 = hist[32 + ((i1 + 1) & 31)];
      float val2 = hist[32 + ((i1 + 31) & 31)];
      float peak = i1 + 0.5f * (val1 - val2) / (2.0f * maxval1 - val1 - val2);
      p->orientation = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
    }
    if (maxval2 > 0.8f * maxval1) {
      float val1 = hist[32 + ((i2 + 1) & 31)];
      float val2 = hist[32 + ((i2 + 31) & 31)];
      float peak = i2 + 0.5f * (val1 - val2) / (2.0f * maxval2 - val1 - val2);
      has2[bx - first] = 1;
      ori2[bx - first] = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
    }
  }
#pragma omp atomic
  g_stats.empty_hists += empties;
  for (int bx = first; bx < last; bx++) {
    if (!has2[bx - first]) continue;
    const SiftPoint *p = &pts[bx];
    unsigned int idx = (*dupCount)++;
    if (idx < (unsigned int)maxPts) {
      SiftPoint *q = &pts[idx];
      q->xpos = p->xpos; q->ypos = p->ypos; q->scale = p->scale;
      q->sharpness = p->sharpness; q->edgeness = p->edgeness;
      q->orientation = ori2[bx - first];
      q->subsampling = p->subsampling;
    } else {
#pragma omp atomic
      g_stats.capacity_drops++;
    }
  }
  free(ori2);
  free(has2);
}

/* ------------------------------------------------------------ descriptors */

/* FastAtan2, cudaSiftD.cu:295-306, with the (0,0) guard of Appendix B #7. */
static inline float fast_atan2(float y, float x)
{
  float absx = fabsf(x), absy = fabsf(y);
  float mx = fmaxf(absx, absy), mn = fminf(absx, absy);
  if (mx == 0.0f) return 0.0f;
  float a = mn / mx;
  float s = a * a;
  float r;
  if (g_contract) r = fmaf(fmaf(fmaf(-0.0464964749f, s, 0.15931422f), s, -0.327622764f) * s, a, a);
  else r = ((-0.0464964749f * s + 0.15931422f) * s - 0.327622764f) * s * a + a;
  r = (absy > absx ? 1.57079637f - r : r);
  r = (x < 0 ? 3.14159274f - r : r);
  r = (y < 0 ? -r : r);
  return r;
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }

/* sin and cos of the descriptor's rotation angle theta in [0, 2 pi].  The reference uses the CUDA hardware
 * approximations __sinf / __cosf (cudaSiftD.cu:331-332, absolute error ~2^-21.4), which no other platform reproduces bit
 * for bit; any ACCURATE sine is therefore an equally faithful restatement.  This one is written out — Cody-Waite
 * reduction by pi/2 in two fused steps, the cephes single-precision minimax kernels on [-pi/4, pi/4] as explicit fmaf
 * chains (max error 1 ulp) — so that the HIP kernels evaluate the IDENTICAL expression (kernels_points.hip
 * det_sincos): the 1024 sample coordinates of a descriptor then agree bit for bit between the two, and with them
 * every 8-bit texture weight.  (With libm sinf on one side and ocml sinf on the other, a last-bit difference moved
 * a weight by 1/256 in ~0.4 % of the descriptors.) */
static inline void det_sincos(float x, float *sn, float *cs)
{
  const float kf = rintf(x * 0.636619747f);                  /* nearest multiple of pi/2 */
  float r = fmaf(kf, -1.57079625f, x);                       /* pi/2 = 1.57079625 + 7.54978942e-08 (+ ...) */
  r = fmaf(kf, -7.54978942e-08f, r);
  const float z = r * r;
  float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float s = fmaf(ps * z, r, r);
  float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)kf & 3;
  const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}
void orc_det_sincos(float x, float *sn, float *cs) { det_sincos(x, sn, cs); }
/* array form for the accuracy tests: fn 0 = exp2(x), 1 = atan2(y, x), 2 = exp(x), 3 = sincos(x) -> out, out2 */
void orc_det_eval(int fn, const float *x, const float *y, float *out, float *out2, long n)
{
  for (long i = 0; i < n; i++) {
    if (fn == 0) out[i] = det_exp2(x[i]);
    else if (fn == 1) out[i] = det_atan2(y[i], x[i]);
    else if (fn == 2) out[i] = det_exp(x[i]);
    else det_sincos(x[i], &out[i], &out2[i]);
  }
}

/* ExtractSiftDescriptorsCONSTNew, cudaSiftD.cu:308-417, for points [first,last). */
void orc_descriptors(const float *img, int w, int h, int pitch, SiftPoint *pts, int first, int last,
                     float subsampling, int fracbits)
{
  float gauss[16];
  for (int t = 0; t < 16; t++) gauss[t] = det_exp(-(t - 7.5f) * (t - 7.5f) / 128.0f);
#pragma omp parallel for schedule(dynamic, 16)
  for (int bx = first; bx < last; bx++) {
    SiftPoint *p = &pts[bx];
    float buffer[128];
    long guards = 0, oob = 0;
    for (int i = 0; i < 128; i++) buffer[i] = 0.0f;
    float theta = 2.0f * 3.1415f / 360.0f * p->orientation;
    float sina, cosa;
    det_sincos(theta, &sina, &cosa);
    float scale = 12.0f / 16.0f * p->scale;
    float ssina = scale * sina;
    float scosa = scale * cosa;
    for (int y = 0; y < 16; y++) {
      for (int tx = 0; tx < 16; tx++) {
        float xpos, ypos;
        if (g_contract) {
          xpos = fmaf(-(y - 7.5f), ssina, fmaf(tx - 7.5f, scosa, p->xpos)) + 0.5f;
          ypos = fmaf(y - 7.5f, scosa, fmaf(tx - 7.5f, ssina, p->ypos)) + 0.5f;
        } else {
          xpos = p->xpos + (tx - 7.5f) * scosa - (y - 7.5f) * ssina + 0.5f;
          ypos = p->ypos + (tx - 7.5f) * ssina + (y - 7.5f) * scosa + 0.5f;
        }
        float dx = tex2d(img, w, h, pitch, xpos + cosa, ypos + sina, fracbits) -
                   tex2d(img, w, h, pitch, xpos - cosa, ypos - sina, fracbits);
        float dy = tex2d(img, w, h, pitch, xpos - sina, ypos + cosa, fracbits) -
                   tex2d(img, w, h, pitch, xpos + sina, ypos - cosa, fracbits);
        float grad = gauss[y] * gauss[tx] * sqrtf(mad(dx, dx, dy * dy));
        if (dx == 0.0f && dy == 0.0f) guards++;
        float angf = mad(4.0f / 3.1415f, fast_atan2(dy, dx), 4.0f);

        int hori = (tx + 2) / 4 - 1;
        float horf = (tx - 1.5f) / 4.0f - hori;
        float ihorf = 1.0f - horf;
        int veri = (y + 2) / 4 - 1;
        float verf = (y - 1.5f) / 4.0f - veri;
        float iverf = 1.0f - verf;
        int angi = (int)angf;
        int angp = (angi < 7 ? angi + 1 : 0);
        angf -= angi;
        float iangf = 1.0f - angf;

        int hist = 8 * (4 * veri + hori);
        int p1 = angi + hist;
        int p2 = angp + hist;
#define VOTE(idx, val) do { int i_ = (idx); if (i_ >= 0 && i_ < 128) buffer[i_] += (val); else oob++; } while (0)
        if (tx >= 2) {
          float grad1 = ihorf * grad;
          if (y >= 2) {
            float grad2 = iverf * grad1;
            VOTE(p1, iangf * grad2);
            VOTE(p2, angf * grad2);
          }
          if (y <= 13) {
            float grad2 = verf * grad1;
            VOTE(p1 + 32, iangf * grad2);
            VOTE(p2 + 32, angf * grad2);
          }
        }
        if (tx <= 13) {
          float grad1 = horf * grad;
          if (y >= 2) {
            float grad2 = iverf * grad1;
            VOTE(p1 + 8, iangf * grad2);
            VOTE(p2 + 8, angf * grad2);
          }
          if (y <= 13) {
            float grad2 = verf * grad1;
            VOTE(p1 + 40, iangf * grad2);
            VOTE(p2 + 40, angf * grad2);
          }
        }
#undef VOTE
      }
    }
    /* normalise, clamp at 0.2, normalise again (cudaSiftD.cu:390-409) */
    float sum = 0.0f;
    for (int i = 0; i < 128; i++) sum += buffer[i] * buffer[i];
    float rs = 1.0f / sqrtf(sum);
    float sum2 = 0.0f;
    for (int i = 0; i < 128; i++) {
      buffer[i] = fminf(buffer[i] * rs, 0.2f);
      sum2 += buffer[i] * buffer[i];
    }
    float rs2 = 1.0f / sqrtf(sum2);
    for (int i = 0; i < 128; i++) p->data[i] = buffer[i] * rs2;
    p->xpos *= subsampling;
    p->ypos *= subsampling;
    p->scale *= subsampling;
    if (guards || oob) {
#pragma omp critical
      { g_stats.nan_guards += guards; g_stats.oob_votes += oob; }
    }
  }
}

/* ------------------------------------------------ descriptor tail: a per-record BOUND (test infrastructure)
 * Against the reference's own code (oracle/_ref, the CPU SIMT emulator) ~0.7 % of the descriptors differ by more than
 * 1e-4 in some element.  The mechanism: the two sides compute a sample coordinate with different
 * roundings (libm vs written-out sincos, contracted vs plain multiply-adds) — a last-bit difference — and the texture
 * unit rounds the interpolation weight to 8 fractional bits, so a fetch whose weight sits within that last bit of a
 * rounding tie moves by 1/256 of the local texel difference.  This function turns the explanation into a bound per
 * record and element: for every one of the 1024 fetches of a descriptor it checks whether a coordinate perturbation of
 * `ulps` units in the last place can change the rounded weight, takes (1/256) x the texel difference along that axis
 * as the fetch's possible change, and propagates it through gradient magnitude, angle split, the spatial vote weights
 * and the two normalisations.  A regression of the same size as the tail but NOT of this origin exceeds the bound.
 *   img/w/h/pitch : the octave image the descriptor was sampled from;  p : the record at THAT octave's scale
 *   bound[128]    : out, largest |change| of each element;  returns the number of fetches that can flip, and in
 *   *wraps the number of samples whose angle sits on the angi = 8 <-> 0 seam (cudaSiftD.cu:353, Appendix B #6).
 *   extra         : further displacement of the sample grid between the two sides, in pixels of this level — the two
 *                   records' orientations may differ by a few ulp (libm vs written-out atan2 in the histogram peak), which
 *                   turns the whole grid: |d theta| x the grid's radius */
static int ialign_up(int a, int b);
static inline float ulp32(float x)
{
  x = fabsf(x);
  if (x < 1.0f) x = 1.0f;
  int e;
  frexpf(x, &e);
  return ldexpf(1.0f, e - 24);
}

static float fetch_flip(const float *img, int w, int h, int pitch, float x, float y, float ulps, float extra, int *nflip)
{
  float xb = x - 0.5f, yb = y - 0.5f;
  float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  float a256 = a * 256.0f, b256 = b * 256.0f;
  float da = fabsf(a256 - floorf(a256) - 0.5f), db = fabsf(b256 - floorf(b256) - 0.5f);
  float ra = rintf(a256) * (1.0f / 256.0f), rb = rintf(b256) * (1.0f / 256.0f);
  fx = fminf(fmaxf(fx, -2.0f), (float)w);
  fy = fminf(fmaxf(fy, -2.0f), (float)h);
  int ix = (int)fx, iy = (int)fy;
  int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
  int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
  float t00 = img[(size_t)y0 * pitch + x0], t10 = img[(size_t)y0 * pitch + x1];
  float t01 = img[(size_t)y1 * pitch + x0], t11 = img[(size_t)y1 * pitch + x1];
  float d = 0.0f;
  /* a floor() that falls the other way at an integer coordinate changes nothing: weight 0 of one texel = weight 1 of the next */
  if (da <= 256.0f * (ulps * ulp32(x) + extra)) { d += (1.0f / 256.0f) * fabsf((1.0f - rb) * (t10 - t00) + rb * (t11 - t01)); (*nflip)++; }
  if (db <= 256.0f * (ulps * ulp32(y) + extra)) { d += (1.0f / 256.0f) * fabsf((1.0f - ra) * (t01 - t00) + ra * (t11 - t10)); (*nflip)++; }
  return d;
}

int orc_descriptor_bound(const float *img, int w, int h, int pitch, const SiftPoint *p, float ulps, float extra, float *bound, int *wraps)
{
  float gauss[16];
  for (int t = 0; t < 16; t++) gauss[t] = det_exp(-(t - 7.5f) * (t - 7.5f) / 128.0f);
  double raw[128], del[128];
  for (int i = 0; i < 128; i++) { raw[i] = 0.0; del[i] = 0.0; }
  float theta = 2.0f * 3.1415f / 360.0f * p->orientation;
  float sina, cosa;
  det_sincos(theta, &sina, &cosa);
  float scale = 12.0f / 16.0f * p->scale;
  float ssina = scale * sina, scosa = scale * cosa;
  int nflip = 0, nwrap = 0;
  for (int y = 0; y < 16; y++)
    for (int tx = 0; tx < 16; tx++) {
      float xpos = p->xpos + (tx - 7.5f) * scosa - (y - 7.5f) * ssina + 0.5f;
      float ypos = p->ypos + (tx - 7.5f) * ssina + (y - 7.5f) * scosa + 0.5f;
      float dx = tex2d(img, w, h, pitch, xpos + cosa, ypos + sina, 8) - tex2d(img, w, h, pitch, xpos - cosa, ypos - sina, 8);
      float dy = tex2d(img, w, h, pitch, xpos - sina, ypos + cosa, 8) - tex2d(img, w, h, pitch, xpos + sina, ypos - cosa, 8);
      int nf = 0;
      float ddx = fetch_flip(img, w, h, pitch, xpos + cosa, ypos + sina, ulps, extra, &nf) +
                  fetch_flip(img, w, h, pitch, xpos - cosa, ypos - sina, ulps, extra, &nf);
      float ddy = fetch_flip(img, w, h, pitch, xpos - sina, ypos + cosa, ulps, extra, &nf) +
                  fetch_flip(img, w, h, pitch, xpos + sina, ypos - cosa, ulps, extra, &nf);
      nflip += nf;
      float g = sqrtf(dx * dx + dy * dy), gw = gauss[y] * gauss[tx];
      float grad = gw * g;
      float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
      /* The seam (Appendix B #6): 4/3.1415 * pi = 4.000118, so a sample whose gradient angle lies within 9.3e-5 rad below
       * +pi gets angf >= 8, angi = 8, and its vote lands on index hist + 8 — the NEXT cell's bin 0 — instead of this cell's
       * bin 0; one step further (dy < 0) it is bin 0 of this cell again.  Both sides reproduce the quirk, but which side of
       * the 9.3e-5-rad band's edges a sample falls on hangs on dy to a few ulp of the texel values (~6e-5), weight flip or not. */
      const float noise = 6e-5f;
      int seam = dx < 0.0f && g > 0.0f &&
                 (angf >= 8.0f - (4.0f / 3.1415f) * (ddx + ddy + 2.0f * noise) / g - 1e-6f || fabsf(dy) <= ddy + noise);
      if (seam && grad > 0.0f) nwrap++;
      int hori = (tx + 2) / 4 - 1, veri = (y + 2) / 4 - 1;
      float horf = (tx - 1.5f) / 4.0f - hori, verf = (y - 1.5f) / 4.0f - veri;
      int angi = (int)angf;
      if (angi > 7) angi = 7;
      float af = angf - angi;
      /* what this sample can move: its magnitude by gw * (ddx + ddy), its angle by (4/pi) (ddx + ddy) / g bins (at most
       * everything it has); on the seam the whole vote may sit in the other end bin (or in the next cell's bin 0) */
      float dmag = gw * (ddx + ddy);
      float dang = (g > 0.0f) ? (4.0f / 3.1415f) * (ddx + ddy) / g : 0.0f;
      if (dang > 1.0f) dang = 1.0f;
      /* (x 1.5 on the seam: the moved vote also shifts the norm both normalisations divide by — measured worst case 0.92 of
       *  the plain sum over 140 k records, profiles/r05_desc_bound_report.json) */
      float move = dmag + dang * grad + (seam ? 1.5f * grad : 0.0f);
      for (int cy = 0; cy < 2; cy++)
        for (int cx = 0; cx < 2; cx++) {
          int vh = hori + cx, vv = veri + cy;
          if (vh < 0 || vh > 3 || vv < 0 || vv > 3) continue;
          float ws = (cx ? horf : 1.0f - horf) * (cy ? verf : 1.0f - verf);
          int hist = 8 * (4 * vv + vh);
          raw[hist + angi] += ws * (1.0f - af) * grad;
          raw[hist + ((angi + 1) & 7)] += ws * af * grad;
          if (move > 0.0f) {
            /* the two bins the vote is split over and their outer neighbours (the split point may cross a bin edge) */
            for (int k = -1; k <= 2; k++) del[hist + ((angi + k) & 7)] += ws * move;
            if (seam && hist + 8 < 128) del[hist + 8] += 1.5f * ws * grad;
          }
        }
    }
  double n1 = 0.0, dn = 0.0;
  for (int i = 0; i < 128; i++) { n1 += raw[i] * raw[i]; dn += del[i] * del[i]; }
  n1 = sqrt(n1); dn = sqrt(dn);
  if (!(n1 > 0.0)) { for (int i = 0; i < 128; i++) bound[i] = 1.0f; if (wraps) *wraps = nwrap; return nflip; }
  double sum2 = 0.0;
  for (int i = 0; i < 128; i++) { double v = raw[i] / n1; if (v > 0.2) v = 0.2; sum2 += v * v; }
  double rs2 = 1.0 / sqrt(sum2);
  for (int i = 0; i < 128; i++) {
    /* d(v/|v|) <= (d_i + v_i |d| / |v|) / |v|; the clamp is 1-Lipschitz; the second normalisation scales by rs2 and adds
     * the same relative term once more */
    double v = raw[i] / n1;
    double b1 = (del[i] + v * dn) / n1;
    bound[i] = (float)(rs2 * (b1 + (v < 0.2 ? v : 0.2) * rs2 * dn / n1));
  }
  if (wraps) *wraps = nwrap;
  return nflip;
}

/* The same for the records of one ExtractSift call: rebuilds the pyramid of `img` (prefilter + ScaleDowns, as
 * orc_extract does), finds every record's level from its `subsampling`, and writes bound[n][128], flips[n], wraps[n]. */
/* The pyramid levels an ExtractSift call on `img` samples its descriptors from (scaleUp: built from the up-sampled image,
 * cudaSiftH.cu:119-123).  lev[k] are malloc'ed; returns the number of levels. */
static int build_levels(const float *img, int width, int height, int pitch, int numOctaves, float initBlur, int scaleUp,
                        float **lev, int *lw, int *lh, int *lp)
{
  float blur = initBlur > 0.001f ? initBlur : 0.001f;
  const int w = width * (scaleUp ? 2 : 1), h = height * (scaleUp ? 2 : 1);
  lw[0] = w; lh[0] = h; lp[0] = ialign_up(w, 128);
  lev[0] = (float *)malloc(sizeof(float) * (size_t)lh[0] * lp[0]);
  if (scaleUp) {
    float *up = (float *)malloc(sizeof(float) * (size_t)lh[0] * lp[0]);
    orc_scaleup(img, width, height, pitch, up, lp[0]);
    orc_lowpass(up, w, h, lp[0], lev[0], lp[0], blur);
    free(up);
  } else {
    orc_lowpass(img, width, height, pitch, lev[0], lp[0], blur);
  }
  for (int k = 1; k < numOctaves; k++) {
    lw[k] = lw[k - 1] / 2; lh[k] = lh[k - 1] / 2; lp[k] = ialign_up(lw[k], 128);
    lev[k] = (float *)malloc(sizeof(float) * (size_t)(lh[k] > 0 ? lh[k] : 1) * lp[k]);
    if (lw[k] > 0 && lh[k] > 0) orc_scaledown(lev[k - 1], lw[k - 1], lh[k - 1], lp[k - 1], lev[k], lp[k]);
  }
  return numOctaves;
}
/* record i at its octave's scale: coord_scale[i] (optional) undoes RescalePositions first (2 for the records a scaleUp
 * call has rescaled, cudaSiftD.cu:753-761) */
static SiftPoint level_record(const SiftPoint *pts, int i, const float *coord_scale, int numOctaves, int *level)
{
  int k = 0;
  while ((float)(1 << k) < pts[i].subsampling && k < numOctaves - 1) k++;
  SiftPoint q = pts[i];
  const float cs = coord_scale ? coord_scale[i] : 1.0f;
  q.xpos = q.xpos * cs / q.subsampling; q.ypos = q.ypos * cs / q.subsampling; q.scale = q.scale * cs / q.subsampling;
  *level = k;
  return q;
}
void orc_descriptor_bounds2(const float *img, int width, int height, int pitch, int numOctaves, float initBlur, int scaleUp,
                            const SiftPoint *pts, int n, const float *coord_scale, float ulps, const float *dtheta_deg,
                            float *bound, int *flips, int *wraps)
{
  float *lev[16];
  int lw[16], lh[16], lp[16];
  build_levels(img, width, height, pitch, numOctaves, initBlur, scaleUp, lev, lw, lh, lp);
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n; i++) {
    int k;
    SiftPoint q = level_record(pts, i, coord_scale, numOctaves, &k);
    /* the grid reaches 7.5 sqrt(2) sample pitches (0.75 x scale px each) + the +-1 px of the central differences from the keypoint */
    const float radius = 7.5f * 1.41421356f * 0.75f * q.scale + 1.5f;
    const float extra = dtheta_deg ? fabsf(dtheta_deg[i]) * (3.14159265f / 180.0f) * radius : 0.0f;
    flips[i] = orc_descriptor_bound(lev[k], lw[k], lh[k], lp[k], &q, ulps, extra, bound + (size_t)128 * i, wraps ? &wraps[i] : NULL);
  }
  for (int k = 0; k < numOctaves; k++) free(lev[k]);
}
void orc_descriptor_bounds(const float *img, int width, int height, int pitch, int numOctaves, float initBlur,
                           const SiftPoint *pts, int n, float ulps, const float *dtheta_deg, float *bound, int *flips, int *wraps)
{
  orc_descriptor_bounds2(img, width, height, pitch, numOctaves, initBlur, 0, pts, n, NULL, ulps, dtheta_deg, bound, flips, wraps);
}
float orc_descriptor_explain(const float *img, int w, int h, int pitch, const SiftPoint *p, const float *target,
                             float ulps, float extra, float tol, int max_flips, int *nset, int *ncand);
/* targets[n][128], target_geom[n][4]: the other side's descriptors and {xpos, ypos, scale, orientation}; residual[n] = max |difference| the search leaves, nset / ncand: overrides
 * used / candidates per record */
void orc_descriptor_explains(const float *img, int width, int height, int pitch, int numOctaves, float initBlur, int scaleUp,
                             const SiftPoint *pts, int n, const float *coord_scale, const float *targets,
                             const float *target_geom, float ulps, float tol, int max_flips, float *residual, int *nset,
                             int *ncand)
{
  float *lev[16];
  int lw[16], lh[16], lp[16];
  build_levels(img, width, height, pitch, numOctaves, initBlur, scaleUp, lev, lw, lh, lp);
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < n; i++) {
    int k;
    SiftPoint q = level_record(pts, i, coord_scale, numOctaves, &k);
    /* the other side sampled ITS grid: centred on its own position, spaced by its own scale, turned by its own orientation
     * (each may differ from ours in the last bits: libm vs the written-out exp2 / atan2, contraction of the refinement) —
     * no allowance needed, they are known: target_geom[i] = {xpos, ypos, scale, orientation} of the other record */
    if (target_geom) {
      SiftPoint t = pts[i];
      t.xpos = target_geom[4 * i + 0]; t.ypos = target_geom[4 * i + 1]; t.scale = target_geom[4 * i + 2];
      int k2;
      const SiftPoint tq = level_record(&t, 0, coord_scale ? &coord_scale[i] : NULL, numOctaves, &k2);
      q.xpos = tq.xpos; q.ypos = tq.ypos; q.scale = tq.scale;
      q.orientation = target_geom[4 * i + 3];
    }
    residual[i] = orc_descriptor_explain(lev[k], lw[k], lh[k], lp[k], &q, targets + (size_t)128 * i, ulps, 0.0f, tol, max_flips,
                                         nset ? &nset[i] : NULL, ncand ? &ncand[i] : NULL);
  }
  for (int k = 0; k < numOctaves; k++) free(lev[k]);
}
static int ialign_up(int a, int b) { return (a % b != 0) ? (a - a % b + b) : a; }
/* ------------------------------------------------ descriptor tail: an exact EXPLANATION per record (r06, test infrastructure)
 * orc_descriptor_bound() above is a worst case (every candidate fetch flipping the same way): the largest observed
 * difference sits at 0.5-0.6 of it, the median at 0.06, so a regression several times today's differences would pass.
 * This is the tight form of the same statement: the other side's descriptor is REPRODUCED by flipping the rounding of a
 * few of those tie weights (and the seam decision of a few seam samples).  desc_eval() is orc_descriptors() (same
 * expressions, same contraction mode) with two override tables; descriptor_explain() searches greedily for the set of overrides that brings the
 * result within `tol` of the target.  A difference of any other origin leaves a residual and fails. */
/* sample (tx, y) of the 16 x 16 grid, the expression of orc_descriptors() in the current contraction mode */
static inline void sample_pos(const SiftPoint *p, int tx, int y, float ssina, float scosa, float *xpos, float *ypos)
{
  if (g_contract) {
    *xpos = fmaf(-(y - 7.5f), ssina, fmaf(tx - 7.5f, scosa, p->xpos)) + 0.5f;
    *ypos = fmaf(y - 7.5f, scosa, fmaf(tx - 7.5f, ssina, p->ypos)) + 0.5f;
  } else {
    *xpos = p->xpos + (tx - 7.5f) * scosa - (y - 7.5f) * ssina + 0.5f;
    *ypos = p->ypos + (tx - 7.5f) * ssina + (y - 7.5f) * scosa + 0.5f;
  }
}
static inline float tex2d_flip(const float *img, int w, int h, int pitch, float x, float y, int flip)
{
  float xb = x - 0.5f, yb = y - 0.5f;
  float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  float a256 = a * 256.0f, b256 = b * 256.0f;
  float ra = rintf(a256), rb = rintf(b256);
  if (flip & 1) ra = (a256 > ra || (a256 == ra && 0)) ? ra + 1.0f : ra - 1.0f;      /* the other neighbour of the tie */
  if (flip & 2) rb = (b256 > rb) ? rb + 1.0f : rb - 1.0f;
  if (ra < 0.0f) ra = 1.0f;                 /* (a weight cannot leave [0, 1]: the tie at 1/512 has 0 and 1/256 as neighbours) */
  if (rb < 0.0f) rb = 1.0f;
  a = ra * (1.0f / 256.0f); b = rb * (1.0f / 256.0f);
  fx = fminf(fmaxf(fx, -2.0f), (float)w);
  fy = fminf(fmaxf(fy, -2.0f), (float)h);
  int ix = (int)fx, iy = (int)fy;
  int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
  int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
  float t00 = img[(size_t)y0 * pitch + x0], t10 = img[(size_t)y0 * pitch + x1];
  float t01 = img[(size_t)y1 * pitch + x0], t11 = img[(size_t)y1 * pitch + x1];
  float ia = 1.0f - a, ib = 1.0f - b;
  float v = (ia * ib) * t00;
  v = fmaf(a * ib, t10, v);
  v = fmaf(ia * b, t01, v);
  v = fmaf(a * b, t11, v);
  return v;
}
/* flip[256][4]: bit 0 / bit 1 = the x / y weight of fetch k of sample (y, tx) takes the other rounding;
 * seam[256]: 1 = the sample's angle decision falls on the other side of the angi = 8 seam */
static void desc_eval(const float *img, int w, int h, int pitch, const SiftPoint *p, const unsigned char *flip,
                      const unsigned char *seam, const float *gauss, float *out)
{
  float buffer[128];
  for (int i = 0; i < 128; i++) buffer[i] = 0.0f;
  float theta = 2.0f * 3.1415f / 360.0f * p->orientation;
  float sina, cosa;
  det_sincos(theta, &sina, &cosa);
  float scale = 12.0f / 16.0f * p->scale;
  float ssina = scale * sina, scosa = scale * cosa;
  for (int y = 0; y < 16; y++)
    for (int tx = 0; tx < 16; tx++) {
      const unsigned char *f = flip + 4 * (16 * y + tx);
      float xpos, ypos;
      sample_pos(p, tx, y, ssina, scosa, &xpos, &ypos);
      float dx = tex2d_flip(img, w, h, pitch, xpos + cosa, ypos + sina, f[0]) - tex2d_flip(img, w, h, pitch, xpos - cosa, ypos - sina, f[1]);
      float dy = tex2d_flip(img, w, h, pitch, xpos - sina, ypos + cosa, f[2]) - tex2d_flip(img, w, h, pitch, xpos + sina, ypos - cosa, f[3]);
      float grad = gauss[y] * gauss[tx] * sqrtf(mad(dx, dx, dy * dy));
      float angf = mad(4.0f / 3.1415f, fast_atan2(dy, dx), 4.0f);
      int hori = (tx + 2) / 4 - 1, veri = (y + 2) / 4 - 1;
      float horf = (tx - 1.5f) / 4.0f - hori, ihorf = 1.0f - horf;
      float verf = (y - 1.5f) / 4.0f - veri, iverf = 1.0f - verf;
      int angi = (int)angf;
      if (seam[16 * y + tx]) {             /* the other side of the seam: whole vote in bin 0 of the next cell <-> of this cell */
        if (angi >= 8) { angi = 0; angf = 0.0f; }
        else { angi = 8; angf = 8.0f; }
      }
      int angp = (angi < 7 ? angi + 1 : 0);
      angf -= angi;
      float iangf = 1.0f - angf;
      int hist = 8 * (4 * veri + hori);
      int p1 = angi + hist, p2 = angp + hist;
#define VOTE(idx, val) do { int i_ = (idx); if (i_ >= 0 && i_ < 128) buffer[i_] += (val); } while (0)
      if (tx >= 2) {
        float grad1 = ihorf * grad;
        if (y >= 2) { float g2 = iverf * grad1; VOTE(p1, iangf * g2); VOTE(p2, angf * g2); }
        if (y <= 13) { float g2 = verf * grad1; VOTE(p1 + 32, iangf * g2); VOTE(p2 + 32, angf * g2); }
      }
      if (tx <= 13) {
        float grad1 = horf * grad;
        if (y >= 2) { float g2 = iverf * grad1; VOTE(p1 + 8, iangf * g2); VOTE(p2 + 8, angf * g2); }
        if (y <= 13) { float g2 = verf * grad1; VOTE(p1 + 40, iangf * g2); VOTE(p2 + 40, angf * g2); }
      }
#undef VOTE
    }
  float sum = 0.0f;
  for (int i = 0; i < 128; i++) sum += buffer[i] * buffer[i];
  float rs = 1.0f / sqrtf(sum), sum2 = 0.0f;
  for (int i = 0; i < 128; i++) { buffer[i] = fminf(buffer[i] * rs, 0.2f); sum2 += buffer[i] * buffer[i]; }
  float rs2 = 1.0f / sqrtf(sum2);
  for (int i = 0; i < 128; i++) out[i] = buffer[i] * rs2;
}
/* Diagnostics (tools/): the 256 samples of a record's descriptor as the model sees them.
 * out[s] = {xpos, ypos, dx, dy, grad (Gaussian-weighted magnitude), angf, 0, 0}, s = 16 y + tx; p at its octave's scale. */
void orc_descriptor_samples(const float *img, int w, int h, int pitch, const SiftPoint *p, float *out)
{
  float gauss[16];
  for (int t = 0; t < 16; t++) gauss[t] = det_exp(-(t - 7.5f) * (t - 7.5f) / 128.0f);
  float theta = 2.0f * 3.1415f / 360.0f * p->orientation;
  float sina, cosa;
  det_sincos(theta, &sina, &cosa);
  float scale = 12.0f / 16.0f * p->scale;
  float ssina = scale * sina, scosa = scale * cosa;
  for (int y = 0; y < 16; y++)
    for (int tx = 0; tx < 16; tx++) {
      float xpos, ypos;
      sample_pos(p, tx, y, ssina, scosa, &xpos, &ypos);
      float dx = tex2d(img, w, h, pitch, xpos + cosa, ypos + sina, 8) - tex2d(img, w, h, pitch, xpos - cosa, ypos - sina, 8);
      float dy = tex2d(img, w, h, pitch, xpos - sina, ypos + cosa, 8) - tex2d(img, w, h, pitch, xpos + sina, ypos - cosa, 8);
      float *o = out + 8 * (16 * y + tx);
      o[0] = xpos; o[1] = ypos; o[2] = dx; o[3] = dy;
      o[4] = gauss[y] * gauss[tx] * sqrtf(mad(dx, dx, dy * dy));
      o[5] = mad(4.0f / 3.1415f, fast_atan2(dy, dx), 4.0f);
      o[6] = 0.0f; o[7] = 0.0f;
    }
}
static float maxdiff128(const float *a, const float *b)
{
  float m = 0.0f;
  for (int i = 0; i < 128; i++) { float d = fabsf(a[i] - b[i]); if (!(d <= m)) m = d; }
  return m;
}
static double l2diff128(const float *a, const float *b)
{
  double m = 0.0;
  for (int i = 0; i < 128; i++) { double d = (double)a[i] - b[i]; m += d * d; }
  return m;
}
/* Candidates: every fetch weight within `ulps` units in the last place of its coordinate (+ `extra` px) of a rounding tie,
 * every sample whose gradient angle sits on the seam (same criteria as orc_descriptor_bound).  Greedy search: add the
 * override that lowers max |desc - target| most, until it is <= tol, nothing helps, or max_flips are set.
 * Returns the residual; *nset = overrides used, *ncand = candidates there were. */
float orc_descriptor_explain(const float *img, int w, int h, int pitch, const SiftPoint *p, const float *target,
                             float ulps, float extra, float tol, int max_flips, int *nset, int *ncand)
{
  float gauss[16];
  for (int t = 0; t < 16; t++) gauss[t] = det_exp(-(t - 7.5f) * (t - 7.5f) / 128.0f);
  unsigned char flip[1024], seam[256];
  memset(flip, 0, sizeof(flip));
  memset(seam, 0, sizeof(seam));
  /* candidate list: code = sample * 16 + fetch * 4 + axis (axis 0 = x weight, 1 = y weight, 2 = seam of the sample) */
  int *cand = (int *)malloc(sizeof(int) * (2048 + 256));
  int nc = 0;
  float theta = 2.0f * 3.1415f / 360.0f * p->orientation;
  float sina, cosa;
  det_sincos(theta, &sina, &cosa);
  float scale = 12.0f / 16.0f * p->scale;
  float ssina = scale * sina, scosa = scale * cosa;
  for (int y = 0; y < 16; y++)
    for (int tx = 0; tx < 16; tx++) {
      float xpos, ypos;
      sample_pos(p, tx, y, ssina, scosa, &xpos, &ypos);
      const float fxs[4] = {xpos + cosa, xpos - cosa, xpos - sina, xpos + sina};
      const float fys[4] = {ypos + sina, ypos - sina, ypos + cosa, ypos - cosa};
      for (int k = 0; k < 4; k++) {
        float xb = fxs[k] - 0.5f, yb = fys[k] - 0.5f;
        float a256 = (xb - floorf(xb)) * 256.0f, b256 = (yb - floorf(yb)) * 256.0f;
        float da = fabsf(a256 - floorf(a256) - 0.5f), db = fabsf(b256 - floorf(b256) - 0.5f);
        if (da <= 256.0f * (ulps * ulp32(fxs[k]) + extra)) cand[nc++] = (16 * y + tx) * 16 + k * 4 + 0;
        if (db <= 256.0f * (ulps * ulp32(fys[k]) + extra)) cand[nc++] = (16 * y + tx) * 16 + k * 4 + 1;
      }
      float dx = tex2d(img, w, h, pitch, fxs[0], fys[0], 8) - tex2d(img, w, h, pitch, fxs[1], fys[1], 8);
      float dy = tex2d(img, w, h, pitch, fxs[2], fys[2], 8) - tex2d(img, w, h, pitch, fxs[3], fys[3], 8);
      float g = sqrtf(dx * dx + dy * dy);
      float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
      /* on or next to the seam: dx < 0 and |dy| tiny against the gradient (the band is 9.3e-5 rad wide; texel-level noise
       * and a weight flip of this very sample move dy by up to ~1/256 of a texel difference) */
      if (dx < 0.0f && g > 0.0f && (angf >= 8.0f - 2e-3f || fabsf(dy) <= 2e-3f * g + 1e-4f)) cand[nc++] = (16 * y + tx) * 16 + 2;
    }
  float cur[128], tryd[128];
  desc_eval(img, w, h, pitch, p, flip, seam, gauss, cur);
  float err = maxdiff128(cur, target);
  double obj = l2diff128(cur, target);
  int used = 0;
  /* the choice is made on the squared distance (smooth: a flip that repairs one of two wrong elements still counts), the
   * verdict on the largest element; any override may be taken back again (a toggle), so `used` counts toggles */
  while (err > tol && used < max_flips) {
    int best = -1;
    double best_obj = obj;
    float best_err = err;
    for (int c = 0; c < nc; c++) {
      const int code = cand[c];
      const int smp = code >> 4, k = (code >> 2) & 3, axis = code & 3;
      if (axis == 2) seam[smp] ^= 1; else flip[4 * smp + k] ^= (unsigned char)(1 << axis);
      desc_eval(img, w, h, pitch, p, flip, seam, gauss, tryd);
      const double o = l2diff128(tryd, target);
      if (axis == 2) seam[smp] ^= 1; else flip[4 * smp + k] ^= (unsigned char)(1 << axis);
      if (o < best_obj) { best_obj = o; best = c; best_err = maxdiff128(tryd, target); }
    }
    if (best < 0) {
      /* no single toggle helps.  Two toggles whose effects nearly cancel (the "+v" fetch of one sample and the "-v" fetch of
       * its neighbour 0.08 px away sitting on the same tie: the pair leaves a small net difference, either one alone a large
       * one — synthetic frame 187, r06) are invisible to a one-at-a-time search: try pairs. */
      int b1 = -1, b2 = -1;
      if (nc <= 400) {
        for (int c1 = 0; c1 < nc; c1++) {
          const int code1 = cand[c1];
          const int s1 = code1 >> 4, k1 = (code1 >> 2) & 3, a1 = code1 & 3;
          if (a1 == 2) seam[s1] ^= 1; else flip[4 * s1 + k1] ^= (unsigned char)(1 << a1);
          for (int c2 = c1 + 1; c2 < nc; c2++) {
            const int code2 = cand[c2];
            const int s2 = code2 >> 4, k2 = (code2 >> 2) & 3, a2 = code2 & 3;
            if (a2 == 2) seam[s2] ^= 1; else flip[4 * s2 + k2] ^= (unsigned char)(1 << a2);
            desc_eval(img, w, h, pitch, p, flip, seam, gauss, tryd);
            const double o = l2diff128(tryd, target);
            if (a2 == 2) seam[s2] ^= 1; else flip[4 * s2 + k2] ^= (unsigned char)(1 << a2);
            if (o < best_obj) { best_obj = o; b1 = c1; b2 = c2; best_err = maxdiff128(tryd, target); }
          }
          if (a1 == 2) seam[s1] ^= 1; else flip[4 * s1 + k1] ^= (unsigned char)(1 << a1);
        }
      }
      if (b1 < 0) break;
      const int codes[2] = {cand[b1], cand[b2]};
      for (int t = 0; t < 2; t++) {
        const int smp = codes[t] >> 4, k = (codes[t] >> 2) & 3, axis = codes[t] & 3;
        if (axis == 2) seam[smp] ^= 1; else flip[4 * smp + k] ^= (unsigned char)(1 << axis);
      }
      err = best_err;
      obj = best_obj;
      used += 2;
      continue;
    }
    const int code = cand[best];
    const int smp = code >> 4, k = (code >> 2) & 3, axis = code & 3;
    if (axis == 2) seam[smp] ^= 1; else flip[4 * smp + k] ^= (unsigned char)(1 << axis);
    err = best_err;
    obj = best_obj;
    used++;
  }
  free(cand);
  if (nset) *nset = used;
  if (ncand) *ncand = nc;
  return err;
}


/* Scratch layout and sizes of cudaSiftH.cu:39-64 / :80-95 (in floats). */
size_t orc_scratch_floats(int width, int height, int numOctaves, int scaleUp)
{
  const int nd = NUM_SCALES + 3;
  int w = width * (scaleUp ? 2 : 1), h = height * (scaleUp ? 2 : 1);
  int p = ialign_up(w, 128);
  size_t size = (size_t)h * p, sizeTmp = (size_t)nd * h * p;
  for (int i = 0; i < numOctaves; i++) {
    w /= 2; h /= 2;
    int p2 = ialign_up(w, 128);
    size += (size_t)h * p2;
    sizeTmp += (size_t)nd * h * p2;
  }
  return size + sizeTmp;
}

typedef struct {
  const float *taps;
  float thresh, lowestScale;
  int maxPts, fracbits;
  SiftPoint *pts;
  unsigned int cnt[17];
} orc_job_t;

/* ExtractSiftOctave, cudaSiftH.cu:169-232. */
static void extract_octave(orc_job_t *job, const float *img, int w, int h, int pitch, int octave,
                           float subsampling, float *memoryTmp)
{
  unsigned int *cnt = job->cnt;
  orc_laplace(img, w, h, pitch, job->taps, octave, memoryTmp);
  /* counter protocol, cudaSiftD.cu:1297-1300, :1419-1420 */
  unsigned int start = cnt[2 * octave - 1];
  if (cnt[2 * octave] < start) cnt[2 * octave] = start;
  if (cnt[2 * octave + 1] < start) cnt[2 * octave + 1] = start;
  int ndet = orc_findpoints(memoryTmp, w, h, pitch, job->thresh, 10.0f, 1.0f / NUM_SCALES,
                            job->lowestScale / subsampling, subsampling, job->pts, (int)cnt[2 * octave],
                            job->maxPts);
  cnt[2 * octave] += (unsigned int)ndet;
  /* orientation, cudaSiftD.cu:978-979, :1038-1040 */
  int fst = (int)(start < (unsigned int)job->maxPts ? start : (unsigned int)job->maxPts);
  int tot = (int)(cnt[2 * octave] < (unsigned int)job->maxPts ? cnt[2 * octave] : (unsigned int)job->maxPts);
  if (cnt[2 * octave + 1] < cnt[2 * octave]) cnt[2 * octave + 1] = cnt[2 * octave];
  orc_orientations(img, w, h, pitch, job->pts, fst, tot, &cnt[2 * octave + 1], job->maxPts, job->fracbits);
  /* descriptors, cudaSiftD.cu:320-322 */
  int tot2 = (int)(cnt[2 * octave + 1] < (unsigned int)job->maxPts ? cnt[2 * octave + 1] : (unsigned int)job->maxPts);
  orc_descriptors(img, w, h, pitch, job->pts, fst, tot2, subsampling, job->fracbits);
}

/* ExtractSiftLoop, cudaSiftH.cu:146-167: coarsest octave is processed first. */
static void extract_loop(orc_job_t *job, const float *img, int w, int h, int pitch, int numOctaves,
                         float subsampling, float *memoryTmp, float *memorySub)
{
  if (numOctaves > 1) {
    int p = ialign_up(w / 2, 128);
    orc_scaledown(img, w, h, pitch, memorySub, p);
    extract_loop(job, memorySub, w / 2, h / 2, p, numOctaves - 1, subsampling * 2.0f, memoryTmp,
                 memorySub + (size_t)(h / 2) * p);
  }
  extract_octave(job, img, w, h, pitch, numOctaves, subsampling, memoryTmp);
}

/* ExtractSift, cudaSiftH.cu:72-144.  Host-memory image (row stride `pitch`
 * floats).  Returns numPts = min(cnt[2*numOctaves], maxPts) — or
 * min(cnt[2*numOctaves+1], maxPts) when fixNumPts is set (Appendix B #1).
 * counters17 (optional) receives the 17 counters. */
int orc_extract(const float *img, int width, int height, int pitch, int numOctaves, float initBlur,
                float thresh, float lowestScale, int scaleUp, SiftPoint *pts, int maxPts, int fracbits,
                int fixNumPts, unsigned int *counters17)
{
  float taps[8 * 12 * 16];
  orc_laplace_taps(numOctaves, taps);
  const int nd = NUM_SCALES + 3;
  int w = width * (scaleUp ? 2 : 1), h = height * (scaleUp ? 2 : 1);
  int p = ialign_up(w, 128);
  size_t total = orc_scratch_floats(width, height, numOctaves, scaleUp);
  size_t sizeTmp = (size_t)nd * h * p;
  {
    int ww = w, hh = h;
    for (int i = 0; i < numOctaves; i++) {
      ww /= 2; hh /= 2;
      sizeTmp += (size_t)nd * hh * ialign_up(ww, 128);
    }
  }
  float *memoryTmp = (float *)tls_buf(3, sizeof(float) * total);
  float *memorySub = memoryTmp + sizeTmp;
  float *lowImg = memorySub;
  orc_job_t job;
  memset(&job, 0, sizeof(job));
  job.taps = taps; job.thresh = thresh; job.maxPts = maxPts; job.fracbits = fracbits; job.pts = pts;
  float blur = initBlur > 0.001f ? initBlur : 0.001f;
  if (!scaleUp) {
    job.lowestScale = lowestScale;
    orc_lowpass(img, w, h, pitch, lowImg, p, blur);
  } else {
    float *upImg = memoryTmp;      /* cudaSiftH.cu:119-123: the up-sampled image borrows the DoG region */
    orc_scaleup(img, width, height, pitch, upImg, p);
    orc_lowpass(upImg, w, h, p, lowImg, p, blur);
    job.lowestScale = lowestScale * 2.0f;
  }
  extract_loop(&job, lowImg, w, h, p, numOctaves, 1.0f, memoryTmp, memorySub + (size_t)h * p);
  unsigned int c = job.cnt[2 * numOctaves + (fixNumPts ? 1 : 0)];
  int numPts = (int)(c < (unsigned int)maxPts ? c : (unsigned int)maxPts);
  if (scaleUp) {                                   /* RescalePositions, cudaSiftD.cu:753-761 */
    for (int i = 0; i < numPts; i++) {
      pts[i].xpos *= 0.5f; pts[i].ypos *= 0.5f; pts[i].scale *= 0.5f;
    }
  }
  if (counters17) memcpy(counters17, job.cnt, sizeof(job.cnt));
  return numPts;
}

/* Frame-parallel form for the CPU baseline of bench.py: `nframes` tightly packed frames, one per outer OpenMP
 * thread (the stage loops inside then run on `inner_threads` threads each).  pts: nframes*maxPts records,
 * numPts: nframes ints, counters17 (optional): nframes x 17.  Same results as nframes calls of orc_extract. */
#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __GLIBC__
#include <malloc.h>
#endif
void orc_extract_batch(const float *imgs, int nframes, int width, int height, int numOctaves, float initBlur,
                       float thresh, float lowestScale, SiftPoint *pts, int maxPts, int fracbits, int *numPts,
                       unsigned int *counters17, int outer_threads, int inner_threads)
{
#ifdef __GLIBC__
  /* An extraction allocates and frees ~100 work buffers of megabytes each.  glibc serves those with mmap / munmap by
   * default, and with a frame on every core the page faults and the TLB shoot-downs of the unmaps serialise the whole
   * machine (256 cores: 47 frames/s, 0.18 per core, against 3.2 per core for 8 frames on 8 cores).  Keep the blocks in
   * the per-thread heaps instead: after the first frame of a thread nothing is mapped or unmapped any more. */
  static int tuned = 0;
  if (!tuned) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    tuned = 1;
  }
#endif
#ifdef _OPENMP
  const int levels_saved = omp_get_max_active_levels();
  omp_set_max_active_levels(2);
  if (outer_threads < 1) outer_threads = 1;
  if (inner_threads < 1) inner_threads = 1;
  /* static: frame f runs on thread f % outer_threads, so a warm-up batch of outer_threads frames touches every thread's
   * work buffers once */
#pragma omp parallel for schedule(static, 1) num_threads(outer_threads)
#endif
  for (int f = 0; f < nframes; f++) {
#ifdef _OPENMP
    omp_set_num_threads(inner_threads);       /* per-thread ICV: team size of the nested stage loops */
#endif
    numPts[f] = orc_extract(imgs + (size_t)f * width * height, width, height, width, numOctaves, initBlur, thresh,
                            lowestScale, 0, pts + (size_t)f * maxPts, maxPts, fracbits, 0,
                            counters17 ? counters17 + 17 * (size_t)f : NULL);
  }
#ifdef _OPENMP
  omp_set_max_active_levels(levels_saved);
#endif
}

/* ---------------------------------------------------------------- matcher */

/* One correlation, the sequential fp32 FMA chain of matching.cu:343-346
 * (k = 0..127 in order, accumulator starts at 0). */
static inline float dot128(const float *a, const float *b)
{
  float s = 0.0f;
  for (int k = 0; k < 128; k++) s = fmaf(a[k], b[k], s);
  return s;
}
float orc_dot128(const float *a, const float *b) { return dot128(a, b); }

/* MatchSiftData: CleanMatches (matching.cu:289) + FindMaxCorr10 (:301-397).
 * flags bit0: use all n2 columns instead of 32*floor(n2/32) (Appendix B #9);
 * flags bit1: exact second best instead of the 8-class merge (Appendix B #10).
 * Rows [row0, row0+nrows) of set 1 are processed (row-block form). */
void orc_match_rows(SiftPoint *s1, int row0, int nrows, const SiftPoint *s2, int n2, int flags)
{
  const int full = flags & 1, exact = flags & 2;
  const int ncols = full ? n2 : 32 * (n2 / 32);
  /* transposed copy of set 2 so the k-ordered chains of 32 columns vectorise */
  float *bt = (float *)malloc(sizeof(float) * 128 * (size_t)(ncols > 0 ? ncols : 1));
  for (int j = 0; j < ncols; j++)
    for (int k = 0; k < 128; k++) bt[(size_t)k * ncols + j] = s2[j].data[k];
#pragma omp parallel for schedule(dynamic, 8)
  for (int r = row0; r < row0 + nrows; r++) {
    const float *a = s1[r].data;
    float cmax[8], csec[8];
    int cidx[8];
    for (int c = 0; c < 8; c++) { cmax[c] = 0.0f; csec[c] = 0.0f; cidx[c] = -1; }
    float emax = 0.0f, esec = 0.0f;   /* exact top-2 */
    int eidx = -1;
    for (int bp2 = 0; bp2 < ncols; bp2 += 32) {
      int nb = ncols - bp2 < 32 ? ncols - bp2 : 32;
      float acc[32];
      for (int j = 0; j < 32; j++) acc[j] = 0.0f;
      for (int k = 0; k < 128; k++) {
        const float ak = a[k];
        const float *brow = bt + (size_t)k * ncols + bp2;
        for (int j = 0; j < nb; j++) acc[j] = fmaf(ak, brow[j], acc[j]);
      }
      for (int j = 0; j < nb; j++) {                 /* ascending p2 within every class */
        int c = j >> 2;
        float sc = acc[j];
        if (sc > cmax[c]) { csec[c] = cmax[c]; cmax[c] = sc; cidx[c] = bp2 + j; }
        else if (sc > csec[c]) csec[c] = sc;
        if (sc > emax) { esec = emax; emax = sc; eidx = bp2 + j; }
        else if (sc > esec) esec = sc;
      }
    }
    float max_score, sec_score;
    int index;
    if (exact) {
      max_score = emax; sec_score = esec; index = eidx;
    } else {                                         /* matching.cu:375-390 */
      max_score = cmax[0]; sec_score = csec[0]; index = cidx[0];
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
    s1[r].score = max_score;
    s1[r].match = index;
    s1[r].match_xpos = index >= 0 ? s2[index].xpos : 0.0f;   /* never reads sift2[-1] */
    s1[r].match_ypos = index >= 0 ? s2[index].ypos : 0.0f;
    s1[r].ambiguity = sec_score / (max_score + 1e-6f);
  }
  free(bt);
}

void orc_match(SiftPoint *s1, int n1, const SiftPoint *s2, int n2, int flags)
{
  if (n1 <= 0 || n2 <= 0) return;                    /* matching.cu:1095-1096 */
  orc_match_rows(s1, 0, n1, s2, n2, flags);
}

/* Plain argmax matcher over packed [n][128] arrays (the problem statement of
 * match.cu:57-72, MatchC1) — used to cross-check against oracle/_ref. */
void orc_match_argmax(const float *pts1, int n1, const float *pts2, int n2, float *score, int *index)
{
#pragma omp parallel for schedule(static)
  for (int p1 = 0; p1 < n1; p1++) {
    float best = 0.0f;
    int bi = -1;
    for (int p2 = 0; p2 < n2; p2++) {
      float s = dot128(pts1 + (size_t)p1 * 128, pts2 + (size_t)p2 * 128);
      if (s > best) { best = s; bi = p2; }
    }
    score[p1] = best;
    index[p1] = bi;
  }
}

/* ------------------------------------------------------------------ FindHomography
 * RANSAC homography over the matches stored in a SiftPoint array.  Follows
 * FindHomography (matching.cu:1000-1087): valid points = score > minScore &&
 * ambiguity < maxAmbiguity (:1033-1037); numLoops rounded up to 16 (:1013); per loop
 * four distinct valid points drawn with libc rand() in the order of :1041-1053;
 * hypothesis = inverse(A)·b with the 8x8 matrix of ComputeHomographies (:907-948)
 * inverted by InvertMatrix<8> (:821-905: Crout LU, implicit row scaling, one
 * forward/back substitution per unit vector); inliers counted over ALL numPts points
 * with TestHomographies' round-toward-zero products (:971-984); winner = first loop
 * with the largest count (:1063-1068).
 * Contraction: a·b accumulated into sum (":sum -= a*b", "sum += a*b") is one fmaf, as
 * nvcc contracts it; __fmul_rz products are never contracted.
 * Deviation: the reference counts over numPts rounded up to 16, i.e.
 * reads up to 15 uninitialised coordinates; exactly numPts points are tested here. */
static float mul_rz(float a, float b)
{
  float p = a * b;
  float e = fmaf(a, b, -p);                 /* exact residual of the RN product */
  if ((p > 0.0f && e < 0.0f) || (p < 0.0f && e > 0.0f)) {
    uint32_t u;
    memcpy(&u, &p, 4);
    u -= 1u;                                /* one ulp toward zero */
    memcpy(&p, &u, 4);
  }
  return p;
}

static void invert8(float a[8][8], float res[8][8])
{
  int indx[8];
  float vv[8], col[8];
  int imax = 0;
  for (int i = 0; i < 8; i++) {
    float big = 0.0f;
    for (int j = 0; j < 8; j++) {
      float t = fabsf(a[i][j]);
      if (t > big) big = t;
    }
    vv[i] = big > 0.0f ? 1.0f / big : 1e16f;
  }
  for (int j = 0; j < 8; j++) {
    for (int i = 0; i < j; i++) {
      float sum = a[i][j];
      for (int k = 0; k < i; k++) sum = fmaf(-a[i][k], a[k][j], sum);
      a[i][j] = sum;
    }
    float big = 0.0f;
    for (int i = j; i < 8; i++) {
      float sum = a[i][j];
      for (int k = 0; k < j; k++) sum = fmaf(-a[i][k], a[k][j], sum);
      a[i][j] = sum;
      float dum = vv[i] * fabsf(sum);
      if (dum >= big) { big = dum; imax = i; }
    }
    if (j != imax) {
      for (int k = 0; k < 8; k++) { float t = a[imax][k]; a[imax][k] = a[j][k]; a[j][k] = t; }
      vv[imax] = vv[j];
    }
    indx[j] = imax;
    if (a[j][j] == 0.0f) a[j][j] = 1e-16f;
    if (j != 7) {
      float dum = 1.0f / a[j][j];
      for (int i = j + 1; i < 8; i++) a[i][j] *= dum;
    }
  }
  for (int j = 0; j < 8; j++) {
    for (int k = 0; k < 8; k++) col[k] = 0.0f;
    col[j] = 1.0f;
    int ii = -1;
    for (int i = 0; i < 8; i++) {
      int ip = indx[i];
      float sum = col[ip];
      col[ip] = col[i];
      if (ii != -1) {
        for (int k = ii; k < i; k++) sum = fmaf(-a[i][k], col[k], sum);
      } else if (sum != 0.0f) {
        ii = i;
      }
      col[i] = sum;
    }
    for (int i = 7; i >= 0; i--) {
      float sum = col[i];
      for (int k = i + 1; k < 8; k++) sum = fmaf(-a[i][k], col[k], sum);
      col[i] = sum / a[i][i];
    }
    for (int i = 0; i < 8; i++) res[i][j] = col[i];
  }
}

int orc_find_homography(const SiftPoint *pts, int numPts, float *homography, int *numMatches, int numLoops,
                        float minScore, float maxAmbiguity, float thresh)
{
  *numMatches = 0;
  for (int i = 0; i < 9; i++) homography[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  if (!pts || numPts < 8) return -1;
  numLoops = (numLoops + 15) / 16 * 16;
  int *validPts = (int *)malloc(sizeof(int) * (size_t)numPts);
  int numValid = 0;
  for (int i = 0; i < numPts; i++)
    if (pts[i].score > minScore && pts[i].ambiguity < maxAmbiguity) validPts[numValid++] = i;
  int bestLoop = -1;
  if (numValid >= 8) {
    int *sel = (int *)malloc(sizeof(int) * 4 * (size_t)numLoops);
    for (int i = 0; i < numLoops; i++) {
      int p1 = rand() % numValid;
      int p2 = rand() % numValid;
      int p3 = rand() % numValid;
      int p4 = rand() % numValid;
      while (p2 == p1) p2 = rand() % numValid;
      while (p3 == p1 || p3 == p2) p3 = rand() % numValid;
      while (p4 == p1 || p4 == p2 || p4 == p3) p4 = rand() % numValid;
      sel[4 * i + 0] = validPts[p1];
      sel[4 * i + 1] = validPts[p2];
      sel[4 * i + 2] = validPts[p3];
      sel[4 * i + 3] = validPts[p4];
    }
    float *homo = (float *)malloc(sizeof(float) * 8 * (size_t)numLoops);
    int *counts = (int *)malloc(sizeof(int) * (size_t)numLoops);
    const float thresh2 = thresh * thresh;
#pragma omp parallel for schedule(static)
    for (int loop = 0; loop < numLoops; loop++) {
      float a[8][8], ia[8][8], b[8];
      for (int i = 0; i < 4; i++) {
        const SiftPoint *p = &pts[sel[4 * loop + i]];
        float x1 = p->xpos, y1 = p->ypos, x2 = p->match_xpos, y2 = p->match_ypos;
        float *r1 = a[2 * i], *r2 = a[2 * i + 1];
        r1[0] = x1; r1[1] = y1; r1[2] = 1.0f; r1[3] = r1[4] = r1[5] = 0.0f;
        r1[6] = -x2 * x1; r1[7] = -x2 * y1;
        r2[0] = r2[1] = r2[2] = 0.0f; r2[3] = x1; r2[4] = y1; r2[5] = 1.0f;
        r2[6] = -y2 * x1; r2[7] = -y2 * y1;
        b[2 * i] = x2;
        b[2 * i + 1] = y2;
      }
      invert8(a, ia);
      float *h = homo + 8 * (size_t)loop;
      for (int j = 0; j < 8; j++) {
        float sum = 0.0f;
        for (int i = 0; i < 8; i++) sum = fmaf(ia[j][i], b[i], sum);
        h[j] = sum;
      }
      int cnt = 0;
      for (int i = 0; i < numPts; i++) {
        float x1 = pts[i].xpos, y1 = pts[i].ypos, x2 = pts[i].match_xpos, y2 = pts[i].match_ypos;
        float nomx = mul_rz(h[0], x1) + mul_rz(h[1], y1) + h[2];
        float nomy = mul_rz(h[3], x1) + mul_rz(h[4], y1) + h[5];
        float deno = mul_rz(h[6], x1) + mul_rz(h[7], y1) + 1.0f;
        float errx = mul_rz(x2, deno) - nomx;
        float erry = mul_rz(y2, deno) - nomy;
        float err2 = mul_rz(errx, errx) + mul_rz(erry, erry);
        if (err2 < mul_rz(thresh2, mul_rz(deno, deno))) cnt++;
      }
      counts[loop] = cnt;
    }
    int maxCount = -1;
    for (int i = 0; i < numLoops; i++)
      if (counts[i] > maxCount) { maxCount = counts[i]; bestLoop = i; }
    *numMatches = maxCount;
    memcpy(homography, homo + 8 * (size_t)bestLoop, 8 * sizeof(float));
    homography[8] = 1.0f;
    free(counts);
    free(homo);
    free(sel);
  }
  free(validPts);
  return bestLoop;
}

/* ------------------------------------------------------------------ ImproveHomography
 * Iterative least-squares refinement of a homography over the stored matches, geomFuncs.cpp:6-72: numLoops times,
 * accumulate the 8x8 normal equations over the points that pass the score / ambiguity gates AND currently reproject
 * within `thresh` (weight 1, else 0), solve by Cholesky (cv::solve DECOMP_CHOLESKY; a matrix that is not positive
 * definite ZEROES the estimate, as OpenCV's solve does); finally write match_error = sqrt(err) for every point and return the
 * number with err < thresh^2.  Types as in the reference: the estimate A and the sums are double, the point fields
 * float; `den`, `dx`, `dy`, `err` are float variables assigned from double expressions; the products
 * -xpos*match_xpos etc. are float products stored in double (geomFuncs.cpp:38-39, :49-50). */
static int cholesky_solve8(const double *Min, const double *Xin, double *out)
{
  double A[64], B[8];
  memcpy(A, Min, sizeof(A));
  memcpy(B, Xin, sizeof(B));
  for (int i = 0; i < 8; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * 8 + j];
      for (int k = 0; k < j; k++) s -= A[i * 8 + k] * A[j * 8 + k];
      if (i == j) {
        if (!(s > 0)) {                     /* cv::solve zeroes the solution when the factorisation fails */
          memset(out, 0, sizeof(B));
          return 0;
        }
        A[i * 8 + i] = sqrt(s);
      } else {
        A[i * 8 + j] = s / A[j * 8 + j];
      }
    }
  for (int i = 0; i < 8; i++) {
    double s = B[i];
    for (int k = 0; k < i; k++) s -= A[i * 8 + k] * B[k];
    B[i] = s / A[i * 8 + i];
  }
  for (int i = 7; i >= 0; i--) {
    double s = B[i];
    for (int k = i + 1; k < 8; k++) s -= A[k * 8 + i] * B[k];
    B[i] = s / A[i * 8 + i];
  }
  memcpy(out, B, sizeof(B));
  return 1;
}

int orc_improve_homography(SiftPoint *pts, int numPts, float *homography, int numLoops, float minScore,
                           float maxAmbiguity, float thresh)
{
  if (!pts) return 0;
  const float limit = thresh * thresh;
  double A[8];
  for (int i = 0; i < 8; i++) A[i] = homography[i] / homography[8];       /* float division, geomFuncs.cpp:20-21 */
  for (int loop = 0; loop < numLoops; loop++) {
    double M[64], X[8], Y[8];
    memset(M, 0, sizeof(M));
    memset(X, 0, sizeof(X));
    for (int i = 0; i < numPts; i++) {
      const SiftPoint *pt = &pts[i];
      if (pt->score < minScore || pt->ambiguity > maxAmbiguity) continue;
      float den = A[6] * pt->xpos + A[7] * pt->ypos + 1.0f;
      float dx = (A[0] * pt->xpos + A[1] * pt->ypos + A[2]) / den - pt->match_xpos;
      float dy = (A[3] * pt->xpos + A[4] * pt->ypos + A[5]) / den - pt->match_ypos;
      float err = dx * dx + dy * dy;
      float wei = (err < limit ? 1.0f : 0.0f);
      Y[0] = pt->xpos; Y[1] = pt->ypos; Y[2] = 1.0; Y[3] = Y[4] = Y[5] = 0.0;
      Y[6] = -pt->xpos * pt->match_xpos;
      Y[7] = -pt->ypos * pt->match_xpos;
      for (int c = 0; c < 8; c++)
        for (int r = 0; r < 8; r++) M[r * 8 + c] += (Y[c] * Y[r] * wei);
      for (int r = 0; r < 8; r++) X[r] += Y[r] * pt->match_xpos * wei;
      Y[0] = Y[1] = Y[2] = 0.0; Y[3] = pt->xpos; Y[4] = pt->ypos; Y[5] = 1.0;
      Y[6] = -pt->xpos * pt->match_ypos;
      Y[7] = -pt->ypos * pt->match_ypos;
      for (int c = 0; c < 8; c++)
        for (int r = 0; r < 8; r++) M[r * 8 + c] += (Y[c] * Y[r] * wei);
      for (int r = 0; r < 8; r++) X[r] += Y[r] * pt->match_ypos * wei;
    }
    cholesky_solve8(M, X, A);                        /* not positive definite: A = 0 (cv::solve returns false, dst zeroed) */
  }
  int numfit = 0;
  for (int i = 0; i < numPts; i++) {
    SiftPoint *pt = &pts[i];
    float den = A[6] * pt->xpos + A[7] * pt->ypos + 1.0;
    float dx = (A[0] * pt->xpos + A[1] * pt->ypos + A[2]) / den - pt->match_xpos;
    float dy = (A[3] * pt->xpos + A[4] * pt->ypos + A[5]) / den - pt->match_ypos;
    float err = dx * dx + dy * dy;
    if (err < limit) numfit++;
    pt->match_error = sqrtf(err);
  }
  for (int i = 0; i < 8; i++) homography[i] = (float)A[i];
  homography[8] = 1.0f;
  return numfit;
}

// shim_cudasift.cpp — the C++ drop-in layer: cudaSift.h / cudaImage.h on top of the C-ABI.
//
// Plain C++ (built by g++, no HIP headers): every function below is a thin adapter from
// the reference's C++-linkage API (cudaSift.h:35-43, cudaImage.h:8-32) to the extern "C"
// entry points of include/misift.h, so the reference's own mainSift.cpp / geomFuncs.cpp
// link against libcudasift.so + libmisift.so unchanged.
//
// Behaviour kept from the reference: one process-global device context chosen by
// InitCuda (cudaSiftH.cu:19-37); runtime failures print to stderr and exit(-1)
// (cudautils.h:15-39); ExtractSift prints "SIFT extraction time" and "Incl prefiltering
// & memcpy" (cudaSiftH.cu:117, :143) and MatchSiftData prints "MatchSiftData time"
// (matching.cu:1203) unless MISIFT_QUIET=1.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "cudaImage.h"
#include "cudaSift.h"
#include "misift.h"

static misift_ctx *g_ctx = nullptr;

static void die(const char *what)
{
  fprintf(stderr, "misift error in %s: %s\n", what, misift_last_error());
  exit(-1);
}
#define SAFE(call)                       \
  do {                                   \
    if ((call) != MISIFT_OK) die(#call); \
  } while (0)

// Every read-back of this API (ExtractSift's record copy, MatchSiftData's field copy) is queued on the context's own
// stream right behind the kernels, so its calls may return at the last kernel's host flag instead of a full stream
// synchronisation (misift_ctx_set_early_return; the C-ABI itself defaults to the full synchronisation).
static void make_ctx(int dev)
{
  SAFE(misift_ctx_create(dev, nullptr, &g_ctx));
  SAFE(misift_ctx_set_early_return(g_ctx, 1));
  // reference-identical by default (r06): the 32-extrema-per-block cap of FindPointsMultiNew (cudaSiftD.cu:1369-1377) is
  // ON behind this API unless the environment says otherwise — at the fused kernels' speed, see misift.h reference_cap
  if (!getenv("MISIFT_REFERENCE_CAP")) {
    misift_options o;
    SAFE(misift_get_options(g_ctx, &o));
    o.reference_cap = 1;
    SAFE(misift_set_options(g_ctx, &o));
  }
}

static misift_ctx *ctx()
{
  if (!g_ctx) make_ctx(0);
  return g_ctx;
}

// The reference's two build flavours (cudaSift.h:24-33): separate host / device arrays, or — with -DMANAGEDMEM — ONE
// managed pointer valid on both sides (cudaSiftH.cu:239-240).  This file is compiled once per flavour
// (libcudasift.so / libcudasift_managed.so).
#ifdef MANAGEDMEM
static inline SiftPoint *dev_ptr(SiftData &d) { return d.m_data; }
static inline SiftPoint *host_ptr(SiftData &d) { return d.m_data; }
#else
static inline SiftPoint *dev_ptr(SiftData &d) { return d.d_data; }
static inline SiftPoint *host_ptr(SiftData &d) { return d.h_data; }
#endif

static bool quiet()
{
  misift_options o;
  misift_get_options(ctx(), &o);
  return o.quiet != 0;
}

// ------------------------------------------------------------- cudaImage.h
int iDivUp(int a, int b) { return (a % b != 0) ? (a / b + 1) : (a / b); }
int iDivDown(int a, int b) { return a / b; }
int iAlignUp(int a, int b) { return (a % b != 0) ? (a - a % b + b) : a; }
int iAlignDown(int a, int b) { return a - a % b; }

static std::chrono::steady_clock::time_point g_timers[16];
void StartTimer(unsigned int *hTimer)
{
  static unsigned next = 0;
  *hTimer = next++ % 16;
  g_timers[*hTimer] = std::chrono::steady_clock::now();
}
double StopTimer(unsigned int hTimer)
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_timers[hTimer % 16]).count();
}

CudaImage::CudaImage()
    : width(0), height(0), pitch(0), h_data(NULL), d_data(NULL), t_data(NULL), d_internalAlloc(false),
      h_internalAlloc(false)
{
}

CudaImage::~CudaImage()
{
  if (d_internalAlloc && d_data != NULL) misift_free(d_data);
  d_data = NULL;
  if (h_internalAlloc && h_data != NULL) free(h_data);
  h_data = NULL;
  if (t_data != NULL) misift_free(t_data);
  t_data = NULL;
}

void CudaImage::Allocate(int w, int h, int p, bool host, float *devmem, float *hostmem)
{
  width = w;
  height = h;
  pitch = p;
  d_data = devmem;
  h_data = hostmem;
  t_data = NULL;
  d_internalAlloc = false;
  h_internalAlloc = false;
  if (devmem == NULL) {
    ctx();
    SAFE(misift_image_alloc(width, height, &d_data, &pitch));   // pitch overwritten, in floats (cudaImage.cu:24-25)
    if (d_data == NULL) printf("Failed to allocate device data\n");
    d_internalAlloc = true;
  }
  if (host && hostmem == NULL) {
    h_data = (float *)malloc(sizeof(float) * pitch * height);
    h_internalAlloc = true;
  }
}

double CudaImage::Download()
{
  auto t0 = std::chrono::steady_clock::now();
  if (d_data != NULL && h_data != NULL) SAFE(misift_upload_2d(ctx(), d_data, pitch, h_data, width, width, height));
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

double CudaImage::Readback()
{
  auto t0 = std::chrono::steady_clock::now();
  SAFE(misift_download_2d(ctx(), h_data, width, d_data, pitch, width, height));
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Legacy cudaArray path (cudaImage.cu:80-115): no pipeline user; kept working as a plain
// device buffer of pitch x height floats.
double CudaImage::InitTexture()
{
  void *p = nullptr;
  SAFE(misift_malloc(sizeof(float) * (size_t)pitch * height, &p));
  t_data = (float *)p;
  if (t_data == NULL) printf("Failed to allocated texture data\n");
  return 0.0;
}

double CudaImage::CopyToTexture(CudaImage &dst, bool host)
{
  if (dst.t_data == NULL) {
    printf("Error CopyToTexture: No texture data\n");
    return 0.0;
  }
  if ((!host || h_data == NULL) && (host || d_data == NULL)) {
    printf("Error CopyToTexture: No source data\n");
    return 0.0;
  }
  const size_t bytes = sizeof(float) * (size_t)pitch * dst.height;
  if (host) {
    SAFE(misift_copy_h2d(ctx(), dst.t_data, h_data, bytes));
  } else {
    float *tmp = (float *)malloc(bytes);
    SAFE(misift_copy_d2h(ctx(), tmp, d_data, bytes));
    SAFE(misift_copy_h2d(ctx(), dst.t_data, tmp, bytes));
    free(tmp);
  }
  return 0.0;
}

// -------------------------------------------------------------- cudaSift.h
void InitCuda(int devNum)
{
  int nDevices = misift_device_count();
  if (!nDevices) {
    std::cerr << "No HIP devices available" << std::endl;
    return;
  }
  devNum = std::min(nDevices - 1, devNum);
  if (g_ctx) {
    misift_ctx_destroy(g_ctx);
    g_ctx = nullptr;
  }
  make_ctx(devNum);
  char name[256];
  int memClockKHz = 0, busWidth = 0, cus = 0, lds = 0;
  size_t mem = 0;
  SAFE(misift_device_info(devNum, name, sizeof(name), &memClockKHz, &busWidth, &mem, &cus, &lds));
  printf("Device Number: %d\n", devNum);
  printf("  Device name: %s\n", name);
  printf("  Memory Clock Rate (MHz): %d\n", memClockKHz / 1000);
  printf("  Memory Bus Width (bits): %d\n", busWidth);
  // the reference assumes double data rate (cudaSiftH.cu:35-36); HBM3E on gfx950 moves 4 bits per pin and reported
  // memory clock (8 Gb/s at 2 GHz, 8192-bit bus = 8.2 TB/s)
  char arch[64] = "";
  SAFE(misift_device_arch(devNum, arch, sizeof(arch)));
  const double rate = strncmp(arch, "gfx950", 6) == 0 ? 4.0 : 2.0;
  printf("  Peak Memory Bandwidth (GB/s): %.1f\n\n", rate * memClockKHz * (busWidth / 8) / 1.0e6);
}

float *AllocSiftTempMemory(int width, int height, int numOctaves, bool scaleUp)
{
  ctx();
  void *p = nullptr;
  SAFE(misift_malloc(sizeof(float) * misift_scratch_floats(width, height, numOctaves, scaleUp ? 1 : 0), &p));
  return (float *)p;
}

void FreeSiftTempMemory(float *memoryTmp)
{
  if (memoryTmp) SAFE(misift_free(memoryTmp));
}

void ExtractSift(SiftData &siftData, CudaImage &img, int numOctaves, double initBlur, float thresh, float lowestScale,
                 bool scaleUp, float *tempMemory)
{
  auto t0 = std::chrono::steady_clock::now();
  int numPts = 0;
  SAFE(misift_extract(ctx(), img.d_data, img.width, img.height, img.pitch, numOctaves, (float)initBlur, thresh,
                      lowestScale, scaleUp ? 1 : 0, tempMemory, dev_ptr(siftData), siftData.maxPts, &numPts));
  siftData.numPts = numPts;
  const double t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const bool q = quiet();
  if (!q) printf("SIFT extraction time =        %.2f ms %d\n", t1, siftData.numPts);
#ifdef MANAGEDMEM
  SAFE(misift_ctx_sync(ctx()));                        // cudaSiftH.cu:136-137: the host may now read m_data
#else
  if (siftData.h_data && siftData.numPts > 0)
    SAFE(misift_copy_d2h(ctx(), siftData.h_data, siftData.d_data, sizeof(SiftPoint) * (size_t)siftData.numPts));
  else
    // no same-stream read-back behind the kernels: the early return (the last kernel's flag) must not stand in for the
    // reference's blocking contract — a caller reading d_data from its own stream or another device would be unordered
    SAFE(misift_ctx_sync(ctx()));
#endif
  const double t2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!q) printf("Incl prefiltering & memcpy =  %.2f ms %d\n\n", t2, siftData.numPts);
}

void InitSiftData(SiftData &data, int num, bool host, bool dev)
{
  data.numPts = 0;
  data.maxPts = num;
  const size_t sz = sizeof(SiftPoint) * (size_t)num;        // size_t: no int overflow (Appendix B #17)
#ifdef MANAGEDMEM
  (void)host; (void)dev;
  ctx();
  void *p = nullptr;
  SAFE(misift_malloc_managed(sz, &p));                       // cudaSiftH.cu:239-240
  data.m_data = (SiftPoint *)p;
#else
  data.h_data = NULL;
  if (host) data.h_data = (SiftPoint *)malloc(sz);
  data.d_data = NULL;
  if (dev) {
    ctx();
    void *p = nullptr;
    SAFE(misift_malloc(sz, &p));
    data.d_data = (SiftPoint *)p;
  }
#endif
}

void FreeSiftData(SiftData &data)
{
#ifdef MANAGEDMEM
  if (data.m_data != NULL) SAFE(misift_free(data.m_data));   // cudaSiftH.cu:253-254
  data.m_data = NULL;
#else
  if (data.d_data != NULL) SAFE(misift_free(data.d_data));
  data.d_data = NULL;
  if (data.h_data != NULL) free(data.h_data);
  data.h_data = NULL;
#endif
  data.numPts = 0;
  data.maxPts = 0;
}

void PrintSiftData(SiftData &data)
{
#ifdef MANAGEDMEM
  SiftPoint *h_data = data.m_data;                           // cudaSiftH.cu:268-269
#else
  SiftPoint *h_data = data.h_data;
  if (data.h_data == NULL) {
    h_data = (SiftPoint *)malloc(sizeof(SiftPoint) * (size_t)data.maxPts);
    if (data.numPts > 0) SAFE(misift_copy_d2h(ctx(), h_data, data.d_data, sizeof(SiftPoint) * (size_t)data.numPts));
    data.h_data = h_data;
  }
#endif
  for (int i = 0; i < data.numPts; i++) {
    printf("xpos         = %.2f\n", h_data[i].xpos);
    printf("ypos         = %.2f\n", h_data[i].ypos);
    printf("scale        = %.2f\n", h_data[i].scale);
    printf("sharpness    = %.2f\n", h_data[i].sharpness);
    printf("edgeness     = %.2f\n", h_data[i].edgeness);
    printf("orientation  = %.2f\n", h_data[i].orientation);
    printf("score        = %.2f\n", h_data[i].score);
    const float *desc = h_data[i].data;
    for (int j = 0; j < 8; j++) {
      printf(j == 0 ? "data = " : "       ");
      for (int k = 0; k < 16; k++) {
        if (desc[j + 8 * k] < 0.05)
          printf(" .   ");
        else
          printf("%.2f ", desc[j + 8 * k]);
      }
      printf("\n");
    }
  }
  printf("Number of available points: %d\n", data.numPts);
  printf("Number of allocated points: %d\n", data.maxPts);
}

double MatchSiftData(SiftData &data1, SiftData &data2)
{
  auto t0 = std::chrono::steady_clock::now();
  if (!data1.numPts || !data2.numPts) return 0.0;
  if (dev_ptr(data1) == NULL || dev_ptr(data2) == NULL) return 0.0;
  SAFE(misift_match(ctx(), dev_ptr(data1), data1.numPts, dev_ptr(data2), data2.numPts));   // returns synchronised
#ifndef MANAGEDMEM
  if (data1.h_data != NULL)      // score, ambiguity, match, match_xpos, match_ypos (matching.cu:1195-1199)
    SAFE(misift_download_fields(ctx(), data1.h_data, data1.d_data, data1.numPts, offsetof(SiftPoint, score), 5));
  else
    SAFE(misift_ctx_sync(ctx()));                        // (see ExtractSift: nothing same-stream follows the early return)
#else
  SAFE(misift_ctx_sync(ctx()));
#endif
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!quiet()) printf("MatchSiftData time =          %.2f ms\n", ms);
  return ms;
}

double FindHomography(SiftData &data, float *homography, int *numMatches, int numLoops, float minScore,
                      float maxAmbiguity, float thresh)
{
  auto t0 = std::chrono::steady_clock::now();
  SAFE(misift_find_homography(ctx(), dev_ptr(data), data.numPts, homography, numMatches, numLoops, minScore,
                              maxAmbiguity, thresh));
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Extension: the reference's ImproveHomography (geomFuncs.cpp:6-72) is a HOST function over
// SiftData.h_data that the caller compiles itself; this is the same computation on the device-resident records
// (bit-identical result), mirroring match_error back to h_data when there is one.
int ImproveHomographyGPU(SiftData &data, float *homography, int numLoops, float minScore, float maxAmbiguity,
                         float thresh)
{
  int numfit = 0;
  if (dev_ptr(data) == NULL) return 0;
  SAFE(misift_improve_homography(ctx(), dev_ptr(data), data.numPts, homography, numLoops, minScore, maxAmbiguity,
                                 thresh, &numfit));
#ifndef MANAGEDMEM
  if (data.h_data != NULL && data.numPts > 0)
    SAFE(misift_download_fields(ctx(), data.h_data, data.d_data, data.numPts, offsetof(SiftPoint, match_error), 1));
#endif
  return numfit;
}

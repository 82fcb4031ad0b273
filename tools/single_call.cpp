// single_call — the reference demo's inner loop (mainSift.cpp:58-81) on synthetic frames, through the drop-in API
// (include/cudaSift.h -> libcudasift.so): N back-to-back ExtractSift calls on one resident image, then N MatchSiftData
// calls on the two record sets.  Built by `make build/single_call`; tools/single_call.sh runs it under
// rocprofv3 --kernel-trace --hip-trace and tools/single_call_budget.py turns the trace into profiles/*_single_call_*.
//
//   single_call <frame0.f32> <frame1.f32> <width> <height> [calls=200] [octaves=5] [thresh=3.0] [host=1]
//
// host=1 allocates SiftData with a host mirror like mainSift.cpp:60 (ExtractSift then copies the records back, as the
// reference does inside its timed region); host=0 keeps the records on the device (the C-ABI's own cost).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cudaImage.h"
#include "cudaSift.h"

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool load(const char *path, std::vector<float> &v, size_t n)
{
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  v.resize(n);
  const size_t got = fread(v.data(), sizeof(float), n, f);
  fclose(f);
  return got == n;
}

static void stats(const char *what, std::vector<double> t, int npts)
{
  std::sort(t.begin(), t.end());
  const size_t n = t.size();
  printf("{\"what\": \"%s\", \"calls\": %zu, \"p10_ms\": %.4f, \"p50_ms\": %.4f, \"p90_ms\": %.4f, \"min_ms\": %.4f, \"points\": %d}\n",
         what, n, t[n / 10], t[n / 2], t[n * 9 / 10], t[0], npts);
}

int main(int argc, char **argv)
{
  if (argc < 5) {
    fprintf(stderr, "usage: %s frame0.f32 frame1.f32 width height [calls] [octaves] [thresh] [host]\n", argv[0]);
    return 2;
  }
  const int w = atoi(argv[3]), h = atoi(argv[4]);
  const int calls = argc > 5 ? atoi(argv[5]) : 200;
  const int numOctaves = argc > 6 ? atoi(argv[6]) : 5;
  const float thresh = argc > 7 ? (float)atof(argv[7]) : 3.0f;
  const bool host = argc > 8 ? atoi(argv[8]) != 0 : true;
  std::vector<float> f0, f1;
  if (!load(argv[1], f0, (size_t)w * h) || !load(argv[2], f1, (size_t)w * h)) {
    fprintf(stderr, "cannot read %dx%d floats from the frame files\n", w, h);
    return 2;
  }
  setenv("MISIFT_QUIET", "1", 0);          // the per-call printf of the reference would dominate a 0.1 ms call
  InitCuda(0);
  CudaImage img1, img2;
  img1.Allocate(w, h, iAlignUp(w, 128), false, NULL, f0.data());
  img2.Allocate(w, h, iAlignUp(w, 128), false, NULL, f1.data());
  img1.Download();
  img2.Download();
  SiftData siftData1, siftData2;
  InitSiftData(siftData1, 32768, host, true);
  InitSiftData(siftData2, 32768, host, true);
  float *memoryTmp = AllocSiftTempMemory(w, h, numOctaves, false);
  const float initBlur = 1.0f;
  for (int i = 0; i < 20; i++) {           // warm-up: allocations, clocks
    ExtractSift(siftData1, img1, numOctaves, initBlur, thresh, 0.0f, false, memoryTmp);
    ExtractSift(siftData2, img2, numOctaves, initBlur, thresh, 0.0f, false, memoryTmp);
  }
  std::vector<double> te(calls), tm(calls);
  for (int i = 0; i < calls; i++) {
    const double t0 = now_ms();
    ExtractSift(siftData1, img1, numOctaves, initBlur, thresh, 0.0f, false, memoryTmp);
    te[i] = now_ms() - t0;
  }
  ExtractSift(siftData2, img2, numOctaves, initBlur, thresh, 0.0f, false, memoryTmp);
  for (int i = 0; i < 10; i++) MatchSiftData(siftData1, siftData2);
  for (int i = 0; i < calls; i++) {
    const double t0 = now_ms();
    MatchSiftData(siftData1, siftData2);
    tm[i] = now_ms() - t0;
  }
  char what[96];
  snprintf(what, sizeof(what), "ExtractSift %dx%d host=%d", w, h, (int)host);
  stats(what, te, siftData1.numPts);
  snprintf(what, sizeof(what), "MatchSiftData %dx%d host=%d", siftData1.numPts, siftData2.numPts, (int)host);
  stats(what, tm, siftData2.numPts);
  FreeSiftTempMemory(memoryTmp);
  FreeSiftData(siftData1);
  FreeSiftData(siftData2);
  return 0;
}

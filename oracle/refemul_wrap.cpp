// refemul_wrap.cpp — C entry points over the reference's OWN host functions (cudaSiftH.cu, cudaImage.cu,
// matching.cu, compiled against oracle/simt_emul.h by oracle/build_ref.sh) so that tests can drive them
// through ctypes.  TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libcudasift_refemul*.so, never linked by
// the product.  Every function below only marshals arguments; the work is done by reference code:
//   ExtractSift           cudaSiftH.cu:72-144      LowPass      cudaSiftH.cu:406-435
//   ScaleDown             cudaSiftH.cu:308-338     LaplaceMulti cudaSiftH.cu:460-487 (+ PrepareLaplaceKernels :439-458)
//   MatchSiftData         matching.cu:1090-1206    FindHomography matching.cu:1000-1087
//   FindPointsMulti       cudaSiftH.cu:489-514
#include "cudaImage.h"
#include "cudaSift.h"
#include "cudaSiftD.h"
#include "cudaSiftH.h"

extern unsigned int d_PointCounter[8 * 2 + 1];   // cudaSiftD.cu:14
extern int d_MaxNumPoints;                       // cudaSiftD.cu:13
extern float d_LaplaceKernel[8 * 12 * 16];       // cudaSiftD.cu:17

namespace {
struct DevImage {
  CudaImage img;
  DevImage(int w, int h, const float *host = nullptr)
  {
    img.Allocate(w, h, iAlignUp(w, 128), false, NULL, (float *)host);
    if (host) img.Download();
  }
  void read(float *out)
  {
    img.h_data = out;
    img.Readback();
    img.h_data = NULL;
  }
};
}  // namespace

extern "C" {

int refemul_sizeof_siftpoint(void) { return (int)sizeof(SiftPoint); }

// Whole ExtractSift.  `out` has room for maxPts records and receives the device array up to the LAST counter
// (cnt[2*numOctaves+1], clamped): the records past numPts are the finest octave's second orientations.
int refemul_extract(const float *image, int w, int h, int numOctaves, double initBlur, float thresh, float lowestScale,
                    int scaleUp, int maxPts, SiftPoint *out, unsigned int *counters)
{
  SiftData data;
  InitSiftData(data, maxPts, true, true);
  memset(data.d_data, 0, sizeof(SiftPoint) * (size_t)maxPts);
  DevImage src(w, h, image);
  float *tmp = AllocSiftTempMemory(w, h, numOctaves, scaleUp != 0);
  ExtractSift(data, src.img, numOctaves, initBlur, thresh, lowestScale, scaleUp != 0, tmp);
  FreeSiftTempMemory(tmp);
  unsigned int total = d_PointCounter[2 * numOctaves + 1];
  if (total > (unsigned)maxPts) total = (unsigned)maxPts;
  memcpy(out, data.d_data, sizeof(SiftPoint) * (size_t)total);
  memcpy(counters, d_PointCounter, sizeof(unsigned int) * 17);
  int n = data.numPts;
  FreeSiftData(data);
  return n;
}

void refemul_lowpass(const float *image, int w, int h, float sigma, float *out)
{
  DevImage src(w, h, image), dst(w, h);
  LowPass(dst.img, src.img, sigma);
  dst.read(out);
}

void refemul_scaledown(const float *image, int w, int h, float *out)   // out: (w/2) x (h/2)
{
  DevImage src(w, h, image);
  // the kernel writes up to 7 rows past h/2 (cudaSiftD.cu:116-166, no y guard): give it the slack the
  // reference's scratch arena has
  CudaImage dst;
  float *mem = NULL;
  int p = iAlignUp(w / 2, 128);
  cudaMalloc((void **)&mem, sizeof(float) * (size_t)p * (h / 2 + 16));
  dst.Allocate(w / 2, h / 2, p, false, mem);
  ScaleDown(dst, src.img, 0.5f);
  dst.h_data = out;
  dst.Readback();
  dst.h_data = NULL;
  cudaFree(mem);
}

// The 7 DoG planes of one octave.  `octave` is the reference's octave index (numOctaves = finest ... 1).
void refemul_laplace(const float *image, int w, int h, int octave, int numOctaves, float *out)
{
  float kernel[8 * 12 * 16];
  PrepareLaplaceKernels(numOctaves, 0.0f, kernel);
  cudaMemcpyToSymbol(d_LaplaceKernel, kernel, sizeof(kernel));
  DevImage src(w, h, image);
  int p = iAlignUp(w, 128);
  float *mem = NULL;
  cudaMalloc((void **)&mem, sizeof(float) * (size_t)p * h * 8);
  CudaImage planes[8];
  for (int i = 0; i < 7; i++) planes[i].Allocate(w, h, p, false, mem + (size_t)i * p * h);
  LaplaceMulti(0, src.img, planes, octave);
  for (int i = 0; i < 7; i++) {
    planes[i].h_data = out + (size_t)i * w * h;
    planes[i].Readback();
    planes[i].h_data = NULL;
  }
  cudaFree(mem);
}

void refemul_laplace_taps(int numOctaves, float *kernel /* 8*12*16 */) { PrepareLaplaceKernels(numOctaves, 0.0f, kernel); }

// FindPointsMulti (cudaSiftH.cu:489-514 -> FindPointsMultiNew, cudaSiftD.cu:1292-1431) on a caller-supplied stack of 7
// DoG planes [7][h][w]: the detections of ONE octave appended from slot 0.  Returns the detection counter
// d_PointCounter[2*octave] (NOT clamped to maxPts); `out` receives min(count, maxPts) records.
int refemul_findpoints(const float *dog7, int w, int h, float thresh, float edgeLimit, float factor, float lowestScale,
                       float subsampling, int octave, int maxPts, SiftPoint *out)
{
  SiftData data;
  InitSiftData(data, maxPts, false, true);
  memset(data.d_data, 0, sizeof(SiftPoint) * (size_t)maxPts);
  const int p = iAlignUp(w, 128);
  float *mem = NULL;
  cudaMalloc((void **)&mem, sizeof(float) * (size_t)p * h * 8);
  memset(mem, 0, sizeof(float) * (size_t)p * h * 8);
  CudaImage planes[8];
  for (int i = 0; i < 7; i++) {
    planes[i].Allocate(w, h, p, false, mem + (size_t)i * p * h);
    for (int y = 0; y < h; y++) memcpy(planes[i].d_data + (size_t)y * p, dog7 + ((size_t)i * h + y) * w, sizeof(float) * (size_t)w);
  }
  memset(d_PointCounter, 0, sizeof(d_PointCounter));             // cudaSiftH.cu:75-78
  cudaMemcpyToSymbol(d_MaxNumPoints, &maxPts, sizeof(int));
  FindPointsMulti(planes, data, thresh, edgeLimit, factor, lowestScale, subsampling, octave);
  cudaDeviceSynchronize();
  const unsigned int count = d_PointCounter[2 * octave];
  const unsigned int n = count < (unsigned)maxPts ? count : (unsigned)maxPts;
  memcpy(out, data.d_data, sizeof(SiftPoint) * (size_t)n);
  cudaFree(mem);
  FreeSiftData(data);
  return (int)count;
}

// MatchSiftData on two host arrays; the five result fields land in pts1 (matching.cu:1195-1199).
double refemul_match(SiftPoint *pts1, int n1, const SiftPoint *pts2, int n2)
{
  SiftData a, b;
  InitSiftData(a, n1 + 32, false, true);     // the kernel writes rows up to 32*ceil(n1/32) (matching.cu:391-395)
  InitSiftData(b, n2 + 32, false, true);
  memset(a.d_data, 0, sizeof(SiftPoint) * (size_t)(n1 + 32));
  memset(b.d_data, 0, sizeof(SiftPoint) * (size_t)(n2 + 32));
  memcpy(a.d_data, pts1, sizeof(SiftPoint) * (size_t)n1);
  memcpy(b.d_data, pts2, sizeof(SiftPoint) * (size_t)n2);
  a.numPts = n1;
  b.numPts = n2;
  a.h_data = pts1;
  double t = MatchSiftData(a, b);
  a.h_data = NULL;
  FreeSiftData(a);
  FreeSiftData(b);
  return t;
}

double refemul_find_homography(const SiftPoint *pts, int n, float *homography, int *numMatches, int numLoops,
                               float minScore, float maxAmbiguity, float thresh, unsigned seed)
{
  SiftData a;
  InitSiftData(a, n + 32, false, true);
  memcpy(a.d_data, pts, sizeof(SiftPoint) * (size_t)n);
  a.numPts = n;
  srand(seed);
  double t = FindHomography(a, homography, numMatches, numLoops, minScore, maxAmbiguity, thresh);
  FreeSiftData(a);
  return t;
}

}  // extern "C"

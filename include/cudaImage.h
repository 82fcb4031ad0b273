// misift — MI355X-native SIFT behind the CudaSift API.
//
// Drop-in boundary, part 1 of 2: the image container.
//
// This header re-declares, in our own words, the public surface that the
// reference exposes in cudaImage.h:8-32 so that the reference's own
// mainSift.cpp / geomFuncs.cpp compile against it unchanged.  It is plain
// C++ (no HIP, no CUDA, no torch): everything device-side happens behind the
// C-ABI of include/misift.h.
//
// Contract kept identical to the reference (cudaImage.cu:15-78):
//   * public field names, order and types (callers read img.h_data,
//     img.width, img.height directly — mainSift.cpp:160-162);
//   * `pitch` is counted in floats, not bytes;
//   * Allocate(w,h,p,host,devMem,hostMem): a NULL devMem makes the object
//     allocate (and own) pitched device memory and *overwrite* `pitch` with
//     the allocator's value; a NULL hostMem with host==true mallocs a host
//     mirror; caller-supplied pointers are borrowed, never freed;
//   * Download() = host -> device, host rows tightly packed (stride = width);
//     Readback() = device -> host; both return elapsed milliseconds.
// Deliberate fixes: the constructor zero-initialises
// `pitch`, and Allocate() resets the ownership flags before setting them.
#ifndef CUDAIMAGE_H
#define CUDAIMAGE_H

#include <cstddef>

class CudaImage {
public:
  int width, height;      // image size in pixels
  int pitch;              // device row stride in floats
  float *h_data;          // host pixels, row stride == width (may be NULL)
  float *d_data;          // device pixels, row stride == pitch
  float *t_data;          // legacy "texture array" handle (InitTexture only)
  bool d_internalAlloc;   // true when d_data is owned by this object
  bool h_internalAlloc;   // true when h_data is owned by this object

public:
  CudaImage();
  ~CudaImage();

  void Allocate(int width, int height, int pitch, bool withHost,
                float *devMem = NULL, float *hostMem = NULL);
  double Download();                                  // H2D, returns ms
  double Readback();                                  // D2H, returns ms
  double InitTexture();                               // legacy, no pipeline user
  double CopyToTexture(CudaImage &dst, bool host);    // legacy, no pipeline user
};

// Integer helpers (reference: cudaImage.cu:10-13).
int iDivUp(int a, int b);      // ceil(a/b)
int iDivDown(int a, int b);    // floor(a/b)
int iAlignUp(int a, int b);    // round a up to a multiple of b
int iAlignDown(int a, int b);  // round a down to a multiple of b

// Declared by the reference (cudaImage.h:31-32) but defined nowhere in it;
// kept declared for source compatibility, defined as thin wall-clock timers.
void StartTimer(unsigned int *hTimer);
double StopTimer(unsigned int hTimer);

#endif // CUDAIMAGE_H

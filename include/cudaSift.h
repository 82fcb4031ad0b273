// misift — MI355X-native SIFT behind the CudaSift API.
//
// Drop-in boundary, part 2 of 2: feature records and the nine entry points.
//
// Re-declares the public surface of the reference's cudaSift.h:6-43 (same
// names, argument order, defaults, struct layout) so mainSift.cpp and
// geomFuncs.cpp build against this file unchanged.  Plain C++; the device
// work is reached through the C-ABI in include/misift.h.
#ifndef CUDASIFT_H
#define CUDASIFT_H

#include <cstddef>
#include "cudaImage.h"

// One feature: 144 32-bit words = 576 bytes.  The byte offsets are part of
// the ABI (MatchSiftData copies 20 bytes starting at `score` with a 576-byte
// stride; the descriptor sits 16-byte aligned at +64) and are asserted below.
typedef struct {
  float xpos;           //   0  sub-pixel x in input-image pixels
  float ypos;           //   4  sub-pixel y
  float scale;          //   8  sigma-like scale in input-image pixels
  float sharpness;      //  12  interpolated DoG response
  float edgeness;       //  16  tr(H)^2 / det(H) of the DoG Hessian
  float orientation;    //  20  degrees, [0,360)
  float score;          //  24  best correlation     (MatchSiftData)
  float ambiguity;      //  28  second best / best   (MatchSiftData)
  int   match;          //  32  index into the other set, -1 = none
  float match_xpos;     //  36  position of the match in the other image
  float match_ypos;     //  40
  float match_error;    //  44  reprojection error   (ImproveHomography)
  float subsampling;    //  48  2^(octaves below the finest)
  float empty[3];       //  52  padding up to 64
  float data[128];      //  64  4x4x8 descriptor, unit L2 norm
} SiftPoint;

typedef struct {
  int numPts;           // features currently valid
  int maxPts;           // capacity of h_data / d_data
#ifdef MANAGEDMEM
  SiftPoint *m_data;    // managed memory: one pointer for host and device (libcudasift_managed.so)
#else
  SiftPoint *h_data;    // host copy (may be NULL)
  SiftPoint *d_data;    // device copy (HBM)
#endif
} SiftData;

static_assert(sizeof(SiftPoint) == 576, "SiftPoint must be 576 bytes");
static_assert(offsetof(SiftPoint, score) == 24, "score @24");
static_assert(offsetof(SiftPoint, match) == 32, "match @32");
static_assert(offsetof(SiftPoint, subsampling) == 48, "subsampling @48");
static_assert(offsetof(SiftPoint, data) == 64, "descriptor @64");

// Select the GPU and print its memory figures.
void InitCuda(int devNum = 0);

// Scratch arena sized for (width x height, numOctaves); pass it to every
// ExtractSift call to keep allocation out of the loop.
float *AllocSiftTempMemory(int width, int height, int numOctaves, bool scaleUp = false);
void FreeSiftTempMemory(float *memoryTmp);

// Gaussian pyramid -> DoG -> 3x3x3 extrema -> orientation -> descriptors.
// Fills siftData.d_data[0..numPts) (and h_data when present), sets numPts.
void ExtractSift(SiftData &siftData, CudaImage &img, int numOctaves, double initBlur,
                 float thresh, float lowestScale = 0.0f, bool scaleUp = false,
                 float *tempMemory = 0);

void InitSiftData(SiftData &data, int num = 1024, bool host = false, bool dev = true);
void FreeSiftData(SiftData &data);
void PrintSiftData(SiftData &data);

// Brute-force best / second-best correlation of every feature of data1
// against data2; fills score, ambiguity, match, match_xpos, match_ypos of
// data1 (device, mirrored to host when present).  Returns milliseconds.
double MatchSiftData(SiftData &data1, SiftData &data2);

// RANSAC homography over the matches stored in `data`.
double FindHomography(SiftData &data, float *homography, int *numMatches,
                      int numLoops = 1000, float minScore = 0.85f,
                      float maxAmbiguity = 0.95f, float thresh = 5.0f);

// ---- extension (not in the reference header): ImproveHomography of geomFuncs.cpp:6-72 on the device-resident
// records instead of h_data; same arguments, same result.
int ImproveHomographyGPU(SiftData &data, float *homography, int numLoops, float minScore,
                         float maxAmbiguity, float thresh);

#endif // CUDASIFT_H

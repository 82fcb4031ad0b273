// simt_emul.h — a CPU emulation of the CUDA subset the reference (Celebrandil/CudaSift) is written in.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under cudasift_amd/ includes, links or executes this file.
// oracle/build_ref.sh force-includes it (`g++ -include`) into the reference's OWN translation units
// (cudaImage.cu, cudaSiftH.cu [+ the cudaSiftD.cu it #includes], matching.cu), streamed from /root/reference
// through a pipe with one textual rewrite (`Kernel<<<grid, block>>>(args);` is not C++, it becomes
// `simt::Launcher(grid, block).run([&]{ Kernel(args); });`), and links the result into
// oracle/_ref/libcudasift_refemul.so.  That library is the reference's own control flow, constants, tilings,
// shared-memory protocols, warp shuffles, atomics and counter protocol, executed thread by thread on the CPU —
// the thing oracle/sift_oracle.c (a restatement) is pinned against (tests/test_refemul_cpu.py).
//
// Execution model (engine in simt_emul.cpp):
//   * one fiber per CUDA thread of a block; blocks run one after another on an OS thread (OpenMP across
//     blocks; SIMT_THREADS=1 gives the fully sequential schedule: blocks ascending, threads ascending);
//   * __syncthreads() parks a fiber until every live thread of the block has arrived;
//   * __shfl_{up,down,}_sync / __any_sync / __all_sync / __ballot_sync park a fiber until every live lane of its
//     32-lane warp (warp = linear thread id / 32, CUDA's rule) has arrived, then the scheduler resolves the
//     collective with CUDA's semantics (out-of-range source lane -> own value, `width` sub-segments);
//   * atomics are real atomics (blocks may run on different OS threads); __shared__ is per-OS-thread static
//     storage, uninitialised between blocks exactly like the hardware's;
//   * tex2D<float>: unnormalised coordinates, clamp addressing, linear filtering with the weights in CUDA's
//     documented 1.8 fixed-point format (CUDA C Programming Guide, "Texture Fetching / Linear Filtering");
//   * what stays hardware-defined and is mapped to libm / IEEE here: __expf __sinf __cosf exp2f atan2f
//     rsqrtf __fdividef (documented residue: tests allow for it), nvcc's choice of fused multiply-adds
//     (the library is built twice: -ffp-contract=off and =fast).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

using std::abs;   // CUDA's global overloads: abs(float) must not decay to abs(int)
using std::exp;
using std::fabs;
using std::pow;
using std::sqrt;

#define CUDART_VERSION 11000
// a kernel is compiled as its own function with the floating-point options in effect where it is DEFINED: the host code
// that launches it is built without contraction (oracle/build_ref.sh), and inlining a kernel there would drop its FMAs
#define __global__ __attribute__((noinline))
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline
#define __shared__ static thread_local

// ---------------------------------------------------------------------------------------------- vector types
struct uint3 { unsigned int x, y, z; };
struct dim3 {
  unsigned int x, y, z;
  dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static inline float2 make_float2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }

// CUDA's mixed-type global min/max (host and device): the usual arithmetic conversions apply
template <class A, class B> static inline typename std::common_type<A, B>::type min(A a, B b)
{ typedef typename std::common_type<A, B>::type T; return (T)b < (T)a ? (T)b : (T)a; }
template <class A, class B> static inline typename std::common_type<A, B>::type max(A a, B b)
{ typedef typename std::common_type<A, B>::type T; return (T)a < (T)b ? (T)b : (T)a; }

// ---------------------------------------------------------------------------------------------- engine API
namespace simt {
extern thread_local uint3 t_threadIdx, t_blockIdx;
extern thread_local dim3 t_blockDim, t_gridDim;
void launch(dim3 grid, dim3 block, void (*fn)(void *), void *arg, const char *name);
void sync_threads();
enum { SHFL_IDX, SHFL_UP, SHFL_DOWN, VOTE_ANY, VOTE_ALL, VOTE_BALLOT };
uint64_t warp_collective(int kind, uint64_t value, int arg, int width);
struct Launcher {
  dim3 grid, block;
  const char *kernel = "?";
  Launcher(dim3 g, dim3 b, size_t = 0, void * = 0) : grid(g), block(b) {}
  Launcher &name(const char *n) { kernel = n; return *this; }
  template <class F> void run(F &&f)
  {
    typedef typename std::remove_reference<F>::type Fn;
    launch(grid, block, [](void *p) { (*(Fn *)p)(); }, (void *)&f, kernel);
  }
};
struct Texture { const float *ptr; int width, height; size_t pitchBytes; int linear; };
float tex_fetch(const Texture *t, float x, float y);
}  // namespace simt

#define threadIdx (simt::t_threadIdx)
#define blockIdx (simt::t_blockIdx)
#define blockDim (simt::t_blockDim)
#define gridDim (simt::t_gridDim)
#define warpSize 32

static inline void __syncthreads() { simt::sync_threads(); }

template <class T> static inline T simt_shfl(int kind, T var, int arg, int width)
{
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t bits = 0;
  memcpy(&bits, &var, sizeof(T));
  bits = simt::warp_collective(kind, bits, arg, width);
  T out;
  memcpy(&out, &bits, sizeof(T));
  return out;
}
template <class T> static inline T __shfl_sync(unsigned, T var, int lane, int width = 32) { return simt_shfl(simt::SHFL_IDX, var, lane, width); }
template <class T> static inline T __shfl_up_sync(unsigned, T var, unsigned delta, int width = 32) { return simt_shfl(simt::SHFL_UP, var, (int)delta, width); }
template <class T> static inline T __shfl_down_sync(unsigned, T var, unsigned delta, int width = 32) { return simt_shfl(simt::SHFL_DOWN, var, (int)delta, width); }
static inline int __any_sync(unsigned, int pred) { return (int)simt::warp_collective(simt::VOTE_ANY, pred != 0, 0, 32); }
static inline int __all_sync(unsigned, int pred) { return (int)simt::warp_collective(simt::VOTE_ALL, pred != 0, 0, 32); }
static inline unsigned __ballot_sync(unsigned, int pred) { return (unsigned)simt::warp_collective(simt::VOTE_BALLOT, pred != 0, 0, 32); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
// cvt.rzi.s32.f32: NaN -> 0, out of range saturates (x86's cvttss2si gives INT_MIN for both).  build_ref.sh routes
// the one float->int conversion of the reference that sees NaN (descriptor angle bin, cudaSiftD.cu:353) through it.
static inline int simt_f2i_rz(float f) { return f != f ? 0 : f >= 2147483648.0f ? 2147483647 : f <= -2147483648.0f ? (-2147483647 - 1) : (int)f; }

// ---------------------------------------------------------------------------------------------- atomics
static inline int atomicAdd(int *a, int v) { return __atomic_fetch_add(a, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned *a, unsigned v) { return __atomic_fetch_add(a, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float *a, float v)
{
  uint32_t *p = (uint32_t *)a, old = __atomic_load_n(p, __ATOMIC_RELAXED), neu;
  float f;
  do {
    memcpy(&f, &old, 4);
    f += v;
    memcpy(&neu, &f, 4);
  } while (!__atomic_compare_exchange_n(p, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}
static inline unsigned atomicInc(unsigned *a, unsigned lim)   // old >= lim ? 0 : old + 1
{
  unsigned old = __atomic_load_n(a, __ATOMIC_RELAXED);
  while (!__atomic_compare_exchange_n(a, &old, old >= lim ? 0u : old + 1u, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T> static inline T simt_atomic_max(T *a, T v)
{
  T old = __atomic_load_n(a, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(a, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline unsigned atomicMax(unsigned *a, unsigned v) { return simt_atomic_max(a, v); }
static inline int atomicMax(int *a, int v) { return simt_atomic_max(a, v); }
static inline int atomicExch(int *a, int v) { return __atomic_exchange_n(a, v, __ATOMIC_SEQ_CST); }
static inline int atomicCAS(int *a, int cmp, int v)
{
  __atomic_compare_exchange_n(a, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}

// ---------------------------------------------------------------------------------------------- device math
// Hardware-approximate intrinsics -> the accurate libm / IEEE operation (the documented residue).
// (glibc declares __expf/__sinf/__cosf itself: route the CUDA spellings through macros)
static inline float simt_expf(float x) { return expf(x); }
static inline float simt_sinf(float x) { return sinf(x); }
static inline float simt_cosf(float x) { return cosf(x); }
#define __expf(x) simt_expf(x)
#define __sinf(x) simt_sinf(x)
#define __cosf(x) simt_cosf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __fmul_rz(float a, float b)   // product rounded towards zero: exact in double, then truncate
{
  double p = (double)a * (double)b;
  float f = (float)p;
  if (std::isfinite(f) && fabs((double)f) > fabs(p)) f = nextafterf(f, 0.0f);
  return f;
}

// ---------------------------------------------------------------------------------------------- runtime API
typedef enum cudaError { cudaSuccess = 0, cudaErrorMemoryAllocation = 2, cudaErrorInvalidValue = 11 } cudaError_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
typedef void *cudaStream_t;
struct simt_event { std::chrono::steady_clock::time_point t; };
typedef simt_event *cudaEvent_t;
struct cudaDeviceProp { char name[256]; int major, minor, memoryClockRate, memoryBusWidth; };
struct cudaArray;
struct cudaChannelFormatDesc { int x, y, z, w, f; };
template <class T> static inline cudaChannelFormatDesc cudaCreateChannelDesc() { cudaChannelFormatDesc d = {32, 0, 0, 0, 2}; return d; }
typedef unsigned long long cudaTextureObject_t;
enum cudaResourceType { cudaResourceTypeArray, cudaResourceTypeMipmappedArray, cudaResourceTypeLinear, cudaResourceTypePitch2D };
enum cudaTextureAddressMode { cudaAddressModeWrap, cudaAddressModeClamp, cudaAddressModeMirror, cudaAddressModeBorder };
enum cudaTextureFilterMode { cudaFilterModePoint, cudaFilterModeLinear };
enum cudaTextureReadMode { cudaReadModeElementType, cudaReadModeNormalizedFloat };
struct cudaResourceDesc {
  cudaResourceType resType;
  struct {
    struct { cudaArray *array; } array;
    struct { void *devPtr; cudaChannelFormatDesc desc; size_t sizeInBytes; } linear;
    struct { void *devPtr; cudaChannelFormatDesc desc; size_t width, height, pitchInBytes; } pitch2D;
  } res;
};
struct cudaTextureDesc {
  cudaTextureAddressMode addressMode[3];
  cudaTextureFilterMode filterMode;
  cudaTextureReadMode readMode;
  int sRGB;
  float borderColor[4];
  int normalizedCoords;
};
struct cudaResourceViewDesc;

cudaError_t cudaMalloc(void **p, size_t bytes);
cudaError_t cudaMallocManaged(void **p, size_t bytes, unsigned flags = 1);
cudaError_t cudaMallocPitch(void **p, size_t *pitch, size_t widthBytes, size_t height);
cudaError_t cudaFree(void *p);
cudaError_t cudaMemcpy(void *dst, const void *src, size_t bytes, cudaMemcpyKind kind);
cudaError_t cudaMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t widthBytes, size_t height, cudaMemcpyKind kind);
cudaError_t cudaMemset(void *p, int v, size_t bytes);
cudaError_t cudaMallocArray(cudaArray **a, const cudaChannelFormatDesc *d, size_t w, size_t h = 0, unsigned flags = 0);
cudaError_t cudaFreeArray(cudaArray *a);
cudaError_t cudaMemcpyToArray(cudaArray *dst, size_t wOff, size_t hOff, const void *src, size_t bytes, cudaMemcpyKind kind);
cudaError_t cudaDeviceSynchronize();
cudaError_t cudaGetLastError();
const char *cudaGetErrorString(cudaError_t e);
cudaError_t cudaGetDeviceCount(int *n);
cudaError_t cudaSetDevice(int d);
cudaError_t cudaGetDeviceProperties(cudaDeviceProp *p, int dev);
cudaError_t cudaEventCreate(cudaEvent_t *e);
cudaError_t cudaEventDestroy(cudaEvent_t e);
cudaError_t cudaEventRecord(cudaEvent_t e, cudaStream_t s = 0);
cudaError_t cudaEventSynchronize(cudaEvent_t e);
cudaError_t cudaEventElapsedTime(float *ms, cudaEvent_t a, cudaEvent_t b);
cudaError_t cudaCreateTextureObject(cudaTextureObject_t *t, const cudaResourceDesc *r, const cudaTextureDesc *d, const cudaResourceViewDesc *v);
cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t);

// __constant__ / __device__ symbols are plain globals here
template <class T> static inline cudaError_t cudaGetSymbolAddress(void **p, T &sym) { *p = (void *)&sym; return cudaSuccess; }
template <class T> static inline cudaError_t cudaMemcpyToSymbol(T &sym, const void *src, size_t bytes, size_t off = 0, cudaMemcpyKind = cudaMemcpyHostToDevice)
{ memcpy((char *)&sym + off, src, bytes); return cudaSuccess; }
template <class T> static inline cudaError_t cudaMemcpyToSymbolAsync(T &sym, const void *src, size_t bytes, size_t off = 0, cudaMemcpyKind = cudaMemcpyHostToDevice, cudaStream_t = 0)
{ memcpy((char *)&sym + off, src, bytes); return cudaSuccess; }

template <class T> static inline T tex2D(cudaTextureObject_t t, float x, float y)
{
  static_assert(std::is_same<T, float>::value, "only float textures are emulated");
  return simt::tex_fetch((const simt::Texture *)(uintptr_t)t, x, y);
}

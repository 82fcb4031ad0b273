// simt_emul.cpp — the engine behind simt_emul.h: fibers, the block scheduler, warp collectives, the fake
// runtime API and the texture unit.  TEST INFRASTRUCTURE ONLY (see the header).  x86-64 System V only
// (the context switch is 14 instructions of assembly; both this container and the GPU boxes are x86-64).
#include "simt_emul.h"

#include <omp.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <vector>

#if !defined(__x86_64__)
#error "simt_emul: the fiber switch is written for x86-64"
#endif

// void simt_switch(void **save_sp, void *load_sp): park the callee-saved state of the running context on its own
// stack, remember the stack pointer, continue on the other stack.
extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size simt_switch,.-simt_switch
)");

namespace simt {

thread_local uint3 t_threadIdx = {0, 0, 0}, t_blockIdx = {0, 0, 0};
thread_local dim3 t_blockDim, t_gridDim;

namespace {

enum State { RUNNABLE, AT_BARRIER, AT_COLLECTIVE, DONE };
constexpr size_t STACK_BYTES = 128 * 1024, GUARD_BYTES = 4096;

struct Fiber {
  void *sp;
  State state;
  uint3 tid;
  int kind, arg, width;
  uint64_t value, result;
};

struct Worker {
  std::vector<Fiber> fibers;
  char *pool = nullptr;
  size_t pool_fibers = 0;
  void *sched_sp = nullptr;
  Fiber *cur = nullptr;
  void (*fn)(void *) = nullptr;
  void *arg = nullptr;
  ~Worker() { if (pool) munmap(pool, pool_fibers * (STACK_BYTES + GUARD_BYTES)); }
  void reserve(size_t n)
  {
    if (n <= pool_fibers) return;
    if (pool) munmap(pool, pool_fibers * (STACK_BYTES + GUARD_BYTES));
    size_t bytes = n * (STACK_BYTES + GUARD_BYTES);
    pool = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (pool == MAP_FAILED) { perror("simt_emul: mmap of fiber stacks"); abort(); }
    for (size_t i = 0; i < n; i++) mprotect(pool + i * (STACK_BYTES + GUARD_BYTES), GUARD_BYTES, PROT_NONE);
    pool_fibers = n;
    fibers.resize(n);
  }
  char *stack_top(size_t i) { return pool + (i + 1) * (STACK_BYTES + GUARD_BYTES); }
};

thread_local Worker tl_worker;

extern "C" void simt_fiber_main()
{
  Worker &w = tl_worker;
  w.fn(w.arg);
  Worker &w2 = tl_worker;
  w2.cur->state = DONE;
  simt_switch(&w2.cur->sp, w2.sched_sp);
  __builtin_unreachable();
}

inline void park(State s)
{
  Worker &w = tl_worker;
  if (!w.cur) { fprintf(stderr, "simt_emul: block-level primitive called outside a kernel\n"); abort(); }
  w.cur->state = s;
  simt_switch(&w.cur->sp, w.sched_sp);
}

void resolve_warp(Fiber *lane0, int nl)
{
  int kind = -1;
  unsigned ballot = 0;
  bool all = true;
  for (int l = 0; l < nl; l++) {
    Fiber &f = lane0[l];
    if (f.state != AT_COLLECTIVE) continue;
    if (kind < 0) kind = f.kind;
    else if (kind != f.kind) { fprintf(stderr, "simt_emul: lanes of one warp wait in different collectives\n"); abort(); }
    if (f.value & 1) ballot |= 1u << l; else all = false;
  }
  for (int l = 0; l < nl; l++) {
    Fiber &f = lane0[l];
    if (f.state != AT_COLLECTIVE) continue;
    const int wd = f.width, seg = l / wd * wd, rel = l - seg;
    int src = l;
    switch (kind) {
      case SHFL_IDX:  src = seg + (int)((unsigned)f.arg % (unsigned)wd); break;
      case SHFL_UP:   src = rel - f.arg >= 0 ? l - f.arg : l; break;
      case SHFL_DOWN: src = rel + f.arg < wd ? l + f.arg : l; break;
      default: break;
    }
    if (kind <= SHFL_DOWN) {
      if (src >= nl || lane0[src].state != AT_COLLECTIVE) src = l;   // inactive source lane: undefined on hardware
      f.result = lane0[src].value;
    } else if (kind == VOTE_ANY) f.result = ballot != 0;
    else if (kind == VOTE_ALL) f.result = all;
    else f.result = ballot;
  }
  for (int l = 0; l < nl; l++)
    if (lane0[l].state == AT_COLLECTIVE) lane0[l].state = RUNNABLE;
}

void run_block(Worker &w, dim3 bd)
{
  const int n = (int)(bd.x * bd.y * bd.z);
  w.reserve((size_t)n);
  for (int t = 0; t < n; t++) {
    Fiber &f = w.fibers[t];
    f.tid.x = t % bd.x;
    f.tid.y = t / bd.x % bd.y;
    f.tid.z = t / (bd.x * bd.y);
    f.state = RUNNABLE;
    void **sp = (void **)w.stack_top(t);
    *--sp = nullptr;                          // fake return address: the entry never returns
    *--sp = (void *)&simt_fiber_main;         // `ret` of the first switch lands here, rsp % 16 == 8 as after a call
    for (int r = 0; r < 6; r++) *--sp = nullptr;
    f.sp = sp;
  }
  int live = n;
  while (live > 0) {
    bool progress = false;
    // Warps advance round-robin, one scheduling quantum (= up to the next barrier / warp collective / exit) each per
    // sweep, like the warp schedulers of an SM keep the warps of a block within a few instructions of each other.
    // (Running one warp ahead to its next barrier is also a legal schedule, but it exposes benign races of the
    // reference that the hardware never loses: ExtractSiftDescriptorsCONSTNew re-uses sums[] for the second
    // normalisation with no barrier after the first read, cudaSiftD.cu:396-403.)
    for (int w0 = 0; w0 < n; w0 += 32) {
      const int nl = std::min(32, n - w0);
      int waiting = 0, alive = 0;
      for (int l = 0; l < nl; l++) {
        Fiber &f = w.fibers[w0 + l];
        if (f.state == RUNNABLE) {
          t_threadIdx = f.tid;
          w.cur = &f;
          simt_switch(&w.sched_sp, f.sp);
          w.cur = nullptr;
          progress = true;
          if (f.state == DONE) live--;
        }
        if (f.state != DONE) alive++;
        if (f.state == AT_COLLECTIVE) waiting++;
      }
      if (waiting == 0) continue;
      if (waiting != alive) {
        fprintf(stderr, "simt_emul: warp %d: %d lanes in a warp collective while %d live lanes wait elsewhere\n",
                w0 / 32, waiting, alive - waiting);
        abort();
      }
      resolve_warp(&w.fibers[w0], nl);
      progress = true;
    }
    if (live == 0) break;
    int at_bar = 0;
    for (int t = 0; t < n; t++) at_bar += w.fibers[t].state == AT_BARRIER;
    if (at_bar == live) {
      for (int t = 0; t < n; t++)
        if (w.fibers[t].state == AT_BARRIER) w.fibers[t].state = RUNNABLE;
    } else if (!progress) {
      fprintf(stderr, "simt_emul: deadlock (%d live threads, %d at the barrier)\n", live, at_bar);
      abort();
    }
  }
}

int env_threads()
{
  static int n = -1;
  if (n < 0) {
    const char *e = getenv("SIMT_THREADS");
    n = e ? atoi(e) : std::min(omp_get_max_threads(), 32);
    if (n < 1) n = 1;
  }
  return n;
}

}  // namespace

void sync_threads() { park(AT_BARRIER); }

uint64_t warp_collective(int kind, uint64_t value, int arg, int width)
{
  Worker &w = tl_worker;
  if (!w.cur) { fprintf(stderr, "simt_emul: warp collective outside a kernel\n"); abort(); }
  Fiber *f = w.cur;
  f->kind = kind;
  f->value = value;
  f->arg = arg;
  f->width = width > 0 && width <= 32 ? width : 32;
  park(AT_COLLECTIVE);
  return f->result;
}

namespace {
const char *volatile g_kernel = "?";
void on_fault(int sig, siginfo_t *si, void *)
{
  // best effort: say where the emulated kernel died, then die the default way
  char buf[256];
  int n = snprintf(buf, sizeof(buf), "simt_emul: signal %d (address %p) in kernel %s, block (%u,%u,%u), thread (%u,%u,%u)\n", sig, si->si_addr, g_kernel,
                   t_blockIdx.x, t_blockIdx.y, t_blockIdx.z, t_threadIdx.x, t_threadIdx.y, t_threadIdx.z);
  if (n > 0) (void)!write(2, buf, (size_t)n);
  void *frames[32];
  backtrace_symbols_fd(frames, backtrace(frames, 32), 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
}  // namespace

void launch(dim3 grid, dim3 block, void (*fn)(void *), void *arg, const char *name)
{
  static bool hooked = false;
  if (!hooked && getenv("SIMT_FAULT_REPORT")) {
    hooked = true;
    static char altstack[65536];               // the faulting fiber's stack may be the problem
    stack_t ss = {altstack, 0, sizeof(altstack)};
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  }
  g_kernel = name;
  const long nblocks = (long)grid.x * grid.y * grid.z;
  if (nblocks <= 0 || block.x * block.y * block.z == 0) return;
  const int nthreads = (int)std::min<long>(env_threads(), nblocks);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
  for (long b = 0; b < nblocks; b++) {
    Worker &w = tl_worker;
    w.fn = fn;
    w.arg = arg;
    t_gridDim = grid;
    t_blockDim = block;
    t_blockIdx.x = (unsigned)(b % grid.x);
    t_blockIdx.y = (unsigned)(b / grid.x % grid.y);
    t_blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
    run_block(w, block);
  }
}

// --------------------------------------------------------------------------------------------- texture unit
// CUDA C Programming Guide, "Texture Fetching": unnormalised coordinates, xB = x - 0.5, i = floor(xB), alpha =
// frac(xB) "stored in 9-bit fixed point format with 8 bits of fractional value"; clamp addressing;
// tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1].
// How the fraction is rounded into 1.8 is not documented: nearest (ties to even) by default, SIMT_TEX_ROUND=trunc
// truncates, SIMT_TEX_ROUND=exact keeps the full fp32 fraction.
float tex_fetch(const Texture *t, float x, float y)
{
  static int mode = -1;
  if (mode < 0) {
    const char *e = getenv("SIMT_TEX_ROUND");
    mode = !e ? 0 : !strcmp(e, "trunc") ? 1 : !strcmp(e, "exact") ? 2 : 0;
  }
  const int w = t->width, h = t->height;
  const size_t pitch = t->pitchBytes / sizeof(float);
  if (!t->linear) {
    int ix = (int)floorf(fminf(fmaxf(x, -1.0f), (float)w)), iy = (int)floorf(fminf(fmaxf(y, -1.0f), (float)h));
    ix = std::min(std::max(ix, 0), w - 1);
    iy = std::min(std::max(iy, 0), h - 1);
    return t->ptr[(size_t)iy * pitch + ix];
  }
  const float xb = x - 0.5f, yb = y - 0.5f;
  float fi = floorf(xb), fj = floorf(yb);
  float a = xb - fi, b = yb - fj;
  if (mode == 0) {
    a = nearbyintf(a * 256.0f) / 256.0f;
    b = nearbyintf(b * 256.0f) / 256.0f;
  } else if (mode == 1) {
    a = floorf(a * 256.0f) / 256.0f;
    b = floorf(b * 256.0f) / 256.0f;
  }
  fi = fminf(fmaxf(fi, -2.0f), (float)w);
  fj = fminf(fmaxf(fj, -2.0f), (float)h);
  const int i = (int)fi, j = (int)fj;
  const int i0 = std::min(std::max(i, 0), w - 1), i1 = std::min(std::max(i + 1, 0), w - 1);
  const int j0 = std::min(std::max(j, 0), h - 1), j1 = std::min(std::max(j + 1, 0), h - 1);
  const float t00 = t->ptr[(size_t)j0 * pitch + i0], t10 = t->ptr[(size_t)j0 * pitch + i1];
  const float t01 = t->ptr[(size_t)j1 * pitch + i0], t11 = t->ptr[(size_t)j1 * pitch + i1];
  // the filter arithmetic of the hardware is not specified beyond the formula: evaluate it in double (exact
  // products of an 8-bit weight pair with an fp32 texel, one rounding at the end)
  const double da = a, db = b;
  const double v = (1.0 - da) * (1.0 - db) * t00 + da * (1.0 - db) * t10 + (1.0 - da) * db * t01 + da * db * t11;
  return (float)v;
}

}  // namespace simt

// ------------------------------------------------------------------------------------------------ runtime API
// Every allocation gets a zeroed 4 KiB front pad: the reference reads sift2[-1] when a row has no match
// (matching.cu:393-394) and the emulation must survive what the hardware survives.
// The payload is filled with 0xFF bytes (a NaN as float, -1 as int): the reference also reads memory it never wrote —
// FindHomography hands TestHomographies numPtsUp = 16 * ceil(numPts / 16) points of which only numPts were copied
// (matching.cu:1021-1064) — and with whatever the heap held there the emulated reference was not a function of its
// inputs (a freed d_coord of an earlier call put real coordinates behind the 8 points of a later one: 6 inliers
// instead of 5, one CPU run in ten).  NaN coordinates never pass the inlier test, i.e. the padding counts for nothing,
// which is what the oracle and homography.hip implement (deliberate deviations).
namespace {
constexpr size_t PAD = 4096;
cudaError_t last_error = cudaSuccess;
}

cudaError_t cudaMalloc(void **p, size_t bytes)
{
  const size_t total = (bytes + 2 * PAD + 4095) / 4096 * 4096;
  char *raw = (char *)aligned_alloc(4096, total);
  if (!raw) { *p = nullptr; return last_error = cudaErrorMemoryAllocation; }
  memset(raw, 0, PAD);
  memset(raw + PAD, 0xFF, total - PAD);
  *p = raw + PAD;
  return cudaSuccess;
}
cudaError_t cudaMallocManaged(void **p, size_t bytes, unsigned) { return cudaMalloc(p, bytes); }
cudaError_t cudaMallocPitch(void **p, size_t *pitch, size_t widthBytes, size_t height)
{
  *pitch = (widthBytes + 511) / 512 * 512;      // the CUDA allocator's 512-byte row alignment (= 128 floats)
  return cudaMalloc(p, *pitch * height);
}
cudaError_t cudaFree(void *p) { if (p) free((char *)p - PAD); return cudaSuccess; }
cudaError_t cudaMemcpy(void *dst, const void *src, size_t bytes, cudaMemcpyKind) { memmove(dst, src, bytes); return cudaSuccess; }
cudaError_t cudaMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t widthBytes, size_t height, cudaMemcpyKind)
{
  for (size_t r = 0; r < height; r++) memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, widthBytes);
  return cudaSuccess;
}
cudaError_t cudaMemset(void *p, int v, size_t bytes) { memset(p, v, bytes); return cudaSuccess; }
cudaError_t cudaMallocArray(cudaArray **a, const cudaChannelFormatDesc *, size_t w, size_t h, unsigned)
{ return cudaMalloc((void **)a, w * std::max<size_t>(h, 1) * sizeof(float)); }
cudaError_t cudaFreeArray(cudaArray *a) { return cudaFree(a); }
cudaError_t cudaMemcpyToArray(cudaArray *dst, size_t, size_t, const void *src, size_t bytes, cudaMemcpyKind)
{ memmove(dst, src, bytes); return cudaSuccess; }
cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
cudaError_t cudaGetLastError() { cudaError_t e = last_error; last_error = cudaSuccess; return e; }
const char *cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : e == cudaErrorMemoryAllocation ? "out of memory" : "error"; }
cudaError_t cudaGetDeviceCount(int *n) { *n = 1; return cudaSuccess; }
cudaError_t cudaSetDevice(int) { return cudaSuccess; }
cudaError_t cudaGetDeviceProperties(cudaDeviceProp *p, int)
{
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "CPU SIMT emulation (oracle/simt_emul)");
  p->major = 7; p->minor = 5; p->memoryClockRate = 1000000; p->memoryBusWidth = 64;
  return cudaSuccess;
}
cudaError_t cudaEventCreate(cudaEvent_t *e) { *e = new simt_event(); return cudaSuccess; }
cudaError_t cudaEventDestroy(cudaEvent_t e) { delete e; return cudaSuccess; }
cudaError_t cudaEventRecord(cudaEvent_t e, cudaStream_t) { e->t = std::chrono::steady_clock::now(); return cudaSuccess; }
cudaError_t cudaEventSynchronize(cudaEvent_t) { return cudaSuccess; }
cudaError_t cudaEventElapsedTime(float *ms, cudaEvent_t a, cudaEvent_t b)
{ *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return cudaSuccess; }
cudaError_t cudaCreateTextureObject(cudaTextureObject_t *t, const cudaResourceDesc *r, const cudaTextureDesc *d, const cudaResourceViewDesc *)
{
  if (r->resType != cudaResourceTypePitch2D || d->normalizedCoords || d->addressMode[0] != cudaAddressModeClamp ||
      d->addressMode[1] != cudaAddressModeClamp) {
    fprintf(stderr, "simt_emul: only unnormalised clamped pitch2D float textures are emulated\n");
    return last_error = cudaErrorInvalidValue;
  }
  simt::Texture *x = new simt::Texture();
  x->ptr = (const float *)r->res.pitch2D.devPtr;
  x->width = (int)r->res.pitch2D.width;
  x->height = (int)r->res.pitch2D.height;
  x->pitchBytes = r->res.pitch2D.pitchInBytes;
  x->linear = d->filterMode == cudaFilterModeLinear;
  *t = (cudaTextureObject_t)(uintptr_t)x;
  return cudaSuccess;
}
cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t) { delete (simt::Texture *)(uintptr_t)t; return cudaSuccess; }

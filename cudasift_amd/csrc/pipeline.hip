// pipeline.hip — host-fed extraction pipeline (SURVEY §8f "next" row 2).
//
// Replaces, for streams of frames, the blocking upload CudaImage::Download (reference
// cudaImage.cu:55-66: one synchronous cudaMemcpy2D per image) and the blocking SiftPoint
// read-back at the end of ExtractSift (cudaSiftH.cu:139-140).  The reference's README notes
// that the upload alone costs more than the extraction (README.md:49); on an MI355X the
// extraction of a 1080p frame takes ~40 us while its fp32 upload takes ~130 us of PCIe, so
// the feed decides the delivered rate:
//   * frames may be uploaded as 8-bit (4x fewer PCIe bytes; the first kernel converts in
//     registers, bit-identical to an fp32 upload of the same values),
//   * HIP streams for upload, compute, counts and record read-back with per-slot events, so the
//     upload of batch k+1 and the read-back of batch k-1 overlap the extraction of batch k (the
//     record copy has its own stream: queued behind the counts of LATER batches it would wait
//     for their extraction — measured 9.4 k instead of 16 k frames/s),
//   * the valid records of all frames of a batch are packed contiguously on the device
//     (pack kernels below) so the read-back is ONE sized copy instead of one per frame,
//   * nothing is silently dropped: a batch in which a candidate list overflowed is redone with the
//     exact dense kernels when it is collected.
#include <string.h>
#include <vector>
#include "common.hpp"

// ------------------------------------------------------------------ counts / offsets / packing
// staged_noct > 0: the reference counters are not published yet (descr_all does that) — derive numPts from the
// staged per-octave detection / duplicate counts exactly as descr_all will (cudaSiftD.cu:1297-1300 protocol;
// `slot` odd = fix_numpts, the finest octave's duplicates are counted too).
__global__ __launch_bounds__(1024) void export_counts_kernel(const unsigned *__restrict__ counters, int nframes, int slot,
                                                              int max_pts, int *__restrict__ counts,
                                                              int *__restrict__ offsets, int staged_noct)
{
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int f0 = 0; f0 < nframes; f0 += 1024) {
    const int f = f0 + tid;
    int c = 0;
    if (f < nframes) {
      const unsigned *cnt = counters + (size_t)f * CNT_STRIDE;
      unsigned n = cnt[slot];
      if (staged_noct > 0) {
        n = 0;
        for (int k = 1; k <= staged_noct; k++) {
          n += cnt[CNT_DET + k];
          if (k < staged_noct || (slot & 1)) n += cnt[CNT_DUP + k];
        }
      }
      c = (int)(n < (unsigned)max_pts ? n : (unsigned)max_pts);      // cudaSiftH.cu:116
      counts[f] = cnt[CNT_CANDOVF] ? -1 : c;
      if (cnt[CNT_CANDOVF]) c = 0;
    }
    if (!offsets) continue;
    // inclusive scan inside the wavefront, then across the 16 wavefronts
    int incl = c;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = carry_s;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    if (f < nframes) offsets[f] = base + incl - c;
    __syncthreads();
    if (tid == 1023) carry_s = base + incl;
    __syncthreads();
  }
  if (offsets && tid == 0) offsets[nframes] = carry_s;
}

// One wavefront moves one 576-byte record as 36 x 16 bytes; blockIdx.y = frame.
__global__ __launch_bounds__(256) void pack_records_kernel(const SiftPointD *__restrict__ pts, int max_pts,
                                                            const int *__restrict__ offsets,
                                                            SiftPointD *__restrict__ packed)
{
  const int f = blockIdx.y;
  const int begin = offsets[f], n = offsets[f + 1] - begin;
  const int lane = threadIdx.x & 63;
  const float4 *src = reinterpret_cast<const float4 *>(pts + (size_t)f * max_pts);
  float4 *dst = reinterpret_cast<float4 *>(packed + begin);
  const long long total = (long long)n * 36;                         // 16-byte words of this frame
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    dst[i] = src[i];
  (void)lane;
}

int launch_export_counts(misift_ctx *ctx, int nframes, int num_octaves, int max_pts, int *counts_out, int *offsets_out)
{
  const int slot = 2 * num_octaves + (ctx->opt.fix_numpts ? 1 : 0);
  LaunchScope ls(ctx, "export_counts");
  hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_counters, nframes, slot, max_pts,
                     counts_out, offsets_out, 0);
  return ls.finish();
}

int launch_export_counts_staged(misift_ctx *ctx, int nframes, int num_octaves, int max_pts, int *counts_out,
                                int *offsets_out)
{
  const int slot = 2 * num_octaves + (ctx->opt.fix_numpts ? 1 : 0);
  LaunchScope ls(ctx, "export_counts");
  hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_counters, nframes, slot, max_pts,
                     counts_out, offsets_out, num_octaves);
  return ls.finish();
}

int launch_pack_records(misift_ctx *ctx, const SiftPointD *pts, int max_pts, int nframes, const int *offsets,
                        SiftPointD *packed)
{
  LaunchScope ls(ctx, "pack_records");
  hipLaunchKernelGGL(pack_records_kernel, dim3(32, nframes), dim3(256), 0, ctx->stream, pts, max_pts, offsets, packed);
  return ls.finish();
}

// ------------------------------------------------------------------ the pipe
struct PipeSlot {
  void *d_frames;          // batch of source frames (u8 or fp32), tightly packed rows
  SiftPointD *d_packed;    // valid records of the batch, contiguous
  int *d_counts;           // [batch] counts then [batch + 1] offsets
  int *h_counts;           // pinned mirror
  hipEvent_t ev_uploaded, ev_done, ev_counts;
  int nframes;             // frames submitted in this slot (0 = free)
};

struct misift_pipe {
  misift_ctx *ctx;
  int width, height, batch, src_u8, num_octaves, max_pts, depth;
  float init_blur, thresh, lowest_scale;
  size_t frame_elems;
  hipStream_t s_up, s_compute, s_down, s_rec;
  hipStream_t saved_stream;
  float *d_scratch;
  SiftPointD *d_pts;       // unpacked records of the batch in flight on the compute stream
  std::vector<PipeSlot> slots;
  long long submitted, collected;
};

#define PIPE_TRY(expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      misift_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      misift_pipe_destroy(p);                                                                \
      return MISIFT_EHIP;                                                                    \
    }                                                                                        \
  } while (0)

extern "C" void misift_pipe_destroy(misift_pipe *p)
{
  if (!p) return;
  hipSetDevice(p->ctx->device);
  if (p->s_compute) hipStreamSynchronize(p->s_compute);
  if (p->s_up) hipStreamSynchronize(p->s_up);
  if (p->s_down) hipStreamSynchronize(p->s_down);
  if (p->s_rec) hipStreamSynchronize(p->s_rec);
  if (p->ctx->stream == p->s_compute) p->ctx->stream = p->saved_stream;
  for (PipeSlot &s : p->slots) {
    if (s.d_frames) misift_dev_free(s.d_frames);
    if (s.d_packed) misift_dev_free(s.d_packed);
    if (s.d_counts) misift_dev_free(s.d_counts);
    if (s.h_counts) hipHostFree(s.h_counts);
    if (s.ev_uploaded) hipEventDestroy(s.ev_uploaded);
    if (s.ev_done) hipEventDestroy(s.ev_done);
    if (s.ev_counts) hipEventDestroy(s.ev_counts);
  }
  if (p->d_scratch) misift_dev_free(p->d_scratch);
  if (p->d_pts) misift_dev_free(p->d_pts);
  if (p->s_up) hipStreamDestroy(p->s_up);
  if (p->s_compute) hipStreamDestroy(p->s_compute);
  if (p->s_down) hipStreamDestroy(p->s_down);
  if (p->s_rec) hipStreamDestroy(p->s_rec);
  delete p;
}

extern "C" int misift_pipe_create(misift_ctx *ctx, int width, int height, int batch_frames, int src_u8,
                                  int num_octaves, float init_blur, float thresh, float lowest_scale, int max_pts,
                                  int depth, misift_pipe **out)
{
  if (!ctx || !out || width < 1 || height < 1 || batch_frames < 1 || max_pts < 1 || depth < 1 || depth > 8 ||
      num_octaves < 1 || num_octaves > MISIFT_MAX_OCTAVES) {
    misift_set_error("misift_pipe_create: invalid argument");
    return MISIFT_EINVAL;
  }
  // same size rule as misift_extract (r06: any size from 1 x 1; tiny frames / deep pyramids run on the dense kernels)
  if (width >= 16384 || height >= 16384) {
    misift_set_error("misift_pipe_create: %dx%d is too large (candidate codes hold 14-bit coordinates)", width, height);
    return MISIFT_EINVAL;
  }
  *out = nullptr;
  misift_pipe *p = new misift_pipe();
  p->ctx = ctx;
  p->width = width; p->height = height; p->batch = batch_frames; p->src_u8 = src_u8 ? 1 : 0;
  p->num_octaves = num_octaves; p->max_pts = max_pts; p->depth = depth;
  p->init_blur = init_blur; p->thresh = thresh; p->lowest_scale = lowest_scale;
  p->frame_elems = (size_t)width * height;
  p->s_up = p->s_compute = p->s_down = p->s_rec = nullptr;
  misift_warn_hw_queues("misift_pipe_create");      // the upload / read-back streams want hardware queues of their own
  p->saved_stream = ctx->stream;
  p->d_scratch = nullptr; p->d_pts = nullptr;
  p->submitted = p->collected = 0;
  PIPE_TRY(hipSetDevice(ctx->device));
  PIPE_TRY(hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking));
  PIPE_TRY(hipStreamCreateWithFlags(&p->s_compute, hipStreamNonBlocking));
  PIPE_TRY(hipStreamCreateWithFlags(&p->s_down, hipStreamNonBlocking));
  PIPE_TRY(hipStreamCreateWithFlags(&p->s_rec, hipStreamNonBlocking));
  const size_t S = misift_scratch_floats(width, height, num_octaves, 0);
  PIPE_TRY(misift_dev_alloc((void **)&p->d_scratch, sizeof(float) * S * batch_frames, "pipe_scratch"));
  PIPE_TRY(misift_dev_alloc((void **)&p->d_pts, sizeof(SiftPointD) * (size_t)max_pts * batch_frames, "pipe_points"));
  // ExtractSift does not write a record's match fields (score ... match_error: MatchSiftData's), and the records collected
  // from this staging array carry them along: start them out as zeros instead of whatever the allocation held (r06: found
  // by running the suite under MISIFT_GUARD=1, where fresh memory is 0xFF)
  PIPE_TRY(hipMemset(p->d_pts, 0, sizeof(SiftPointD) * (size_t)max_pts * batch_frames));
  p->slots.resize(depth);
  for (PipeSlot &s : p->slots) memset(&s, 0, sizeof(s));
  const size_t elem = src_u8 ? 1 : sizeof(float);
  for (PipeSlot &s : p->slots) {
    PIPE_TRY(misift_dev_alloc(&s.d_frames, elem * p->frame_elems * batch_frames, "pipe_frames"));
    PIPE_TRY(misift_dev_alloc((void **)&s.d_packed, sizeof(SiftPointD) * (size_t)max_pts * batch_frames, "pipe_packed"));
    PIPE_TRY(misift_dev_alloc((void **)&s.d_counts, sizeof(int) * (2 * (size_t)batch_frames + 1), "pipe_counts"));
    PIPE_TRY(hipHostMalloc((void **)&s.h_counts, sizeof(int) * (2 * (size_t)batch_frames + 1), hipHostMallocDefault));
    PIPE_TRY(hipEventCreateWithFlags(&s.ev_uploaded, hipEventDisableTiming));
    PIPE_TRY(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    PIPE_TRY(hipEventCreateWithFlags(&s.ev_counts, hipEventDisableTiming));
  }
  *out = p;
  return MISIFT_OK;
}

extern "C" int misift_pipe_pending(const misift_pipe *p) { return p ? (int)(p->submitted - p->collected) : 0; }

extern "C" int misift_pipe_submit(misift_pipe *p, const void *host_frames, int nframes)
{
  if (!p || !host_frames || nframes < 1 || nframes > p->batch) {
    misift_set_error("misift_pipe_submit: invalid argument");
    return MISIFT_EINVAL;
  }
  if (p->submitted - p->collected >= p->depth) {
    misift_set_error("misift_pipe_submit: all %d slots are in flight — collect a batch first", p->depth);
    return MISIFT_EINVAL;
  }
  misift_ctx *ctx = p->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  PipeSlot &s = p->slots[p->submitted % p->depth];
  const size_t elem = p->src_u8 ? 1 : sizeof(float);
  // 1. upload (pinned host memory makes this truly asynchronous)
  HIP_TRY(hipMemcpyAsync(s.d_frames, host_frames, elem * p->frame_elems * nframes, hipMemcpyHostToDevice, p->s_up));
  HIP_TRY(hipEventRecord(s.ev_uploaded, p->s_up));
  // 2. extraction + count export + packing on the compute stream
  HIP_TRY(hipStreamWaitEvent(p->s_compute, s.ev_uploaded, 0));
  hipStream_t saved = ctx->stream;
  const int saved_split = ctx->split_tail;
  ctx->stream = p->s_compute;
  ctx->split_tail = 0;          // the pipe already overlaps batches on its own streams; the split measured -6 % here
  const int fused_saved = ctx->opt.fused;
  // tiny frames / deep pyramids run on the dense kernels (reference_cap: counted on the fused path, a frame that needs the
  // cap applied comes back as overflowed and is redone by misift_pipe_collect like any overflowed frame)
  if (misift_tiny_call(p->width, p->height, p->num_octaves)) ctx->opt.fused = 0;
  int rc = misift_extract_enqueue(ctx, s.d_frames, p->src_u8, nframes, (long long)p->frame_elems, p->width, p->height,
                                  p->width, p->num_octaves, p->init_blur, p->thresh, p->lowest_scale, 0, p->d_scratch,
                                  p->d_pts, p->max_pts);
  if (!rc) rc = launch_export_counts(ctx, nframes, p->num_octaves, p->max_pts, s.d_counts, s.d_counts + p->batch);
  if (!rc) rc = launch_pack_records(ctx, p->d_pts, p->max_pts, nframes, s.d_counts + p->batch, s.d_packed);
  ctx->stream = saved;
  ctx->split_tail = saved_split;
  ctx->opt.fused = fused_saved;
  if (rc) return rc;
  HIP_TRY(hipEventRecord(s.ev_done, p->s_compute));
  // 3. counts + offsets to the host on the read-back stream
  HIP_TRY(hipStreamWaitEvent(p->s_down, s.ev_done, 0));
  HIP_TRY(hipMemcpyAsync(s.h_counts, s.d_counts, sizeof(int) * (2 * (size_t)p->batch + 1), hipMemcpyDeviceToHost,
                         p->s_down));
  HIP_TRY(hipEventRecord(s.ev_counts, p->s_down));
  s.nframes = nframes;
  p->submitted++;
  return MISIFT_OK;
}

extern "C" int misift_pipe_collect(misift_pipe *p, int *nframes_out, int *counts_out, void *host_records,
                                   size_t capacity_records, size_t *nrecords_out)
{
  if (!p || !nframes_out || !counts_out || !nrecords_out) {
    misift_set_error("misift_pipe_collect: invalid argument");
    return MISIFT_EINVAL;
  }
  if (p->submitted == p->collected) {
    misift_set_error("misift_pipe_collect: nothing in flight");
    return MISIFT_EINVAL;
  }
  HIP_TRY(hipSetDevice(p->ctx->device));
  PipeSlot &s = p->slots[p->collected % p->depth];
  HIP_TRY(hipEventSynchronize(s.ev_counts));
  const int n = s.nframes;
  *nframes_out = n;
  *nrecords_out = 0;
  bool overflow = false;
  for (int f = 0; f < n; f++)
    if (s.h_counts[f] < 0) overflow = true;
  int rc = MISIFT_OK;
  if (overflow) {
    // a candidate list of the fused scan overflowed (extreme contrast / tiny thresh): redo THIS batch from its
    // frames (still in the slot) with the synchronous call, which falls back to the exact dense kernels; the
    // batches queued behind it have already packed their results into their own slots
    misift_ctx *ctx = p->ctx;
    HIP_TRY(hipStreamSynchronize(p->s_compute));
    hipStream_t saved = ctx->stream;
    ctx->stream = p->s_compute;
    std::vector<int> tmp((size_t)n);
    // the fused scan is known to overflow on this batch: start at the dense kernels (no wasted fused attempt, and
    // no graph capture/replay of a call that cannot succeed — the graph path only takes fused calls)
    const int fused_saved = ctx->opt.fused;
    ctx->opt.fused = 0;
    rc = misift_extract_sync(ctx, s.d_frames, p->src_u8, n, (long long)p->frame_elems, p->width, p->height, p->width,
                             p->num_octaves, p->init_blur, p->thresh, p->lowest_scale, 0, p->d_scratch, p->d_pts,
                             p->max_pts, tmp.data());
    ctx->opt.fused = fused_saved;
    if (!rc) rc = launch_export_counts(ctx, n, p->num_octaves, p->max_pts, s.d_counts, s.d_counts + p->batch);
    if (!rc) rc = launch_pack_records(ctx, p->d_pts, p->max_pts, n, s.d_counts + p->batch, s.d_packed);
    ctx->stream = saved;
    if (rc) { s.nframes = 0; p->collected++; return rc; }
    HIP_TRY(hipMemcpyAsync(s.h_counts, s.d_counts, sizeof(int) * (2 * (size_t)p->batch + 1), hipMemcpyDeviceToHost,
                           p->s_compute));
    HIP_TRY(hipStreamSynchronize(p->s_compute));
  }
  const int *offs2 = s.h_counts + p->batch;
  const size_t total2 = (size_t)offs2[n];
  *nrecords_out = total2;
  for (int f = 0; f < n; f++) counts_out[f] = s.h_counts[f];
  if (host_records && total2) {
    if (total2 > capacity_records) {
      misift_set_error("misift_pipe_collect: %zu records but room for %zu", total2, capacity_records);
      rc = MISIFT_ENOMEM;
    } else {
      // ev_counts implies the packing of this slot is complete; the copy must not queue behind later batches
      HIP_TRY(hipMemcpyAsync(host_records, s.d_packed, sizeof(SiftPointD) * total2, hipMemcpyDeviceToHost, p->s_rec));
      HIP_TRY(hipStreamSynchronize(p->s_rec));
    }
  }
  s.nframes = 0;
  p->collected++;
  return rc;
}

extern "C" int misift_host_alloc(size_t bytes, void **out)
{
  if (!out) {
    misift_set_error("misift_host_alloc: invalid argument");
    return MISIFT_EINVAL;
  }
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return MISIFT_OK;
}

extern "C" int misift_host_free(void *ptr)
{
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return MISIFT_OK;
}

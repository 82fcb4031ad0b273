/* misift.h — the thin C-ABI between host code (C++ shim, ctypes, cgo, JNI …)
 * and the gfx950 HIP kernels of libmisift.so.
 *
 * Plain pointers and sizes only: no C++ types, no torch types, no HIP types
 * (a hipStream_t crosses as void*).  Every function returns 0 on success or a
 * negative MISIFT_E* code; misift_last_error() gives the message for the
 * calling thread.  Device pointers are ordinary HBM pointers (hipMalloc,
 * torch tensors' data_ptr(), …).
 *
 * Each entry point names the reference interface it replaces (file:line in
 * Celebrandil/CudaSift).  The C++ drop-in layer (include/cudaSift.h,
 * include/cudaImage.h, cudasift_amd/csrc/shim_cudasift.cpp) is written
 * purely on top of this file.
 */
#ifndef MISIFT_H
#define MISIFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISIFT_OK          0
#define MISIFT_EINVAL     -1   /* bad argument                                  */
#define MISIFT_EHIP       -2   /* HIP runtime / kernel launch failure           */
#define MISIFT_ENOMEM     -3   /* allocation failure                            */
#define MISIFT_ENODEV     -4   /* no gfx950 device visible                      */

#define MISIFT_NUM_SCALES      5   /* DoG scales searched per octave (cudaSiftD.h:8) */
#define MISIFT_MAX_OCTAVES     7   /* counter protocol has 17 slots = 2*8+1          */
#define MISIFT_POINT_BYTES   576   /* sizeof(SiftPoint) (cudaSift.h:6-22)            */

/* Same 576-byte record as SiftPoint in include/cudaSift.h (reference
 * cudaSift.h:6-22); declared here so C callers need no C++ header. */
typedef struct misift_point {
  float xpos, ypos, scale, sharpness, edgeness, orientation, score, ambiguity;
  int32_t match;
  float match_xpos, match_ypos, match_error, subsampling;
  float empty[3];
  float data[128];
} misift_point;

typedef struct misift_ctx misift_ctx;   /* opaque: one per device (+stream)  */

/* Behaviour switches.  Defaults reproduce the reference. */
typedef struct misift_options {
  int texfrac_bits;      /* 8 = emulate CUDA's 9-bit texture filter weights
                            (cudaSiftH.cu:196-205); 23 = full fp32 bilinear   */
  int fix_numpts;        /* 0 = numPts excludes finest-octave duplicates
                            (cudaSiftH.cu:115); 1 = include them              */
  int match_full;        /* 0 = ignore the last n2%32 columns
                            (matching.cu:325); 1 = use every column           */
  int match_exact_top2;  /* 0 = 8-class lossy runner-up merge
                            (matching.cu:378-390); 1 = true second best       */
  int quiet;             /* 1 = C++ shim prints nothing per call              */
  int fused;             /* 1 = fused DoG+extrema kernel (no DoG planes in
                            HBM); 0 = separate laplace / findpoints kernels   */
  int deterministic;     /* 0 = records in atomic-append order within an octave
                            segment, like the reference (cudaSiftD.cu:1420,
                            :1043: run-to-run the SET is equal, the order is
                            not); 1 = order fixed by the keypoints themselves
                            (tile, y, x, scale): repeated runs are byte-identical
                            (also MISIFT_DETERMINISTIC=1).
                            The fused path orders by (tile, y, x, scale); the
                            dense kernels (fused = 0, or the automatic exact
                            re-run after a candidate-list overflow) sort every
                            segment by (y, x, scale, orientation) afterwards   */
  int reference_cap;     /* 0 = every scale-space extremum goes on to the refinement
                            (ours); 1 = the reference's cap: a block of
                            FindPointsMultiNew — 30 columns x 8 rows of one scale —
                            keeps its first 32 extrema (by column, then row) and
                            drops the rest (cudaSiftD.cu:1369-1377; SURVEY
                            Appendix B #4).  Natural images never reach 32 per
                            240 pixels.  Costs nothing since r06: the fused path
                            counts the true extrema of every block and only a
                            frame in which one reaches a 33rd is redone on the
                            dense per-level kernels, which apply the cap in the
                            reference's order.  The cudaSift.h shim switches it ON
                            (reference-identical by default); the C-ABI default is
                            0 (also MISIFT_REFERENCE_CAP=1)                      */
} misift_options;

/* ------------------------------------------------------------------ runtime */

/* cudaSiftH.cu:19-37 (InitCuda): device count / properties. */
int misift_device_count(void);
int misift_device_info(int device, char *name, int name_len, int *mem_clock_khz,
                       int *bus_width_bits, size_t *total_mem_bytes,
                       int *num_cus, int *lds_bytes_per_block);
/* ISA name of the device ("gfx950:sramecc+:xnack-"); the shim uses it to print the right HBM data rate. */
int misift_device_arch(int device, char *arch, int arch_len);

/* hipGraph replay of repeated synchronous calls: when misift_extract / misift_extract_batch is called again
 * with exactly the same arguments and buffers (the reference demo does, mainSift.cpp:64-69), the launch
 * sequence is captured on the 2nd occurrence and replayed afterwards as one hipGraphLaunch.  OFF by default:
 * on ROCm 7.2 / MI355X the replay measured SLOWER than the ~10 direct launches (single 1080p frame 0.199 ms vs
 * 0.170 ms; 64-frame batch 1.508 vs 1.502 ms).  Also MISIFT_GRAPH=1 in the environment at context creation. */
int misift_ctx_set_graph_replay(misift_ctx *ctx, int on);

/* One context per device; `stream` is a hipStream_t (NULL = the null stream).
 * The context owns the per-frame point counters (cudaSiftD.cu:13-14), a
 * pinned read-back buffer and the filter tap tables (cudaSiftD.cu:15-17). */
int misift_ctx_create(int device, void *stream, misift_ctx **out);
void misift_ctx_destroy(misift_ctx *ctx);
int misift_ctx_set_stream(misift_ctx *ctx, void *stream);
/* Synchronous calls return at the last kernel's completion flag (1) instead of after a stream synchronisation (0, the
 * default): see misift_extract.  The host polls a word of pinned memory, i.e. it spins a core while it waits. */
int misift_ctx_set_early_return(misift_ctx *ctx, int on);
/* Diagnostics: calls this context re-ran with a stand-alone ScaleDown chain launch because the bounded in-launch wait
 * of the single-call path expired (never on a healthy device; MISIFT_CHAIN_WAIT_US sets the bound, default 100000). */
int misift_ctx_chain_fallbacks(misift_ctx *ctx);
/* Diagnostics: calls this context re-ran with separate orientation and descriptor launches because the bounded in-launch
 * wait of the fused kernel of the single-call path (r06) expired (never on a healthy device; MISIFT_FUSE_WAIT_US sets the
 * bound, default 100000; MISIFT_FUSE_ORIENT=0 keeps the two launches). */
int misift_ctx_fuse_fallbacks(misift_ctx *ctx);
/* Diagnostics: 1 if the last extraction enqueued on this context dealt the workgroups of its per-keypoint kernels out
 * in proportion to the frames' keypoint counts (batches of more than MISIFT_SMALL_FRAMES frames, MISIFT_BALANCE != 0). */
int misift_ctx_last_call_balanced(misift_ctx *ctx);
/* Diagnostics: synchronous calls of a frame or two whose descriptor launch was the last one (r05) but that then needed the
 * global-memory descriptor kernel after all, for a keypoint larger than the LDS window (next to never; MISIFT_FOLD_TAIL=0
 * always launches it). */
int misift_ctx_descr_big_fallbacks(misift_ctx *ctx);
int misift_ctx_sync(misift_ctx *ctx);
const char *misift_last_error(void);

void misift_default_options(misift_options *opt);
/* misift_options grows at its END (reference_cap was added in r05) and has no size member, so the caller's sizeof
 * travels with the call: the _sized entry points copy min(struct_size, the library's sizeof) bytes — fields the caller's
 * header does not know keep their current value (set) / are not written (get) — and nothing is read or written past the
 * caller's struct.  Code compiled against THIS header gets them through the two macros below; the plain exported
 * symbols stay for binaries built against the r04 header and touch the seven fields that header had, no more. */
void misift_default_options_sized(misift_options *opt, size_t struct_size);
int misift_set_options_sized(misift_ctx *ctx, const misift_options *opt, size_t struct_size);
int misift_get_options_sized(misift_ctx *ctx, misift_options *opt, size_t struct_size);
int misift_set_options(misift_ctx *ctx, const misift_options *opt);
int misift_get_options(misift_ctx *ctx, misift_options *opt);
#ifndef MISIFT_NO_OPTION_MACROS
#define misift_default_options(opt) misift_default_options_sized((opt), sizeof(*(opt)))
#define misift_set_options(ctx, opt) misift_set_options_sized((ctx), (opt), sizeof(*(opt)))
#define misift_get_options(ctx, opt) misift_get_options_sized((ctx), (opt), sizeof(*(opt)))
#endif

/* ------------------------------------------------------------------- memory */

/* cudaMalloc / cudaFree / cudaMemcpy as used by cudaSiftH.cu:234-264 and
 * cudaImage.cu:15-78. */
int misift_malloc(size_t bytes, void **out);
int misift_free(void *ptr);
int misift_memset(misift_ctx *ctx, void *ptr, int value, size_t bytes);
int misift_copy_h2d(misift_ctx *ctx, void *dst, const void *src, size_t bytes);
int misift_copy_d2h(misift_ctx *ctx, void *dst, const void *src, size_t bytes);
/* cudaMallocPitch: rows padded to a multiple of 128 floats (cudaImage.cu:24). */
int misift_image_alloc(int width, int height, float **d_out, int *pitch_floats);
/* cudaMemcpy2D, all strides in floats (cudaImage.cu:55-78). */
int misift_upload_2d(misift_ctx *ctx, float *d_dst, int dpitch, const float *h_src,
                     int hpitch, int width, int height);
int misift_download_2d(misift_ctx *ctx, float *h_dst, int hpitch, const float *d_src,
                       int dpitch, int width, int height);
/* Strided gather of `nfields` consecutive 32-bit fields starting at byte
 * `offset` of every 576-byte record (matching.cu:1195-1199). */
int misift_download_fields(misift_ctx *ctx, void *h_pts, const void *d_pts, int npts,
                           int offset, int nfields);

/* Scratch arena: cudaSiftH.cu:39-64 (AllocSiftTempMemory). Size in floats. */
size_t misift_scratch_floats(int width, int height, int num_octaves, int scale_up);

/* --------------------------------------------------------------- extraction */

/* cudaSiftH.cu:72-144 (ExtractSift) for one device-resident frame.
 * d_img: width x height floats, row stride `pitch` floats.
 * d_scratch: misift_scratch_floats() floats, or NULL to allocate per call.
 * d_pts: max_pts records.  *num_pts_out follows the reference's rule
 * numPts = min(counter[2*num_octaves], max_pts) (cudaSiftH.cu:115-116).
 * One host<->device sync (the count read-back), like the reference.
 * Completion: the call returns after the context's stream has been synchronised — every record is written and visible
 * to any stream, device or host copy, like the reference's blocking ExtractSift.  misift_ctx_set_early_return(ctx, 1)
 * (r05: opt-in; it was the default in r04) lets the synchronous calls return as soon as the call's last kernel has
 * handed the counts to the host through pinned memory (~10 us earlier): the records are then ordered only for work
 * enqueued on the CONTEXT'S stream afterwards (misift_match, misift_copy_d2h, the cudaSift.h shim's read-back);
 * a consumer on another stream or device must call misift_ctx_sync() first.  MISIFT_HOST_SPIN=0 / =1 in the
 * environment overrides either way.  The same holds for misift_match / misift_match_rows.
 * Any size from 1 x 1 and any num_octaves <= MISIFT_MAX_OCTAVES is accepted, like the reference (cudaSiftH.cu:72-167):
 * images under 16 x 16, or whose coarsest pyramid level is under 8 px, run on the dense per-level kernels (every access
 * clamped); a level that integer division has shrunk to 0 pixels is empty and skipped (tests/test_gpu_refemul.py:
 * same numPts, counters and keypoints as the emulated reference down to 1 x 1).  Only width, height < 16384
 * (after the doubling of scale_up) and max_pts >= 1 are required, else MISIFT_EINVAL. */
int misift_extract(misift_ctx *ctx, const float *d_img, int width, int height, int pitch,
                   int num_octaves, float init_blur, float thresh, float lowest_scale,
                   int scale_up, float *d_scratch, void *d_pts, int max_pts,
                   int *num_pts_out);

/* Same pipeline over a batch of independent frames in one launch sequence
 * (BASELINE config 4: frames shard across GPUs, one batch per device).
 * d_imgs: nframes images, frame stride `frame_stride` floats.
 * d_scratch: nframes * misift_scratch_floats() floats (or NULL).
 * d_pts: nframes * max_pts records, frame f at d_pts + f*max_pts.
 * num_pts_out: host array of nframes ints. */
int misift_extract_batch(misift_ctx *ctx, const float *d_imgs, int nframes,
                         size_t frame_stride, int width, int height, int pitch,
                         int num_octaves, float init_blur, float thresh,
                         float lowest_scale, float *d_scratch, void *d_pts,
                         int max_pts, int *num_pts_out);
/* As above but does not synchronise: counts stay on the device
 * (d_counts_out: nframes ints, may be NULL) so a caller can queue batches. */
int misift_extract_batch_async(misift_ctx *ctx, const float *d_imgs, int nframes,
                               size_t frame_stride, int width, int height, int pitch,
                               int num_octaves, float init_blur, float thresh,
                               float lowest_scale, float *d_scratch, void *d_pts,
                               int max_pts, int *d_counts_out);

/* As misift_extract_batch_async, and additionally packs the valid records of all
 * frames contiguously (frame after frame) into d_packed_out, ready for ONE
 * device-to-device / xGMI / PCIe transfer: d_counts_out[f] = numPts of frame f
 * (-1: that frame's candidate list overflowed), d_offsets_out[0..nframes] =
 * exclusive prefix sum of the non-negative counts (records).  In the default
 * (fused) mode the descriptor kernel writes the packed array itself — no extra
 * packing pass — and d_pts may be NULL (if given it is filled as well); calls
 * that run on the dense kernels (fused = 0, reference_cap, images under 16 x 16
 * or with a coarsest level under 8 px) need d_pts (MISIFT_EINVAL otherwise).  Nothing
 * synchronises; this is what the multi-GPU gather of SiftData (BASELINE
 * config 4) and the host pipeline send. */
int misift_extract_batch_packed_async(misift_ctx *ctx, const float *d_imgs, int nframes,
                                      size_t frame_stride, int width, int height, int pitch,
                                      int num_octaves, float init_blur, float thresh,
                                      float lowest_scale, float *d_scratch, void *d_pts,
                                      int max_pts, int *d_counts_out, int *d_offsets_out,
                                      void *d_packed_out);

/* As misift_extract_batch, but the frames are 8-bit (pitch and frame_stride in
 * bytes): the prefilter converts in registers, so the result is bit-identical
 * to an fp32 upload of the same pixel values (what mainSift.cpp:41-42 does on
 * the host with convertTo(CV_32FC1)) at a quarter of the PCIe and HBM read bytes. */
int misift_extract_batch_u8(misift_ctx *ctx, const unsigned char *d_imgs, int nframes,
                            size_t frame_stride, int width, int height, int pitch,
                            int num_octaves, float init_blur, float thresh,
                            float lowest_scale, float *d_scratch, void *d_pts,
                            int max_pts, int *num_pts_out);

/* ExtractSift's whole argument surface over a batch: 8-bit or fp32 frames (src_u8; pitch and frame_stride in source
 * elements) and scaleUp (cudaSiftH.cu:118-132: every frame is doubled first, lowestScale doubles, positions and
 * scales are halved at the end).  d_scratch: nframes * misift_scratch_floats(width, height, num_octaves, scale_up). */
int misift_extract_batch_ex(misift_ctx *ctx, const void *d_imgs, int src_u8, int nframes, size_t frame_stride,
                            int width, int height, int pitch, int num_octaves, float init_blur, float thresh,
                            float lowest_scale, int scale_up, float *d_scratch, void *d_pts, int max_pts,
                            int *num_pts_out);

/* ------------------------------------------------- host-fed pipeline
 * Streams batches of HOST frames through upload -> extraction -> read-back on three
 * HIP streams, replacing the blocking CudaImage::Download (cudaImage.cu:55-66) and the
 * blocking SiftPoint read-back of ExtractSift (cudaSiftH.cu:139-140).  Frames are tightly
 * packed (row stride = width, as CudaImage::Download assumes for h_data), fp32 or 8-bit.
 * Up to `depth` batches may be in flight; batches are collected in submission order.
 *   misift_pipe_submit   enqueues upload + extraction of `nframes` (<= batch_frames) frames and
 *                        returns at once; host_frames must stay valid until the batch is
 *                        collected (pinned memory from misift_host_alloc makes the upload async).
 *   misift_pipe_collect  waits for the oldest batch: counts_out[f] = numPts of frame f; the valid
 *                        records of all frames, frame after frame, are copied to host_records
 *                        (capacity in records; may be NULL to skip).  If a candidate list of the
 *                        fused scan overflowed, the batch is redone here with the exact dense
 *                        kernels (as misift_extract_batch does): nothing is dropped silently. */
typedef struct misift_pipe misift_pipe;
int misift_pipe_create(misift_ctx *ctx, int width, int height, int batch_frames, int src_u8,
                       int num_octaves, float init_blur, float thresh, float lowest_scale,
                       int max_pts, int depth, misift_pipe **out);
void misift_pipe_destroy(misift_pipe *pipe);
int misift_pipe_submit(misift_pipe *pipe, const void *host_frames, int nframes);
int misift_pipe_collect(misift_pipe *pipe, int *nframes_out, int *counts_out, void *host_records,
                        size_t capacity_records, size_t *nrecords_out);
int misift_pipe_pending(const misift_pipe *pipe);
/* Pinned host memory for frames / records (truly asynchronous copies). */
int misift_host_alloc(size_t bytes, void **out);
int misift_host_free(void *ptr);

/* Per-frame point counters of the last extraction, 17 per frame, in the
 * reference's layout (cudaSiftD.cu:14, protocol cudaSiftD.cu:1297-1300). */
int misift_get_counters(misift_ctx *ctx, int frame, unsigned int *counters17);
/* Diagnostic: all 64 words of one frame's counter block (the 17 reference counters, then per-octave candidate /
 * detection / duplicate counts and overflow flags); frame == frames of the last call: the call's flag block. */
int misift_get_counter_block(misift_ctx *ctx, int frame, unsigned int *words64);
int misift_set_counters(misift_ctx *ctx, int frame, const unsigned int *counters17);

/* ------------------------------------------------- stage-level entry points
 * The individual launch wrappers of cudaSiftH.cu:308-514, exposed so each
 * kernel can be checked against the oracle in isolation. */

/* LowPass (cudaSiftH.cu:406-435, LowPassBlock cudaSiftD.cu:1986-2037). */
int misift_lowpass(misift_ctx *ctx, const float *d_src, int width, int height, int spitch,
                   float *d_dst, int dpitch, float sigma);
/* LowPass and the first ScaleDown of the pyramid in one pass (what ExtractSift does
 * back to back, cudaSiftH.cu:112 + :153-154): dst = LowPass(src), dst2 = ScaleDown(dst),
 * bit-identical to the two separate calls.  Needs width % 4 == 0 and 16-byte aligned rows
 * (MISIFT_EINVAL otherwise; misift_extract falls back to the separate kernels by itself). */
int misift_lowpass_scaledown(misift_ctx *ctx, const float *d_src, int width, int height, int spitch,
                             float *d_dst, int dpitch, float sigma, float *d_dst2, int dpitch2);
/* ScaleDown (cudaSiftH.cu:308-338, cudaSiftD.cu:84-168): dst is (w/2,h/2). */
int misift_scaledown(misift_ctx *ctx, const float *d_src, int width, int height,
                     int spitch, float *d_dst, int dpitch);
/* ScaleUp (cudaSiftH.cu:340-351, cudaSiftD.cu:170-190): dst is (2w,2h). */
int misift_scaleup(misift_ctx *ctx, const float *d_src, int width, int height, int spitch,
                   float *d_dst, int dpitch);
/* PrepareLaplaceKernels (cudaSiftH.cu:439-458): fills 8*12*16 floats. */
int misift_laplace_taps(int num_octaves, float *taps_8x12x16);
/* LaplaceMulti (cudaSiftH.cu:460-487, LaplaceMultiMem cudaSiftD.cu:1753-1793):
 * 7 DoG planes, plane stride height*pitch.  `octave` = reference octave index
 * (num_octaves = finest … 1 = coarsest) selecting the tap table. */
int misift_laplace(misift_ctx *ctx, const float *d_base, int width, int height, int pitch,
                   int num_octaves, int octave, float *d_dog);
/* FindPointsMulti (cudaSiftH.cu:489-514, FindPointsMultiNew cudaSiftD.cu:1292-1431).
 * Appends to d_pts using the context counters of frame 0 (reset them with
 * misift_reset_counters first). */
int misift_reset_counters(misift_ctx *ctx, int max_pts);
int misift_findpoints(misift_ctx *ctx, const float *d_dog, int width, int height, int pitch,
                      float thresh, float edge_limit, float lowest_scale,
                      float subsampling, int octave, void *d_pts, int max_pts);
/* Fused LaplaceMulti + FindPointsMulti: same result, no DoG planes in HBM. */
int misift_dog_findpoints(misift_ctx *ctx, const float *d_base, int width, int height,
                          int pitch, int num_octaves, int octave, float thresh,
                          float edge_limit, float lowest_scale, float subsampling,
                          void *d_pts, int max_pts);
/* ComputeOrientations (cudaSiftH.cu:353-369, cudaSiftD.cu:972-1057). */
int misift_orientations(misift_ctx *ctx, const float *d_base, int width, int height,
                        int pitch, int octave, void *d_pts, int max_pts);
/* ExtractSiftDescriptors (cudaSiftH.cu:371-382, cudaSiftD.cu:308-417). */
int misift_descriptors(misift_ctx *ctx, const float *d_base, int width, int height,
                       int pitch, float subsampling, int octave, void *d_pts, int max_pts);
/* RescalePositions (cudaSiftH.cu:397-404, cudaSiftD.cu:753-761). */
int misift_rescale_positions(misift_ctx *ctx, void *d_pts, int npts, float scale);

/* ----------------------------------------------------------------- matching */

/* MatchSiftData (matching.cu:1090-1206; CleanMatches :289, FindMaxCorr10
 * :301-397) on device-resident records: fills score, ambiguity, match,
 * match_xpos, match_ypos of d_pts1[0..n1).  fp32 MFMA, k-ordered so every
 * score is bit-identical to the reference's sequential FMA chain. */
int misift_match(misift_ctx *ctx, void *d_pts1, int n1, const void *d_pts2, int n2);
/* Row-block form for BASELINE config 5: rows [row_begin,row_begin+row_count)
 * of set 1 against all of set 2 (each GPU takes one row block). */
int misift_match_rows(misift_ctx *ctx, void *d_pts1, int row_begin, int row_count,
                      const void *d_pts2, int n2);

/* Batches in flight (no reference counterpart: ExtractSift is synchronous, cudaSiftH.cu:72-144).  A context is one
 * in-order pipeline; with K > 1 it owns K child pipelines (own stream, counters, candidate lists, detection staging) and
 * misift_extract_batch_packed_async hands consecutive calls to consecutive pipelines, so the HBM-bound prefilter of one
 * batch runs beside the VALU-bound kernels of another and launch tails are filled (+8-9 % frames/s at K = 3-4 on 64 x
 * 1080p).  Contract with K > 1:
 *   - every call still starts behind whatever was enqueued on the context stream before it (a marker is recorded there);
 *   - its results are NOT ordered on the context stream: observe them with misift_ctx_record_batch (an event of the
 *     caller's behind the most recent batch), misift_ctx_wait_batch (a stream of the caller's waits for it),
 *     misift_gather_post (marks the most recent batch) or misift_ctx_sync;
 *   - the scratch arena and the output buffers of a call must stay untouched until that batch is done: rotate >= K sets.
 * K = 1 (default) is the plain in-order context.  Also MISIFT_BATCHES_IN_FLIGHT at context creation.  Changing K drains
 * the context. */
int misift_ctx_set_batches_in_flight(misift_ctx *ctx, int k);
int misift_ctx_get_batches_in_flight(misift_ctx *ctx);
int misift_ctx_wait_batch(misift_ctx *ctx, void *stream);
/* Record the caller's hipEvent_t (passed as void*) behind the most recent batch, on the stream that batch runs on; the
 * caller then queries / waits on the event as it likes.  Preferable to misift_ctx_wait_batch when many side streams would
 * oversubscribe the hardware queues (a waiting stream that shares a queue with a pipeline holds that pipeline up). */
int misift_ctx_record_batch(misift_ctx *ctx, void *hip_event);

/* ------------------------------------------------------------ multi-GPU
 * The reference is single-GPU (InitCuda picks ONE device, cudaSiftH.cu:19-37); BASELINE configs 4 and 5 shard
 * frames / matcher rows over the GPUs of a node.  One misift_ctx per device (one host thread or process each) and
 * one misift_comm per context; collectives run on RCCL over xGMI, bound at run time (no link dependency: a
 * single-GPU caller never loads RCCL).  Nothing here touches the extraction data path — frames shard
 * embarrassingly; the only exchanges are the gather of SiftData after compute (config 4) and the set-2 /
 * result all-gathers around the matcher sweep (config 5). */
typedef struct misift_comm misift_comm;
#define MISIFT_COMM_ID_BYTES 128
/* Rendezvous like ncclGetUniqueId / ncclCommInitRank: rank 0 makes the id, ships its 128 bytes to the other
 * ranks by any means (MPI, a file, torch.distributed's store), every rank then creates its communicator on its
 * context's device.  misift_comm_adopt wraps a caller-supplied ncclComm_t instead (not destroyed with the comm). */
int misift_comm_unique_id(void *id128);
int misift_comm_create(misift_ctx *ctx, int nranks, int rank, const void *id128, misift_comm **out);
int misift_comm_adopt(misift_ctx *ctx, void *nccl_comm, misift_comm **out);
/* HOST communicator: no device and no context — counts, packed records and the receive buffer are HOST memory and the
 * five primitives of the exchange are the caller's callbacks (return 0 on success; send / recv may only queue, group_end
 * completes everything queued since the last call).  misift_comm_barrier, misift_gather_post (ctx = NULL),
 * misift_gather_test and misift_gather_complete run the SAME code above the transport as on RCCL — count staging,
 * per-rank record counts and offsets, root placement, -1 frames, the collective MISIFT_ENOMEM decision — which is what
 * the CPU-only suite drives over torch.distributed / gloo (tests/test_dist_cpu.py); misift_match_sharded needs a
 * device communicator.  Not a performance path. */
typedef struct misift_host_transport {
  void *user;
  int (*allgather)(void *user, const void *send, void *recv, size_t bytes_per_rank);
  int (*send)(void *user, const void *buf, size_t bytes, int peer);
  int (*recv)(void *user, void *buf, size_t bytes, int peer);
  int (*group_end)(void *user);
} misift_host_transport;
int misift_comm_create_host(int nranks, int rank, const misift_host_transport *transport, misift_comm **out);
/* In-process LOOPBACK WORLD ("fake N ranks on one GPU").  The reference has no multi-device code at
 * all (cudaSiftH.cu:19-37 picks one device), so nothing in it corresponds to this; it exists so that every N > 1
 * branch behind misift_gather_* / misift_match_sharded runs on the hardware a developer has: N communicators, one host
 * thread each, normally N contexts of ONE device, exchanging through a shared rendezvous object with device-to-device
 * copies instead of RCCL.  Same entry points, same code above the transport; functional only, never a scaling number.
 * A rank that never makes the matching call makes its peers fail with an error after MISIFT_LOOPBACK_TIMEOUT_S (60)
 * seconds instead of hanging; mismatching message sizes between the two sides of a send/recv are an error. */
typedef struct misift_loopback_world misift_loopback_world;
int misift_loopback_world_create(int nranks, misift_loopback_world **out);
void misift_loopback_world_destroy(misift_loopback_world *world);    /* after all its communicators */
int misift_comm_create_loopback(misift_ctx *ctx, misift_loopback_world *world, int rank, misift_comm **out);
void misift_comm_destroy(misift_comm *comm);
int misift_comm_rank(const misift_comm *comm);
int misift_comm_size(const misift_comm *comm);
int misift_comm_barrier(misift_comm *comm);        /* all ranks have arrived (host-blocking) */

/* BASELINE config 4 — gather of SiftData on `root`, pipelined under the next batches' extraction.
 *   misift_gather_post      right after misift_extract_batch_packed_async: remembers that batch's device buffers
 *                           (d_counts[nframes], d_packed) in `slot` (0..MISIFT_GATHER_SLOTS-1) and marks the point on
 *                           the stream of `ctx` where they are complete; `ctx` is the context that extracted the batch —
 *                           the communicator's own or any other context of the same device (several contexts per GPU
 *                           keep several batches in flight).  Returns at once.
 *   misift_gather_complete  on the communicator's own high-priority stream: all-gather of the per-frame counts
 *                           (nframes ints per rank, the same nframes on every rank), then ONE point-to-point message
 *                           per sender carrying exactly its valid 576-byte records (xGMI is a full mesh: 7 senders use
 *                           7 distinct links into the root).  h_all_counts[nranks*nframes] (host, every rank; -1 = that
 *                           frame's candidate list overflowed, no records); on the root the records of rank r land at
 *                           d_recv + h_rank_offsets[r] records (h_rank_offsets: nranks+1 entries, host, optional).
 *                           capacity_records = room at d_recv on the root, in records; pass the SAME value on every
 *                           rank: all ranks see all counts, so all of them return MISIFT_ENOMEM without exchanging
 *                           anything when the records do not fit (no rank is left waiting for a message).
 *                           Blocks the host until the transfer is done: the slot's buffers may be reused. */
#define MISIFT_GATHER_SLOTS 8
int misift_gather_post(misift_ctx *ctx, misift_comm *comm, int slot, const int *d_counts, int nframes,
                       const void *d_packed);
int misift_gather_complete(misift_comm *comm, int slot, int root, int *h_all_counts, void *d_recv,
                           size_t capacity_records, size_t *h_rank_offsets);
/* Non-blocking: *ready = 1 once the batch posted in `slot` has finished on the GPU (misift_gather_complete then only
 * waits for the exchange itself), 0 while its kernels are still running.  Lets a caller poll instead of parking a
 * thread in misift_gather_complete. */
int misift_gather_test(misift_comm *comm, int slot, int *ready);
/* Payload bytes this rank has received / sent over the links since the communicator was created (either may be NULL). */
int misift_comm_wire_bytes(misift_comm *comm, unsigned long long *received, unsigned long long *sent);

/* BASELINE config 5 — MatchSiftData (matching.cu:1090-1206) with set 1 split into row blocks, one per rank.
 * d_rows1: this rank's row block (row_count records, updated in place like misift_match); d_shard2: this rank's
 * shard of set 2 (shard_count records; row_count and shard_count must be equal on all ranks — pad the last block).
 * Steps: the shard is packed into MATCH COLUMNS — what the sweep reads of a record and nothing else:
 * MISIFT_MATCH_COLUMN_BYTES = 528 = descriptor[128], xpos, ypos, 2 reserved words (r04: 8 % less on the wire than the
 * 576-byte records r03 shipped) — and all-gathered into d_set2_all (nranks*shard_count columns, rank order; a buffer
 * sized for that many RECORDS is more than enough), then the fp32-MFMA sweep of the rank's rows over all of it, then
 * the all-gather of the 12-byte results {float score, float ambiguity, int match} of every row into d_results_all
 * (nranks*row_count entries, may be NULL).  Match indices refer to the gathered set 2.  Returns with everything in
 * place (matching.cu:1191).
 * Aliasing: d_shard2 must NOT overlap d_set2_all (the shard is re-packed into it; MISIFT_EINVAL otherwise — also with
 * one rank), and after the call d_set2_all holds 528-byte match columns, not SiftPoint records: look matched records
 * up in your own copy of set 2. */
#define MISIFT_MATCH_COLUMN_BYTES 528
int misift_match_sharded(misift_ctx *ctx, misift_comm *comm, void *d_rows1, int row_count, const void *d_shard2,
                         int shard_count, void *d_set2_all, void *d_results_all);

/* FindHomography (matching.cu:1000-1087): RANSAC over stored matches. */
int misift_find_homography(misift_ctx *ctx, const void *d_pts, int npts, float *homography9,
                           int *num_matches, int num_loops, float min_score,
                           float max_ambiguity, float thresh);

/* ImproveHomography (geomFuncs.cpp:6-72 — a host function over SiftData.h_data in the reference) on device-resident
 * records: num_loops rounds of least squares over the matches that pass the gates and currently reproject within
 * `thresh`, then match_error of every record is written (device) and *num_fit = records within thresh.
 * homography9: in = start (e.g. from misift_find_homography), out = refined, [8] = 1.  Sums in the reference's order:
 * bit-identical to the reference's result. */
int misift_improve_homography(misift_ctx *ctx, void *d_pts, int npts, float *homography9, int num_loops,
                              float min_score, float max_ambiguity, float thresh, int *num_fit);

/* cudaMallocManaged as used by the reference's MANAGEDMEM build flavour (cudaSiftH.cu:239-240): one pointer valid on
 * host and device (SiftData.m_data). */
int misift_malloc_managed(size_t bytes, void **out);

/* Test-only entry point (no reference counterpart): the device copies of the written-out elementary functions that
 * replace CUDA's exp2f / atan2f / expf / __sinf,__cosf (cudaSiftD.cu:1417, 1008, 987, 331-332), evaluated on n inputs.
 * fn: 0 = exp2(x), 1 = atan2(y, x), 2 = exp(x), 3 = sin/cos(x) -> d_out, d_out2.  Device pointers; synchronous. */
int misift_test_elementary(misift_ctx *ctx, int fn, const float *d_x, const float *d_y, float *d_out, float *d_out2,
                           int n);

/* Test-only: MatchSiftData with the column sweep cut the way misift_match_sharded cuts it (64-column super-tiles
 * [own_tile_begin, own_tile_end) in a first launch, all others in a second, one merge).  Same results as misift_match. */
int misift_test_match_split(misift_ctx *ctx, void *d_pts1, int n1, const void *d_pts2, int n2, int own_tile_begin,
                            int own_tile_end);

/* Test-only, host-only (no device needed): the matcher's column-chunk plan for n1 x n2 on a chip of num_cus CUs. */
int misift_test_match_plan(int num_cus, int n1, int n2, int *nchunks, int *tiles_per_chunk, int *ntiles);

/* Test-only, host-only: how the balanced per-keypoint launches (MISIFT_BALANCE=1) split `nblocks` workgroups among
 * `nframes` frames holding points[f] keypoints: shares[f] = 1 + floor((nblocks - nframes) * points[f] / sum), the formula
 * frame_shares_kernel evaluates on the device.  nblocks >= nframes. */
int misift_test_frame_shares(int nblocks, int nframes, const unsigned *points, int *shares);

/* Test / tuning only: developer knobs of a context — launch shapes (segment lengths, workgroups per CU, dynamic-LDS padding)
 * and path selection (spatial binning, balanced per-keypoint launches, embedded ScaleDown chain, split pyramid tail, hipGraph
 * replay, LDS-window vs global-memory sampling, descr_big threshold ...).  Defaults are the measured optimum
 * (profiles/r05_knob_sweep.txt) and a production process never changes them: the library reads none of the corresponding
 * MISIFT_* environment variables unless MISIFT_TUNABLES=1 is set (tools/, the variant runs of the test suite).
 * misift_test_knob_names() = "knob=ENVIRONMENT_VARIABLE,..." of everything there is.  Applies to the context's pipelines
 * (misift_ctx_set_batches_in_flight) as well, also to those built later. */
int misift_test_set_knob(misift_ctx *ctx, const char *name, double value);
const char *misift_test_knob_names(void);

/* Test-only: guard mode (out-of-bounds policing in the test build).  While it is on, every device
 * allocation the library makes — misift_malloc for the caller, and its own counters, candidate lists, detection staging,
 * block tables, matcher scratch, pipeline buffers — carries 64 KiB of a byte pattern in front of and behind the payload,
 * and the payload starts out filled with 0xFF (NaN as a float, -1 as an int: nothing may rely on fresh memory being zero).
 * misift_test_check_guards synchronises the device, verifies the bands of every live guarded allocation — those freed
 * since the previous check were verified as they were freed — and returns the number of damaged ones (0 = intact;
 * misift_last_error() names the first), negative on error.  MISIFT_GUARD=1 in the
 * environment switches the mode on from the first allocation.  Allocations made while the mode is off are not guarded. */
int misift_test_set_guard(int on);                 /* returns the previous mode */
int misift_test_check_guards(int *allocations);    /* allocations (optional): how many were checked */

/* ------------------------------------------------------------------- timing */

/* TimerGPU (cudautils.h:61-81): event pair on the context stream. */
int misift_timer_start(misift_ctx *ctx);
int misift_timer_stop_ms(misift_ctx *ctx, float *ms_out);

/* Per-kernel accumulated HIP-event timings of the context's launches
 * (enabled with misift_profile_enable; used by bench.py for the roofline).
 * names/ms/calls: arrays of `cap` entries; *n_out = entries filled. */
int misift_profile_enable(misift_ctx *ctx, int on);
int misift_profile_reset(misift_ctx *ctx);
int misift_profile_read(misift_ctx *ctx, int cap, char (*names)[32], float *total_ms,
                        int *calls, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* MISIFT_H */

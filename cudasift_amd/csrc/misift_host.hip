// misift_host.hip — host side of libmisift.so: context, memory, launch orchestration
// and the extern "C" entry points declared in include/misift.h.
//
// Orchestration follows the reference's host code (cudaSiftH.cu:72-232) in WHAT is
// computed and in the counter protocol, not in how it is scheduled: all kernels take a
// frame dimension so a whole batch of frames goes through each pyramid level in one
// launch; nothing is allocated, uploaded or synchronised inside the launch sequence
// (taps travel as kernel arguments, counters are reset with one async memset, the only
// host<->device sync is the final count read-back).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "common.hpp"
#include "chain.hpp"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void misift_set_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *misift_last_error(void) { return g_err; }

#define ARG_CHECK(cond)                                                       \
  do {                                                                        \
    if (!(cond)) {                                                            \
      misift_set_error("%s: invalid argument: %s", __func__, #cond);          \
      return MISIFT_EINVAL;                                                   \
    }                                                                         \
  } while (0)

// ------------------------------------------------------------------ roctx
// Named ranges for rocprofv3 --marker-trace / rocprof-sys: one range per entry-point call and one
// per kernel launch.  The marker library is bound at run time, and only when a profiler has already mapped it into
// the process or MISIFT_ROCTX=1 asks for it — ordinary runs pay one predictable branch per launch.
#include <dlfcn.h>
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)(void) = nullptr;
static int g_roctx_state = 0;          // 0 = untried, 1 = bound, -1 = off
static void roctx_bind_once(void)
{
  g_roctx_state = -1;
  const char *names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
  const char *env = getenv("MISIFT_ROCTX");
  if (env && atoi(env) == 0) return;
  void *h = nullptr;
  for (const char *n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  if (!h && env && atoi(env) != 0)
    for (const char *n : names)
      if (!h) h = dlopen(n, RTLD_NOW);
  if (!h) return;
  *(void **)(&g_roctx_push) = dlsym(h, "roctxRangePushA");
  *(void **)(&g_roctx_pop) = dlsym(h, "roctxRangePop");
  if (g_roctx_push && g_roctx_pop) g_roctx_state = 1;
}
static void roctx_bind(void)
{
  static std::once_flag once;          // contexts are created from several host threads (one per device)
  std::call_once(once, roctx_bind_once);
}
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char *name) : on(g_roctx_state > 0) { if (on) g_roctx_push(name); }
  ~RoctxRange() { if (on) g_roctx_pop(); }
};

// --------------------------------------------------------------- profiling
struct PendingProf { int slot; hipEvent_t a, b; };
// Identity of one synchronous extraction call; a repeated call (the reference demo extracts the same image
// 1000 times, mainSift.cpp:64-69; a tracker calls with the same buffers every frame) replays a captured hipGraph.
struct CallKey {
  const void *imgs; float *scratch; void *pts;
  int src_u8, nframes, width, height, pitch, num_octaves, scale_up, max_pts, fused, texfrac, fixnum, determ, refcap, alloc_gen;
  long long frame_stride;
  float init_blur, thresh, lowest_scale;
  // field by field: the struct has padding, and plain assignment need not preserve padding bytes
  bool operator==(const CallKey &o) const
  {
    return imgs == o.imgs && scratch == o.scratch && pts == o.pts && src_u8 == o.src_u8 && nframes == o.nframes &&
           width == o.width && height == o.height && pitch == o.pitch && num_octaves == o.num_octaves &&
           scale_up == o.scale_up && max_pts == o.max_pts && fused == o.fused && texfrac == o.texfrac &&
           fixnum == o.fixnum && determ == o.determ && refcap == o.refcap && alloc_gen == o.alloc_gen && frame_stride == o.frame_stride &&
           init_blur == o.init_blur && thresh == o.thresh && lowest_scale == o.lowest_scale;
  }
};
struct CtxExtra {
  std::vector<PendingProf> pending;
  std::vector<hipEvent_t> pool;
  // hipGraph replay of the launch sequence of misift_extract_enqueue (launch-latency bound for a single frame)
  int graph_mode = 0;                 // off by default (measured slower, see misift.h); MISIFT_GRAPH=1 / misift_ctx_set_graph_replay
  bool have_last = false, have_graph = false;
  CallKey last_key, graph_key;
  hipGraphExec_t gexec = nullptr;
  hipStream_t gstream = nullptr;      // capture / replay stream (capture is not allowed on the null stream)
  hipEvent_t gev_in = nullptr, gev_out = nullptr;
  // batches in flight (misift_ctx_set_batches_in_flight): K child contexts = K in-order pipelines (stream, counters,
  // candidate lists, detection staging each) that misift_extract_batch_packed_async rotates over
  std::vector<misift_ctx *> lanes;
  std::vector<hipEvent_t> lane_done;   // 2K events: done[ticket % 2K] is recorded behind batch `ticket` on its lane's stream
  hipEvent_t ev_in = nullptr;          // recorded on the caller's stream at every call: the lane starts behind it
  hipEvent_t ev_single = nullptr;      // K = 1: misift_ctx_wait_batch records this one on the context stream
  unsigned long long ticket = 0;
  misift_ctx *last_lane = nullptr;     // the pipeline that took the most recent batch (nullptr: the context itself)
  hipEvent_t last_done = nullptr;
  hipStream_t own_stream = nullptr;    // a child context owns its stream
  std::vector<std::pair<std::string, double>> knobs;   // misift_test_set_knob calls so far: replayed on pipelines built later
  // host-side tap tables of the last call's (num_octaves, init_blur)
  int taps_noct = -1;
  float taps_table[8 * 12 * 16];
  float k9_blur = -1.0f, k9[9], k5[5];
  bool k5_done = false;
};
static CtxExtra *extra(misift_ctx *ctx);

LaunchScope::LaunchScope(misift_ctx *c, const char *n) : ctx(c), name(n)
{
  if (g_roctx_state > 0) g_roctx_push(n);
  if (ctx->profile) {
    CtxExtra *x = extra(ctx);
    hipEvent_t a, b;
    if (x->pool.size() >= 2) {
      a = x->pool.back(); x->pool.pop_back();
      b = x->pool.back(); x->pool.pop_back();
    } else {
      hipEventCreate(&a);
      hipEventCreate(&b);
    }
    int slot = -1;
    for (int i = 0; i < ctx->nprof; i++)
      if (!strcmp(ctx->prof[i].name, n)) slot = i;
    if (slot < 0 && ctx->nprof < 32) {
      slot = ctx->nprof++;
      strncpy(ctx->prof[slot].name, n, 31);
      ctx->prof[slot].name[31] = 0;
      ctx->prof[slot].total_ms = 0;
      ctx->prof[slot].calls = 0;
    }
    hipEventRecord(a, ctx->stream);
    x->pending.push_back({slot, a, b});
  }
}

int LaunchScope::finish()
{
  if (g_roctx_state > 0) g_roctx_pop();
  if (ctx->profile) {
    CtxExtra *x = extra(ctx);
    hipEventRecord(x->pending.back().b, ctx->stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    misift_set_error("launch of %s failed: %s", name, hipGetErrorString(e));
    return MISIFT_EHIP;
  }
  return MISIFT_OK;
}

static int resolve_profile(misift_ctx *ctx)
{
  CtxExtra *x = extra(ctx);
  if (x->pending.empty()) return MISIFT_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto &p : x->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess && p.slot >= 0) {
      ctx->prof[p.slot].total_ms += ms;
      ctx->prof[p.slot].calls++;
    }
    x->pool.push_back(p.a);
    x->pool.push_back(p.b);
  }
  x->pending.clear();
  return MISIFT_OK;
}

// Streams of one process share HIP's hardware queues (4 by default): the communication stream then queues behind
// the extraction stream's kernels and a pipelined gather loses ~20 %.  Say so once.
static void warn_hw_queues_once(const char *who)
{
  const char *e = getenv("GPU_MAX_HW_QUEUES");
  if (e && atoi(e) >= 8) return;
  // a warning, not an error: stderr only — misift_last_error() of a successful call stays empty
  if (!getenv("MISIFT_QUIET"))
    fprintf(stderr, "misift: warning: %s: GPU_MAX_HW_QUEUES is %s; with fewer than 8 hardware queues the communication / "
                    "copy streams share a queue with the extraction stream (set GPU_MAX_HW_QUEUES=8 before the first HIP "
                    "call, or create the first misift context before any other HIP call: it sets it)\n", who,
            e ? e : "unset (HIP default: 4)");
}
void misift_warn_hw_queues(const char *who)
{
  static std::once_flag once;            // communicators are created from several host threads at the same time
  std::call_once(once, warn_hw_queues_once, who);
}

// ----------------------------------------------------------------- context
struct CtxFull {
  misift_ctx c;
  CtxExtra x;
};
static CtxExtra *extra(misift_ctx *ctx) { return &reinterpret_cast<CtxFull *>(ctx)->x; }

// MISIFT_DEVICES="2,3" (SURVEY section 5): device i of this library = HIP device list[i], the others do not exist for
// it — like HIP_VISIBLE_DEVICES but for libmisift.so only (a process that shares the GPUs with another runtime keeps
// its own numbering).  Unset or empty: every HIP device, identity.  Read once.
// HIP reads GPU_MAX_HW_QUEUES when it initialises; a C++ caller that never heard of it gets 4 hardware queues and loses
// ~20 % through a communicator or a host-fed pipe.  If the variable is unset when the library makes its
// first HIP call (device enumeration), set it — it takes effect when this is also the process's first HIP call (the normal case for
// a program that uses the library through cudaSift.h), and is harmless otherwise.  Never overrides the caller's value.
static void default_hw_queues_once(void)
{
  if (!getenv("GPU_MAX_HW_QUEUES") && !getenv("MISIFT_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

static std::vector<int> build_device_map(void)
{
  default_hw_queues_once();            // before the library's first HIP call
  std::vector<int> m;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  const char *e = getenv("MISIFT_DEVICES");
  if (e && *e) {
    for (const char *p = e; *p;) {
      char *end = nullptr;
      const long d = strtol(p, &end, 10);
      if (end == p) break;
      if (d >= 0 && d < n) m.push_back((int)d);
      p = (*end == ',') ? end + 1 : end;
      if (*end && *end != ',') break;
    }
  } else {
    for (int i = 0; i < n; i++) m.push_back(i);
  }
  return m;
}
static const std::vector<int> &device_map(void)
{
  // C++11 magic static: built exactly once even when several host threads create their contexts at the same time
  // (one thread per device is how the multi-GPU entry points are meant to be driven)
  static const std::vector<int> m = build_device_map();
  return m;
}
static int physical_device(int logical)
{
  const std::vector<int> &m = device_map();
  return (logical >= 0 && logical < (int)m.size()) ? m[(size_t)logical] : -1;
}

extern "C" int misift_device_count(void) { return (int)device_map().size(); }

extern "C" int misift_device_info(int device, char *name, int name_len, int *mem_clock_khz, int *bus_width_bits,
                                  size_t *total_mem_bytes, int *num_cus, int *lds_bytes_per_block)
{
  ARG_CHECK(physical_device(device) >= 0);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, physical_device(device)));
  if (name && name_len > 0) {
    // ROCm 7.2 leaves prop.name empty for some Instinct parts: fall back to the ISA name
    if (prop.name[0]) snprintf(name, name_len, "%s", prop.name);
    else snprintf(name, name_len, "AMD Instinct (%s)", prop.gcnArchName);
  }
  if (mem_clock_khz) *mem_clock_khz = prop.memoryClockRate;
  if (bus_width_bits) *bus_width_bits = prop.memoryBusWidth;
  if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (lds_bytes_per_block) *lds_bytes_per_block = (int)prop.sharedMemPerBlock;
  return MISIFT_OK;
}

extern "C" int misift_device_arch(int device, char *arch, int arch_len)
{
  ARG_CHECK(arch && arch_len > 0 && physical_device(device) >= 0);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, physical_device(device)));
  snprintf(arch, arch_len, "%s", prop.gcnArchName);
  return MISIFT_OK;
}

#undef misift_default_options
static void default_options_full(misift_options *opt);
extern "C" void misift_default_options_sized(misift_options *opt, size_t struct_size)
{
  if (!opt) return;
  misift_options o;
  default_options_full(&o);
  memcpy(opt, &o, struct_size < sizeof(o) ? struct_size : sizeof(o));
}
extern "C" void misift_default_options(misift_options *opt)      // binaries built against the r04 header: its seven fields
{
  misift_default_options_sized(opt, 7 * sizeof(int));
}
static void default_options_full(misift_options *opt)
{
  memset(opt, 0, sizeof(*opt));
  opt->texfrac_bits = 8;
  opt->fix_numpts = 0;
  opt->match_full = 0;
  opt->match_exact_top2 = 0;
  opt->quiet = 0;
  opt->fused = 1;
  const char *e;
  if ((e = getenv("MISIFT_TEXFRAC_BITS"))) opt->texfrac_bits = atoi(e);
  if ((e = getenv("MISIFT_FIX_NUMPTS"))) opt->fix_numpts = atoi(e);
  if ((e = getenv("MISIFT_MATCH_FULL"))) opt->match_full = atoi(e);
  if ((e = getenv("MISIFT_MATCH_EXACT_TOP2"))) opt->match_exact_top2 = atoi(e);
  if ((e = getenv("MISIFT_QUIET"))) opt->quiet = atoi(e);
  if ((e = getenv("MISIFT_FUSED"))) opt->fused = atoi(e);
  if ((e = getenv("MISIFT_DETERMINISTIC"))) opt->deterministic = atoi(e) != 0;
  opt->reference_cap = 0;
  if ((e = getenv("MISIFT_REFERENCE_CAP"))) opt->reference_cap = atoi(e) != 0;
}

// ---- developer / test knobs (launch shapes, path selection).  ONE table: knob name <-> the environment variable that sets it
// when MISIFT_TUNABLES=1.  README.md lists them; tests/test_cabi_cpu.py checks that list against this table.
struct KnobName { const char *name, *env; };
static const KnobName KNOBS[] = {
  {"graph", "MISIFT_GRAPH"}, {"split_tail", "MISIFT_SPLIT_TAIL"}, {"bin", "MISIFT_BIN"}, {"descr_occ", "MISIFT_DESCR_OCC"},
  {"tile", "MISIFT_TILE"}, {"tile_descr", "MISIFT_TILE_DESCR"}, {"tile_orient", "MISIFT_TILE_ORIENT"},
  {"orient_blocks", "MISIFT_ORIENT_BLOCKS"}, {"point_blocks", "MISIFT_POINT_BLOCKS"}, {"strip_waves", "MISIFT_STRIP_WAVES"},
  {"scan_waves", "MISIFT_SCAN_WAVES"}, {"chain_frames", "MISIFT_CHAIN_FRAMES"}, {"chain_embed", "MISIFT_CHAIN_EMBED"},
  {"chain_wait_us", "MISIFT_CHAIN_WAIT_US"}, {"bin_min_frames", "MISIFT_BIN_MIN_FRAMES"}, {"small_frames", "MISIFT_SMALL_FRAMES"},
  {"balance", "MISIFT_BALANCE"}, {"fold_tail", "MISIFT_FOLD_TAIL"}, {"patch_reach", "MISIFT_TEST_PATCH_REACH"},
  {"lds_pad_lpd", "MISIFT_LDS_PAD_LPD"}, {"lds_pad_scan", "MISIFT_LDS_PAD_SCAN"}, {"lds_pad_orient", "MISIFT_LDS_PAD_ORIENT"},
  {"lds_pad_descr", "MISIFT_LDS_PAD_DESCR"}, {"lowpass_tile", "MISIFT_LOWPASS_TILE"}, {"strip_rows_small", "MISIFT_STRIP_ROWS_SMALL"},
  {"scan_rows_small_coarse", "MISIFT_SCAN_ROWS_SMALL_COARSE"}, {"scan_rows_small", "MISIFT_SCAN_ROWS_SMALL"},
  {"host_spin", "MISIFT_HOST_SPIN"}, {"refcap_limit", "MISIFT_TEST_REFCAP_LIMIT"}, {"fuse_orient", "MISIFT_FUSE_ORIENT"},
  {"fuse_wait_us", "MISIFT_FUSE_WAIT_US"},
};
static bool tunables_from_env()
{
  const char *e = getenv("MISIFT_TUNABLES");
  return e && atoi(e) != 0;
}
static int apply_knob(misift_ctx *ctx, const char *name, double v)
{
  const int i = (int)v;
  const auto is = [&](const char *n) { return strcmp(name, n) == 0; };
  if (is("graph")) extra(ctx)->graph_mode = i != 0;
  else if (is("split_tail")) ctx->split_tail = i;
  else if (is("bin")) ctx->bin_detections = i != 0;
  else if (is("descr_occ")) ctx->descr_occ = i;
  else if (is("tile")) ctx->tile_descr = ctx->tile_orient = i != 0;
  else if (is("tile_descr")) ctx->tile_descr = i != 0;
  else if (is("tile_orient")) ctx->tile_orient = i != 0;
  else if (is("orient_blocks")) ctx->orient_blocks_per_cu = i > 0 ? i : 6;
  else if (is("point_blocks")) ctx->point_blocks_per_cu = i > 0 ? i : 8;
  else if (is("strip_waves")) ctx->strip_waves_per_cu = i > 0 ? i : 16;
  else if (is("scan_waves")) ctx->scan_waves_per_cu = i > 0 ? i : 16;
  else if (is("chain_frames")) ctx->chain_max_frames = i;
  else if (is("chain_embed")) ctx->chain_embed = i != 0;
  else if (is("chain_wait_us")) ctx->chain_wait_ticks = (unsigned)(v * 100.0);
  else if (is("bin_min_frames")) ctx->bin_min_frames = i;
  else if (is("small_frames")) ctx->small_frames = i;
  else if (is("balance")) ctx->balance_frames = i != 0;
  else if (is("fold_tail")) ctx->fold_descr_tail = i != 0;
  else if (is("patch_reach")) { if (v > 0.0 && v < 17.9) ctx->patch_reach = (float)v; else if (v >= 17.9) ctx->patch_reach = 17.9f; }
  else if (is("lds_pad_lpd")) ctx->lds_pad_lpd = i;
  else if (is("lds_pad_scan")) ctx->lds_pad_scan = i;
  else if (is("lds_pad_orient")) ctx->lds_pad_orient = i;
  else if (is("lds_pad_descr")) ctx->lds_pad_descr = i;
  else if (is("lowpass_tile")) ctx->lowpass_tile = i != 0;
  else if (is("strip_rows_small")) ctx->strip_rows_small = i >= 2 ? i / 2 * 2 : 6;
  else if (is("scan_rows_small_coarse")) ctx->scan_rows_small_coarse = i > 0 ? i : 2;
  else if (is("scan_rows_small")) ctx->scan_rows_small = i > 0 ? i : 4;
  else if (is("host_spin")) ctx->host_spin = i != 0;
  else if (is("refcap_limit")) ctx->refcap_limit = i >= 0 && i <= 240 ? i : 32;
  else if (is("fuse_orient")) ctx->fuse_orient = i != 0;
  else if (is("fuse_wait_us")) ctx->fuse_wait_ticks = (unsigned)(v * 100.0);
  else { misift_set_error("misift_test_set_knob: unknown knob '%s'", name); return MISIFT_EINVAL; }
  return MISIFT_OK;
}
// Test / tuning entry point: set one knob of this context (and of its pipelines behind misift_ctx_set_batches_in_flight).
// name = NULL: *names_out (optional) receives the comma-separated list "knob=ENV,..." of everything there is.
extern "C" int misift_test_set_knob(misift_ctx *ctx, const char *name, double value);
extern "C" const char *misift_test_knob_names(void)
{
  static std::string all;
  if (all.empty())
    for (const KnobName &k : KNOBS) all += std::string(all.empty() ? "" : ",") + k.name + "=" + k.env;
  return all.c_str();
}

extern "C" void misift_ctx_destroy(misift_ctx *ctx);
// Everything misift_ctx_create sets up after `new CtxFull()`; on any failure the caller destroys the
// half-built context (misift_ctx_destroy tolerates null members), so nothing leaks.
static int ctx_init(misift_ctx *ctx, CtxFull *f, int device, void *stream)
{
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  default_options_full(&ctx->opt);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // ---- launch shapes and path selection: DEFAULTS ONLY.  The developer / test knobs that change them are applied by
  // apply_knob() — through misift_test_set_knob, or from the environment when MISIFT_TUNABLES=1 (tools/, the variant runs of
  // the test suite); a production process reads none of those variables (r06: they were 30 getenv calls right here).
  // work decomposition of the streaming kernels: wavefronts aimed at per CU and launch
  ctx->strip_waves_per_cu = 32;   // measured: lowpass_down 0.247 -> 0.220 ms vs 16 (better balance over the CUs, 64-row segments)
  ctx->scan_waves_per_cu = 32;
  ctx->split_tail = 8;            // batches of >= 8 frames: measured +2.8 % frames/s (0 disables, N sets the smallest batch;
                                  // a single frame is launch-latency bound and loses)
  {
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, hi));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  ctx->bin_detections = 1;
  ctx->descr_occ = 4;
  ctx->tile_descr = 1;
  ctx->tile_orient = 0;
  ctx->orient_blocks_per_cu = 6;          // one round of resident workgroups: orient_all runs 6 waves/SIMD (r06; 5 before)
  ctx->point_blocks_per_cu = 8;
  ctx->chain_max_frames = 4;
  ctx->chain_embed = 1;
  ctx->chain_wait_ticks = 10000000u;          // 100 ms of the 100 MHz wall clock: ~4 orders of magnitude above the chain's own time
  ctx->bin_min_frames = 4;
  ctx->small_frames = 4;
  ctx->balance_frames = 1;       // r05: on by default (full GPU suite + bench A/B both ways: profiles/r05_balance_*); 0 restores per-frame grids
  ctx->fold_descr_tail = 1;
  ctx->fuse_orient = 0;                       // r06: a single call's orientations + descriptors as ONE launch — built, parity-green,
                                              // and 12-30 us SLOWER than the two launches: off
  ctx->fuse_wait_ticks = 10000000u;           // 100 ms, like the chain's
  ctx->patch_reach = 17.9f;                   // = PATCH_REACH of kernels_points.hip: what the LDS window of descr_all covers
  ctx->refcap_limit = 32;                     // MEMWID of FindPointsMultiNew (tests lower it so that natural frames reach it)
  ctx->lowpass_tile = 1;
  ctx->strip_rows_small = 6;
  // (r04 sweep, profiles/r04_single_call_sweep_step5.txt: fine / coarse rows 9/9 31.0 us, 9/4 25.2, 6/4 23.2, 4/3 21.8, 4/2 21.2, 3/3 22.5)
  ctx->scan_rows_small_coarse = 2;
  ctx->scan_rows_small = 4;
  // 0 = a synchronous call returns after hipStreamSynchronize (everything the call wrote is complete and visible to any
  // stream, device or host copy: the reference's contract).  1 = it returns as soon as the last kernel's flag reaches
  // pinned host memory (opt-in: misift_ctx_set_early_return; the cudaSift.h shim does, its read-back is same-stream).
  ctx->host_spin = 0;
  if (tunables_from_env()) {
    for (const KnobName &k : KNOBS)
      if (const char *e = getenv(k.env)) apply_knob(ctx, k.name, atof(e));
  }
  HIP_TRY(hipEventCreate(&ctx->ev0));
  HIP_TRY(hipEventCreate(&ctx->ev1));
  HIP_TRY(misift_dev_alloc((void **)&ctx->d_flags, sizeof(unsigned) * 64, "flags"));
  HIP_TRY(hipMemsetAsync(ctx->d_flags, 0, sizeof(unsigned) * 64, ctx->stream));
  HIP_TRY(hipHostMalloc((void **)&ctx->h_flags, sizeof(unsigned) * 64, hipHostMallocDefault));
  memset(ctx->h_flags, 0, sizeof(unsigned) * 64);
  int rc = misift_ensure_frames(ctx, 1, 65536);
  if (rc) return rc;
  return launch_selftest(ctx);
}

static int ctx_create_physical(int device, void *stream, bool own_stream, misift_ctx **out)
{
  HIP_TRY(hipSetDevice(device));
  CtxFull *f = new CtxFull();
  misift_ctx *ctx = &f->c;
  memset(ctx, 0, sizeof(*ctx));
  if (own_stream) {
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      misift_set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
      delete f;
      return MISIFT_EHIP;
    }
    f->x.own_stream = s;
    stream = s;
  }
  const int rc = ctx_init(ctx, f, device, stream);
  if (rc) {
    misift_ctx_destroy(ctx);                            // keeps the error message of the failing step
    return rc;
  }
  *out = ctx;
  return MISIFT_OK;
}

extern "C" int misift_ctx_set_batches_in_flight(misift_ctx *ctx, int k);

extern "C" int misift_ctx_create(int device, void *stream, misift_ctx **out)
{
  ARG_CHECK(out != nullptr);
  *out = nullptr;
  roctx_bind();
  int n = misift_device_count();
  if (n <= 0) {
    misift_set_error("no HIP device visible");
    return MISIFT_ENODEV;
  }
  if (device < 0 || device >= n) device = n - 1;      // like InitCuda: clamp (cudaSiftH.cu:27)
  device = physical_device(device);                   // MISIFT_DEVICES: from here on the HIP device number
  int rc = ctx_create_physical(device, stream, false, out);
  if (rc) return rc;
  if (const char *e = getenv("MISIFT_BATCHES_IN_FLIGHT")) {
    const int k = atoi(e);
    if (k < 1 || k > 8) {                       // a bad value must not make context creation fail
      if (!getenv("MISIFT_QUIET")) fprintf(stderr, "misift: warning: MISIFT_BATCHES_IN_FLIGHT=%s ignored (1..8)\n", e);
    } else {
      rc = misift_ctx_set_batches_in_flight(*out, k);
      if (rc) { misift_ctx_destroy(*out); *out = nullptr; }
    }
  }
  return rc;
}

// ---- batches in flight: an in-context ring of pipelines
// One misift_ctx is ONE in-order pipeline: batch k+1 starts when batch k is done, so the HBM-bound front end of a batch
// (lowpass_down, 0.24 ms at 0.7 of the HBM peak) never runs beside the VALU-bound kernels of another and every launch
// tail leaves CUs idle.  With K > 1 the context owns K child pipelines (own stream, counters, candidate lists, detection
// staging each — the caller rotates >= K scratch arenas and output buffers) and misift_extract_batch_packed_async hands
// consecutive calls to consecutive pipelines: r02 measured +8-9 % frames/s for K = 3-4 with separate contexts
// (bench.py --contexts); this is the same thing behind ONE context.  The caller's stream only carries a marker per
// call (the batch starts behind everything enqueued on it before the call); completion is observed through
// misift_ctx_wait_batch / misift_gather_post / misift_ctx_sync — NOT through the caller's stream.
extern "C" int misift_ctx_set_batches_in_flight(misift_ctx *ctx, int k)
{
  ARG_CHECK(ctx != nullptr && k >= 1 && k <= 8);
  CtxExtra *x = extra(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  // drain and drop the current ring first
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (misift_ctx *l : x->lanes) { hipStreamSynchronize(l->stream); misift_ctx_destroy(l); }
  x->lanes.clear();
  for (hipEvent_t e : x->lane_done) hipEventDestroy(e);
  x->lane_done.clear();
  x->last_lane = nullptr; x->last_done = nullptr; x->ticket = 0;
  if (k == 1) return MISIFT_OK;
  // build the whole ring or none of it: a half-built ring must never be visible to misift_extract_batch_packed_async
  std::vector<misift_ctx *> lanes;
  std::vector<hipEvent_t> done;
  int rc = MISIFT_OK;
  if (!x->ev_in && hipEventCreateWithFlags(&x->ev_in, hipEventDisableTiming) != hipSuccess) rc = MISIFT_EHIP;
  for (int i = 0; i < k && !rc; i++) {
    misift_ctx *l = nullptr;
    rc = ctx_create_physical(ctx->device, nullptr, true, &l);
    if (!rc) {
      lanes.push_back(l);
      for (const auto &kv : x->knobs) apply_knob(l, kv.first.c_str(), kv.second);
    }
  }
  for (int i = 0; i < 2 * k && !rc; i++) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      misift_set_error("hipEventCreateWithFlags failed while building %d pipelines", k);
      rc = MISIFT_EHIP;
    } else {
      done.push_back(e);
    }
  }
  if (rc) {
    for (misift_ctx *l : lanes) misift_ctx_destroy(l);
    for (hipEvent_t e : done) hipEventDestroy(e);
    return rc;                                        // the context stays a plain in-order one
  }
  x->lanes.swap(lanes);
  x->lane_done.swap(done);
  return MISIFT_OK;
}

extern "C" int misift_test_set_knob(misift_ctx *ctx, const char *name, double value)
{
  ARG_CHECK(ctx != nullptr && name != nullptr);
  int rc = apply_knob(ctx, name, value);
  if (rc) return rc;
  CtxExtra *x = extra(ctx);
  x->knobs.emplace_back(name, value);
  for (misift_ctx *l : x->lanes) apply_knob(l, name, value);
  return MISIFT_OK;
}

extern "C" int misift_ctx_set_early_return(misift_ctx *ctx, int on)
{
  ARG_CHECK(ctx != nullptr);
  if (!(tunables_from_env() && getenv("MISIFT_HOST_SPIN"))) ctx->host_spin = on ? 1 : 0;     // A/B runs: the environment has the last word
  return MISIFT_OK;
}

extern "C" int misift_ctx_last_call_balanced(misift_ctx *ctx) { return ctx ? ctx->cur_balanced : -1; }

extern "C" int misift_ctx_descr_big_fallbacks(misift_ctx *ctx) { return ctx ? ctx->descr_big_fallbacks : -1; }

extern "C" int misift_ctx_chain_fallbacks(misift_ctx *ctx) { return ctx ? ctx->chain_fallbacks : -1; }
extern "C" int misift_ctx_fuse_fallbacks(misift_ctx *ctx) { return ctx ? ctx->fuse_fallbacks : -1; }

extern "C" int misift_ctx_get_batches_in_flight(misift_ctx *ctx)
{
  return ctx ? (extra(ctx)->lanes.empty() ? 1 : (int)extra(ctx)->lanes.size()) : 0;
}

// The pipeline that holds the results (counters, profile) of the most recent batch, and the stream it ran on.
static misift_ctx *result_ctx(misift_ctx *ctx)
{
  CtxExtra *x = extra(ctx);
  return x->last_lane ? x->last_lane : ctx;
}
hipStream_t misift_ctx_result_stream(misift_ctx *ctx) { return result_ctx(ctx)->stream; }

// Record a caller-owned hipEvent_t behind the most recently enqueued batch of `ctx` (on the stream that batch runs on).
// Unlike misift_ctx_wait_batch this adds no waiting stream: on a GPU whose hardware queues are oversubscribed a waiting
// side stream can share a queue with a pipeline and hold it up.
extern "C" int misift_ctx_record_batch(misift_ctx *ctx, void *hip_event)
{
  ARG_CHECK(ctx != nullptr && hip_event != nullptr);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventRecord((hipEvent_t)hip_event, result_ctx(ctx)->stream));
  return MISIFT_OK;
}

// `stream` waits for the most recently enqueued batch of `ctx` (any K; with K = 1 the same as an event recorded on the
// context stream now).
extern "C" int misift_ctx_wait_batch(misift_ctx *ctx, void *stream)
{
  ARG_CHECK(ctx != nullptr);
  CtxExtra *x = extra(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  if (x->last_done) {
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, x->last_done, 0));
    return MISIFT_OK;
  }
  if ((hipStream_t)stream == ctx->stream) return MISIFT_OK;
  if (!x->ev_single) HIP_TRY(hipEventCreateWithFlags(&x->ev_single, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(x->ev_single, ctx->stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, x->ev_single, 0));
  return MISIFT_OK;
}

extern "C" void misift_ctx_destroy(misift_ctx *ctx)
{
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  CtxExtra *x = extra(ctx);
  for (misift_ctx *l : x->lanes) misift_ctx_destroy(l);
  x->lanes.clear();
  for (hipEvent_t e : x->lane_done) hipEventDestroy(e);
  if (x->ev_in) hipEventDestroy(x->ev_in);
  if (x->ev_single) hipEventDestroy(x->ev_single);
  if (x->gexec) hipGraphExecDestroy(x->gexec);
  if (x->gstream) { hipStreamSynchronize(x->gstream); hipStreamDestroy(x->gstream); }
  if (x->gev_in) hipEventDestroy(x->gev_in);
  if (x->gev_out) hipEventDestroy(x->gev_out);
  for (auto &p : x->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  for (auto e : x->pool) hipEventDestroy(e);
  if (ctx->d_counters) misift_dev_free(ctx->d_counters);
  if (ctx->h_counters) hipHostFree(ctx->h_counters);
  if (ctx->d_flags) misift_dev_free(ctx->d_flags);
  if (ctx->h_flags) hipHostFree(ctx->h_flags);
  if (ctx->d_cand) misift_dev_free(ctx->d_cand);
  if (ctx->d_det) misift_dev_free(ctx->d_det);
  if (ctx->d_det_sorted) misift_dev_free(ctx->d_det_sorted);
  if (ctx->d_block_map) misift_dev_free(ctx->d_block_map);
  if (ctx->d_own_scratch) misift_dev_free(ctx->d_own_scratch);
  if (ctx->d_match_tmp) misift_dev_free(ctx->d_match_tmp);
  if (ctx->d_refcap) misift_dev_free(ctx->d_refcap);
  if (ctx->d_capcnt) misift_dev_free(ctx->d_capcnt);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->stream2) { hipStreamSynchronize(ctx->stream2); hipStreamDestroy(ctx->stream2); }
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (x->own_stream) hipStreamDestroy(x->own_stream);
  delete reinterpret_cast<CtxFull *>(ctx);
}

extern "C" int misift_ctx_set_graph_replay(misift_ctx *ctx, int on)
{
  ARG_CHECK(ctx != nullptr);
  extra(ctx)->graph_mode = on ? 1 : 0;
  return MISIFT_OK;
}

extern "C" int misift_ctx_set_stream(misift_ctx *ctx, void *stream)
{
  ARG_CHECK(ctx != nullptr);
  ctx->stream = (hipStream_t)stream;
  return MISIFT_OK;
}

extern "C" int misift_ctx_sync(misift_ctx *ctx)
{
  ARG_CHECK(ctx != nullptr);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (misift_ctx *l : extra(ctx)->lanes) HIP_TRY(hipStreamSynchronize(l->stream));
  return MISIFT_OK;
}

#undef misift_set_options
#undef misift_get_options
#define MISIFT_OPTIONS_R04_BYTES (7 * sizeof(int))       // texfrac_bits ... deterministic: the struct of the r04 header
extern "C" int misift_set_options_sized(misift_ctx *ctx, const misift_options *opt, size_t struct_size)
{
  ARG_CHECK(ctx && opt && struct_size >= sizeof(int) && struct_size % sizeof(int) == 0);
  misift_options o = ctx->opt;                                   // fields the caller's header does not have keep their value
  memcpy(&o, opt, struct_size < sizeof(o) ? struct_size : sizeof(o));
  ARG_CHECK(o.texfrac_bits == 8 || o.texfrac_bits == 23 || o.texfrac_bits == 0);
  ctx->opt = o;
  return MISIFT_OK;
}

extern "C" int misift_get_options_sized(misift_ctx *ctx, misift_options *opt, size_t struct_size)
{
  ARG_CHECK(ctx && opt && struct_size >= sizeof(int));
  memcpy(opt, &ctx->opt, struct_size < sizeof(ctx->opt) ? struct_size : sizeof(ctx->opt));
  return MISIFT_OK;
}

// the symbols binaries built against the r04 header call: its seven fields, nothing behind them
extern "C" int misift_set_options(misift_ctx *ctx, const misift_options *opt)
{
  return misift_set_options_sized(ctx, opt, MISIFT_OPTIONS_R04_BYTES);
}

extern "C" int misift_get_options(misift_ctx *ctx, misift_options *opt)
{
  return misift_get_options_sized(ctx, opt, MISIFT_OPTIONS_R04_BYTES);
}

int misift_ensure_frames(misift_ctx *ctx, int nframes, size_t cand_cap)
{
  if (nframes > ctx->cap_frames) {
    if (ctx->d_counters) HIP_TRY(misift_dev_free(ctx->d_counters));
    if (ctx->h_counters) HIP_TRY(hipHostFree(ctx->h_counters));
    ctx->d_counters = nullptr; ctx->h_counters = nullptr;
    // (CNT_SPARE_BLOCKS more than frames: the call's flags and ticket words live behind the last frame's counters)
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_counters, sizeof(unsigned) * CNT_STRIDE * ((size_t)nframes + CNT_SPARE_BLOCKS), "counters"));
    ctx->alloc_gen++;
    // one block more than frames: the word behind the last frame's counters is the host-export flag (descr_big_kernel)
    HIP_TRY(hipHostMalloc((void **)&ctx->h_counters, sizeof(unsigned) * CNT_STRIDE * ((size_t)nframes + 1), hipHostMallocDefault));
    HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned) * CNT_STRIDE * ((size_t)nframes + CNT_SPARE_BLOCKS), ctx->stream));
    memset(ctx->h_counters, 0, sizeof(unsigned) * CNT_STRIDE * ((size_t)nframes + 1));
  }
  if (nframes > ctx->cap_frames || cand_cap > ctx->cand_cap) {
    const int nf = nframes > ctx->cap_frames ? nframes : ctx->cap_frames;
    const size_t cc = cand_cap > ctx->cand_cap ? cand_cap : ctx->cand_cap;
    if (ctx->d_cand) HIP_TRY(misift_dev_free(ctx->d_cand));
    ctx->d_cand = nullptr;
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_cand, sizeof(unsigned) * cc * nf, "candidates"));
    ctx->alloc_gen++;
    ctx->cand_cap = cc;
    ctx->cap_frames = nf;
  }
  return MISIFT_OK;
}

// grow-only temporary device buffer shared by the matcher (chunk partials) and the homography search
int misift_ensure_tmp(misift_ctx *ctx, size_t bytes)
{
  if (bytes > ctx->match_tmp_bytes) {
    if (ctx->d_match_tmp) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      HIP_TRY(misift_dev_free(ctx->d_match_tmp));
    }
    ctx->d_match_tmp = nullptr; ctx->match_tmp_bytes = 0;
    HIP_TRY(misift_dev_alloc(&ctx->d_match_tmp, bytes, "match_tmp"));
    ctx->match_tmp_bytes = bytes;
  }
  return MISIFT_OK;
}

static int ensure_det(misift_ctx *ctx, int nframes, int max_pts)
{
  if (nframes > ctx->cap_det_frames || max_pts > ctx->det_max_pts) {
    const int nf = nframes > ctx->cap_det_frames ? nframes : ctx->cap_det_frames;
    const int mp = max_pts > ctx->det_max_pts ? max_pts : ctx->det_max_pts;
    if (ctx->d_det) HIP_TRY(misift_dev_free(ctx->d_det));
    if (ctx->d_det_sorted) HIP_TRY(misift_dev_free(ctx->d_det_sorted));
    ctx->d_det = nullptr; ctx->d_det_sorted = nullptr; ctx->cap_det_frames = 0; ctx->det_max_pts = 0;
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_det, sizeof(Detection) * (size_t)nf * MISIFT_MAX_OCTAVES * mp, "detections"));
    HIP_TRY(misift_dev_alloc((void **)&ctx->d_det_sorted, sizeof(Detection) * (size_t)nf * MISIFT_MAX_OCTAVES * mp, "detections_binned"));
    ctx->alloc_gen++;
    ctx->cap_det_frames = nf;
    ctx->det_max_pts = mp;
  }
  return MISIFT_OK;
}

// ------------------------------------------------------------------ memory
// ---- guard mode (test infrastructure inside the product library; inert unless switched on)
#define GUARD_BAND_BYTE 0xA5
#define GUARD_POISON_BYTE 0xFF
struct GuardRec { char *base; size_t bytes; char tag[24]; };
static std::mutex g_guard_mu;
static std::map<void *, GuardRec> g_guard_map;
static int g_guard_mode = -1;                  // -1: not decided yet (MISIFT_GUARD is read at the first allocation)
static bool guard_on()
{
  if (g_guard_mode < 0) {
    const char *e = getenv("MISIFT_GUARD");
    g_guard_mode = (e && atoi(e) != 0) ? 1 : 0;
  }
  return g_guard_mode == 1;
}
hipError_t misift_dev_alloc(void **out, size_t bytes, const char *tag)
{
  if (!guard_on()) return hipMalloc(out, bytes);
  char *base = nullptr;
  const size_t padded = (bytes + 255) / 256 * 256;          // the rear band starts 256-byte aligned behind the payload's end
  hipError_t e = hipMalloc((void **)&base, padded + 2 * (size_t)MISIFT_GUARD_BYTES);
  if (e != hipSuccess) return e;
  if ((e = hipMemset(base, GUARD_BAND_BYTE, padded + 2 * (size_t)MISIFT_GUARD_BYTES)) != hipSuccess) return e;
  if ((e = hipMemset(base + MISIFT_GUARD_BYTES, GUARD_POISON_BYTE, bytes)) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  GuardRec r;
  r.base = base; r.bytes = bytes;
  snprintf(r.tag, sizeof(r.tag), "%s", tag ? tag : "?");
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guard_map[base + MISIFT_GUARD_BYTES] = r;
  *out = base + MISIFT_GUARD_BYTES;
  return hipSuccess;
}
// bands of one guarded allocation: 0 = intact, 1 = damaged (the first damage of a process is described in g_guard_first), -1 = copy failed
static int g_guard_freed_damaged = 0, g_guard_freed_checked = 0;
static char g_guard_first[256];
static int guard_check_one(const GuardRec &r)
{
  const size_t padded = (r.bytes + 255) / 256 * 256;
  const size_t rear = padded - r.bytes + MISIFT_GUARD_BYTES;
  std::vector<unsigned char> host(MISIFT_GUARD_BYTES + rear);
  if (hipMemcpy(host.data(), r.base, MISIFT_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(host.data() + MISIFT_GUARD_BYTES, r.base + MISIFT_GUARD_BYTES + r.bytes, rear, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  for (size_t i = 0; i < host.size(); i++)
    if (host[i] != GUARD_BAND_BYTE) {
      if (!g_guard_first[0]) {
        if (i < MISIFT_GUARD_BYTES)
          snprintf(g_guard_first, sizeof(g_guard_first), "guard band damaged: allocation '%s' (%zu bytes), %zu bytes IN FRONT of the payload",
                   r.tag, r.bytes, (size_t)MISIFT_GUARD_BYTES - i);
        else
          snprintf(g_guard_first, sizeof(g_guard_first), "guard band damaged: allocation '%s' (%zu bytes), %zu bytes BEHIND the payload's end",
                   r.tag, r.bytes, i - MISIFT_GUARD_BYTES);
      }
      return 1;
    }
  return 0;
}
hipError_t misift_dev_free(void *ptr)
{
  if (!ptr) return hipSuccess;
  {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    auto it = g_guard_map.find(ptr);
    if (it != g_guard_map.end()) {
      // a guarded buffer is verified when it goes, too: the caller's images / scratch arenas / record arrays usually live
      // for one call only and are gone by the time anyone asks misift_test_check_guards
      (void)hipDeviceSynchronize();
      g_guard_freed_checked++;
      if (guard_check_one(it->second) != 0) g_guard_freed_damaged++;
      char *base = it->second.base;
      g_guard_map.erase(it);
      return hipFree(base);
    }
  }
  return hipFree(ptr);
}
// Switch guard mode on / off for the allocations made FROM NOW ON (live allocations keep what they have).  Returns the old mode.
extern "C" int misift_test_set_guard(int on)
{
  const int old = guard_on() ? 1 : 0;
  g_guard_mode = on ? 1 : 0;
  return old;
}
// Verify the bands of every live guarded allocation (the whole device is synchronised first).  Returns the number of damaged
// allocations (misift_last_error() names the first: tag, size, band, offset of the first wrong byte), or a negative status;
// *allocations (optional) = live guarded allocations checked.  The payload between tail and rear band (the padding to 256
// bytes) counts as band.
extern "C" int misift_test_check_guards(int *allocations)
{
  if (hipDeviceSynchronize() != hipSuccess) { misift_set_error("misift_test_check_guards: hipDeviceSynchronize failed"); return -MISIFT_EHIP; }
  std::lock_guard<std::mutex> lk(g_guard_mu);
  int bad = g_guard_freed_damaged, n = g_guard_freed_checked;       // allocations freed since the last check were verified as they went
  g_guard_freed_damaged = 0; g_guard_freed_checked = 0;
  for (auto &kv : g_guard_map) {
    const int r = guard_check_one(kv.second);
    if (r < 0) { misift_set_error("misift_test_check_guards: copy of the bands of '%s' failed", kv.second.tag); return -MISIFT_EHIP; }
    n++;
    bad += r;
  }
  if (bad) misift_set_error("%s", g_guard_first);
  g_guard_first[0] = 0;
  if (allocations) *allocations = n;
  return bad;
}

extern "C" int misift_malloc(size_t bytes, void **out)
{
  ARG_CHECK(out != nullptr);
  *out = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = misift_dev_alloc(out, bytes, "misift_malloc");
  if (e != hipSuccess) {
    misift_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MISIFT_ENOMEM;
  }
  return MISIFT_OK;
}

extern "C" int misift_malloc_managed(size_t bytes, void **out)
{
  ARG_CHECK(out != nullptr);
  *out = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMallocManaged(out, bytes, hipMemAttachGlobal);
  if (e != hipSuccess) {
    misift_set_error("hipMallocManaged(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MISIFT_ENOMEM;
  }
  return MISIFT_OK;
}

extern "C" int misift_free(void *ptr)
{
  if (ptr) HIP_TRY(misift_dev_free(ptr));
  return MISIFT_OK;
}

extern "C" int misift_memset(misift_ctx *ctx, void *ptr, int value, size_t bytes)
{
  ARG_CHECK(ctx && ptr);
  HIP_TRY(hipMemsetAsync(ptr, value, bytes, ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_copy_h2d(misift_ctx *ctx, void *dst, const void *src, size_t bytes)
{
  ARG_CHECK(ctx && dst && src);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_copy_d2h(misift_ctx *ctx, void *dst, const void *src, size_t bytes)
{
  ARG_CHECK(ctx && dst && src);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

static inline int ialign_up(int a, int b) { return (a % b != 0) ? (a - a % b + b) : a; }

extern "C" int misift_image_alloc(int width, int height, float **d_out, int *pitch_floats)
{
  ARG_CHECK(width > 0 && height > 0 && d_out && pitch_floats);
  const int p = ialign_up(width, 128);
  void *ptr = nullptr;
  int rc = misift_malloc(sizeof(float) * (size_t)p * height, &ptr);
  if (rc) return rc;
  *d_out = (float *)ptr;
  *pitch_floats = p;
  return MISIFT_OK;
}

extern "C" int misift_upload_2d(misift_ctx *ctx, float *d_dst, int dpitch, const float *h_src, int hpitch, int width,
                                int height)
{
  ARG_CHECK(ctx && d_dst && h_src && width > 0 && height > 0);
  HIP_TRY(hipMemcpy2DAsync(d_dst, sizeof(float) * (size_t)dpitch, h_src, sizeof(float) * (size_t)hpitch,
                           sizeof(float) * (size_t)width, height, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_download_2d(misift_ctx *ctx, float *h_dst, int hpitch, const float *d_src, int dpitch, int width,
                                  int height)
{
  ARG_CHECK(ctx && h_dst && d_src && width > 0 && height > 0);
  HIP_TRY(hipMemcpy2DAsync(h_dst, sizeof(float) * (size_t)hpitch, d_src, sizeof(float) * (size_t)dpitch,
                           sizeof(float) * (size_t)width, height, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_download_fields(misift_ctx *ctx, void *h_pts, const void *d_pts, int npts, int offset, int nfields)
{
  ARG_CHECK(ctx && h_pts && d_pts && npts >= 0 && offset >= 0 && nfields > 0 &&
            offset + 4 * nfields <= MISIFT_POINT_BYTES);
  if (npts == 0) return MISIFT_OK;
  HIP_TRY(hipMemcpy2DAsync((char *)h_pts + offset, MISIFT_POINT_BYTES, (const char *)d_pts + offset,
                           MISIFT_POINT_BYTES, 4 * (size_t)nfields, npts, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

// Reference sizing (cudaSiftH.cu:39-57): sums numOctaves+1 levels, kept for compatibility.
static void scratch_sizes(int width, int height, int num_octaves, int scale_up, size_t *size_img, size_t *size_tmp)
{
  const int nd = NUM_SCALES + 3;
  int w = width * (scale_up ? 2 : 1), h = height * (scale_up ? 2 : 1);
  int p = ialign_up(w, 128);
  size_t size = (size_t)h * p, sizeTmp = (size_t)nd * h * p;
  for (int i = 0; i < num_octaves; i++) {
    w /= 2;
    h /= 2;
    const int p2 = ialign_up(w, 128);
    size += (size_t)h * p2;
    sizeTmp += (size_t)nd * h * p2;
  }
  *size_img = size;
  *size_tmp = sizeTmp;
}

extern "C" size_t misift_scratch_floats(int width, int height, int num_octaves, int scale_up)
{
  size_t a, b;
  scratch_sizes(width, height, num_octaves, scale_up, &a, &b);
  size_t total = a + b;
  return (total + 4095) / 4096 * 4096;     // the reference rounds the arena to 4096-float rows (:58)
}

// -------------------------------------------------------------- host taps
// cudaSiftH.cu:408-418
static void lowpass_taps(float scale, float k[9])
{
  float kernelSum = 0.0f;
  const float ivar2 = 1.0f / (2.0f * scale * scale);
  for (int j = -4; j <= 4; j++) {
    k[j + 4] = (float)expf((float)(-(double)j * j * ivar2));
    kernelSum += k[j + 4];
  }
  for (int j = -4; j <= 4; j++) k[j + 4] /= kernelSum;
}

// cudaSiftH.cu:316-323
static void scaledown_taps(float variance, float k[5])
{
  float kernelSum = 0.0f;
  for (int j = 0; j < 5; j++) {
    k[j] = (float)expf((float)(-(double)(j - 2) * (j - 2) / 2.0 / variance));
    kernelSum += k[j];
  }
  for (int j = 0; j < 5; j++) k[j] /= kernelSum;
}

// cudaSiftH.cu:439-458
static void laplace_taps_rec(int numOctaves, float initBlur, float *kernel)
{
  if (numOctaves > 1) {
    const float totInitBlur = sqrtf(initBlur * initBlur + 0.5f * 0.5f) / 2.0f;
    laplace_taps_rec(numOctaves - 1, totInitBlur, kernel);
  }
  float scale = powf(2.0f, -1.0f / NUM_SCALES);
  const float diffScale = powf(2.0f, 1.0f / NUM_SCALES);
  for (int i = 0; i < NUM_SCALES + 3; i++) {
    float kernelSum = 0.0f;
    const float var = scale * scale - initBlur * initBlur;
    for (int j = 0; j <= 4; j++) {
      kernel[numOctaves * 12 * 16 + 16 * i + j] = (float)expf((float)(-(double)j * j / 2.0 / var));
      kernelSum += (j == 0 ? 1 : 2) * kernel[numOctaves * 12 * 16 + 16 * i + j];
    }
    for (int j = 0; j <= 4; j++) kernel[numOctaves * 12 * 16 + 16 * i + j] /= kernelSum;
    scale *= diffScale;
  }
}

extern "C" int misift_laplace_taps(int num_octaves, float *taps)
{
  ARG_CHECK(taps && num_octaves >= 1 && num_octaves <= MISIFT_MAX_OCTAVES);
  memset(taps, 0, sizeof(float) * 8 * 12 * 16);
  laplace_taps_rec(num_octaves, 0.0f, taps);
  return MISIFT_OK;
}

static LaplaceTaps octave_taps(const float *table, int octave)
{
  LaplaceTaps t;
  for (int s = 0; s < NUM_BLURS; s++)
    for (int j = 0; j < 5; j++) t.k[s][j] = table[octave * 12 * 16 + 16 * s + j];
  return t;
}

// ---------------------------------------------------------------- geometry
static StripGeom make_geom(misift_ctx *ctx, int w, int h, int pitch, int nframes, long long frame_stride,
                           int out_w, int out_rows, int out_lanes)
{
  StripGeom g;
  g.width = w; g.height = h; g.pitch = pitch;
  g.nframes = nframes; g.frame_stride = frame_stride;
  const int nquads = (out_w + 3) / 4;
  g.nstrips = (nquads + out_lanes - 1) / out_lanes;
  if (g.nstrips < 1) g.nstrips = 1;
  // aim for ~16 wavefronts per CU across the launch; segments between 8 and 128 rows
  const long long target = (long long)ctx->num_cus * ctx->strip_waves_per_cu;
  long long want = (target + (long long)nframes * g.nstrips - 1) / ((long long)nframes * g.nstrips);
  if (want < 1) want = 1;
  int seg = (int)((out_rows + want - 1) / want);
  if (nframes <= ctx->small_frames) {         // a frame or two: latency-bound, short (even) segments on every SIMD
    seg = (seg + 1) / 2 * 2;
    if (seg < ctx->strip_rows_small) seg = ctx->strip_rows_small;
  } else {
    seg = (seg + 7) / 8 * 8;
    if (seg < 8) seg = 8;
  }
  if (seg > 128) seg = 128;
  g.seg_rows = seg;
  g.nsegs = (out_rows + seg - 1) / seg;
  if (g.nsegs < 1) g.nsegs = 1;
  // bit 1: strip-fastest item order — the 4 wavefronts of a workgroup stream 4 KB contiguous per image row
  // (measured -20 % on lowpass vs segment-fastest: better HBM page locality); bit 0 would disable the XCD remap
  g.noremap = 2;
  return g;
}

// --------------------------------------------------------------- selftest
__global__ void selftest_kernel(float *out)
{
  const int lane = threadIdx.x;
  const float v = (float)(lane + 1);
  out[lane] = lane_from_left(v);          // expect lane      (0 for lane 0)
  out[64 + lane] = lane_from_right(v);    // expect lane + 2  (0 for lane 63)
}

int launch_selftest(misift_ctx *ctx)
{
  float *d = nullptr;
  float h[128];
  HIP_TRY(hipMalloc((void **)&d, sizeof(h)));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipFree(d));
  for (int l = 1; l < 64; l++)
    if (h[l] != (float)l) {
      misift_set_error("selftest: DPP wave_shr:1 lane %d got %g, expected %d", l, h[l], l);
      return MISIFT_EHIP;
    }
  for (int l = 0; l < 63; l++)
    if (h[64 + l] != (float)(l + 2)) {
      misift_set_error("selftest: DPP wave_shl:1 lane %d got %g, expected %d", l, h[64 + l], l + 2);
      return MISIFT_EHIP;
    }
  return MISIFT_OK;
}

// -------------------------------------------------------------- extraction
struct Level { int w, h, p; float *img; };   // img = frame-0 pointer of that pyramid level

// Images under 16 x 16, or whose coarsest pyramid level is under 8 px: the reference runs them (levels shrink to a few
// pixels, every access clamped: cudaSiftH.cu:72-167) and so do we — on the dense per-level kernels, whose quad loads clamp
// at any width >= 1.  The merged-octave kernels (tiled prefilter, cone chain, strips sized for whole wavefronts) are
// built for images that fill at least a strip and never see such calls.  A level that integer division has shrunk to
// 0 pixels holds nothing and is skipped (CUDA itself refuses a zero-sized grid there).
// (options.reference_cap no longer sends a call here, r06: the fused path counts the true extrema of every 30 x 8 block in
//  refine_all and only a frame in which a block reaches a 33rd is redone on the dense kernels — the candidate-overflow
//  mechanism — where launch_refcap applies the cap in the reference's order.)
static bool dense_call(misift_ctx *ctx, int width, int height, int num_octaves)
{
  (void)ctx;
  return misift_tiny_call(width, height, num_octaves);
}
bool misift_tiny_call(int width, int height, int num_octaves)
{
  if (num_octaves < 1) return false;
  return width < 16 || height < 16 || (width >> (num_octaves - 1)) < 8 || (height >> (num_octaves - 1)) < 8;
}

// Enqueue the whole launch sequence of one batch on the context stream (no synchronisation).
// d_imgs: fp32 frames, or 8-bit frames when src_u8 (pitch / frame_stride in source elements).
int misift_extract_enqueue(misift_ctx *ctx, const void *d_imgs, int src_u8, int nframes, long long frame_stride,
                           int width, int height, int pitch, int num_octaves, float init_blur, float thresh,
                           float lowest_scale, int scale_up, float *d_scratch, SiftPointD *pts, int max_pts)
{
  RoctxRange range("misift_extract");
  ARG_CHECK(ctx && d_imgs && (pts || (ctx->pack_dst && ctx->opt.fused)));
  extra(ctx)->last_lane = nullptr;      // this batch runs on the context itself: its counters / stream are the results'
  extra(ctx)->last_done = nullptr;
  ARG_CHECK(nframes >= 1 && width >= 1 && height >= 1 && pitch >= width);
  ARG_CHECK(num_octaves >= 1 && num_octaves <= MISIFT_MAX_OCTAVES);       // before the shifts below
  const bool tiny = misift_tiny_call(width, height, num_octaves);
  // the entry points route tiny calls to the dense kernels (TinyScope)
  ARG_CHECK(!(tiny && ctx->opt.fused));
  ARG_CHECK(max_pts >= 1);
  ARG_CHECK(width * (scale_up ? 2 : 1) < 16384 && height * (scale_up ? 2 : 1) < 16384);
  HIP_TRY(hipSetDevice(ctx->device));
  // candidate list: pre-candidates of the fused scan (or true extrema of the unfused detect) of ONE
  // octave; sized from the image, not from max_pts (overflow is detected and handled by the callers)
  size_t cap = (size_t)width * height * (scale_up ? 4 : 1) / 4;
  if (cap < 65536) cap = 65536;
  int rc = misift_ensure_frames(ctx, nframes, cap);
  if (rc) return rc;
  const size_t S = misift_scratch_floats(width, height, num_octaves, scale_up);
  if (!d_scratch) {
    if (ctx->own_scratch_floats < S * nframes) {
      if (ctx->d_own_scratch) HIP_TRY(misift_dev_free(ctx->d_own_scratch));
      ctx->d_own_scratch = nullptr; ctx->own_scratch_floats = 0;
      HIP_TRY(misift_dev_alloc((void **)&ctx->d_own_scratch, sizeof(float) * S * nframes, "own_scratch"));
      ctx->own_scratch_floats = S * nframes;
    }
    d_scratch = ctx->d_own_scratch;
  }
  // (the frames' counter blocks are cleared by the prefilter kernel — the first kernel of every path below)
  ctx->exported = 0;

  // the tap tables depend on (num_octaves, init_blur) only: a few hundred expf / powf, i.e. microseconds of host time in
  // front of the first launch of EVERY call — cached per context (r04 single-call budget)
  CtxExtra *xt = extra(ctx);
  if (xt->taps_noct != num_octaves) {
    misift_laplace_taps(num_octaves, xt->taps_table);          // cudaSiftH.cu:109-111
    xt->taps_noct = num_octaves;
  }
  const float blur = init_blur > 0.001f ? init_blur : 0.001f;
  if (!(xt->k9_blur == blur)) {
    lowpass_taps(blur, xt->k9);
    xt->k9_blur = blur;
  }
  if (!xt->k5_done) {
    scaledown_taps(0.5f, xt->k5);
    xt->k5_done = true;
  }
  const float *table = xt->taps_table, *k9 = xt->k9, *k5 = xt->k5;

  // arena layout of cudaSiftH.cu:104-107, :151-159, :179-184 (per frame, frame stride S)
  size_t size_img, size_tmp;
  scratch_sizes(width, height, num_octaves, scale_up, &size_img, &size_tmp);
  const int W = width * (scale_up ? 2 : 1), H = height * (scale_up ? 2 : 1);
  float *memoryTmp = d_scratch;
  float *memorySub = d_scratch + size_tmp;
  std::vector<Level> lv(num_octaves + 1);
  {
    int w = W, h = H;
    float *ptr = memorySub;
    for (int o = num_octaves; o >= 1; o--) {
      lv[o].w = w; lv[o].h = h; lv[o].p = ialign_up(w, 128); lv[o].img = ptr;
      ptr += (size_t)h * lv[o].p;
      w /= 2; h /= 2;
    }
  }
  const long long SS = (long long)S;
  // --- prefilter (cudaSiftH.cu:112 / :119-123); when the shape allows, the first ScaleDown of the pyramid is
  // produced by the same kernel (the prefiltered rows are decimated while still in registers)
  int first_down_done = 0;
  {
    const Level &L = lv[num_octaves];
    const void *pre_src = d_imgs;
    int pre_u8 = src_u8, pre_pitch = pitch, pre_frames = nframes;
    long long pre_stride = frame_stride;
    if (scale_up) {
      float *upImg = memoryTmp;
      // every frame's up-sampled copy borrows the DoG region of its own arena (cudaSiftH.cu:119-123)
      rc = launch_scaleup(ctx, d_imgs, src_u8, width, height, pitch, frame_stride, nframes, upImg, L.p, SS);
      if (rc) return rc;
      pre_src = upImg; pre_u8 = 0; pre_pitch = L.p; pre_frames = nframes; pre_stride = SS;
      lowest_scale *= 2.0f;
    }
    if (num_octaves >= 2 && ctx->opt.fused && ctx->lowpass_tile && nframes <= ctx->small_frames) {
      const Level &D = lv[num_octaves - 1];        // a frame or two: the LDS-tiled form (latency, not throughput)
      rc = launch_lowpass_down_tile(ctx, pre_src, pre_u8, W, H, pre_pitch, pre_stride, pre_frames, L.img, L.p, SS, k9, D.img,
                                    D.p, SS, k5, ctx->d_counters);
      if (rc) return rc;
      first_down_done = 1;
    } else if (num_octaves >= 2 && ctx->opt.fused) {
      const Level &D = lv[num_octaves - 1];
      StripGeom g = make_geom(ctx, W, H, pre_pitch, pre_frames, pre_stride, W, H, 60);
      rc = launch_lowpass_down(ctx, pre_src, pre_u8, g, L.img, L.p, SS, k9, D.img, D.p, SS, k5, &first_down_done,
                               ctx->d_counters);
      if (rc) return rc;
    }
    if (!first_down_done) {
      StripGeom g = make_geom(ctx, W, H, pre_pitch, pre_frames, pre_stride, W, H, 62);
      rc = launch_lowpass(ctx, pre_src, pre_u8, g, L.img, L.p, SS, k9, ctx->d_counters);
      if (rc) return rc;
    }
  }
  // --- merged-octave path: its level table is needed before the pyramid is built (split-tail mode below)
  PyramidInfo P;
  memset(&P, 0, sizeof(P));
  std::vector<LaplaceTaps> tapsv(num_octaves + 1);
  if (ctx->opt.fused) {
    P.noct = num_octaves; P.nframes = nframes; P.frame_stride = SS;
    P.fix_numpts = ctx->opt.fix_numpts ? 1 : 0;
    P.patch_reach = ctx->patch_reach;
    P.out_scale = scale_up ? 0.5f : 1.0f;             // RescalePositions (cudaSiftH.cu:130) folded into the record write
    unsigned off = 0;
    for (int o = 1; o <= num_octaves; o++) {
      OctaveInfo &L = P.o[o];
      L.w = lv[o].w; L.h = lv[o].h; L.p = lv[o].p;
      L.img_off = (long long)(lv[o].img - d_scratch);
      L.subsampling = (float)(1 << (num_octaves - o));
      L.lowest_scale = lowest_scale / L.subsampling;
      size_t c = (size_t)L.w * L.h / 4;
      if (c < 16384) c = 16384;
      L.cand_cap = (unsigned)c;
      L.cand_off = off;
      off += L.cand_cap;
      tapsv[o] = octave_taps(table, o);
    }
    rc = misift_ensure_frames(ctx, nframes, off);
    if (rc) return rc;
    // the staging area is indexed with the CURRENT max_pts (kernels use it as the per-octave stride)
    rc = ensure_det(ctx, nframes, max_pts);
    if (rc) return rc;
  }
  // Split-tail mode: the two finest levels exist as soon as the fused prefilter has run, so their scan (most of
  // the scan work) starts at once on the context stream while the small ScaleDowns of the coarse levels and the
  // scan of those levels run beside it on a second, high-priority stream — their launch gaps and tails hide
  // under the big kernel.
  bool scanned = false;
  if (ctx->opt.fused && ctx->split_tail && nframes >= ctx->split_tail && !ctx->in_capture && first_down_done &&
      num_octaves >= 3 && ctx->stream2) {
    HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    rc = launch_dog_scan_all(ctx, d_scratch, P, tapsv.data(), thresh, 0, 2);
    if (rc) return rc;
    hipStream_t saved = ctx->stream;
    ctx->stream = ctx->stream2;
    for (int o = num_octaves - 1; o >= 2 && !rc; o--) {
      const Level &src = lv[o], &dst = lv[o - 1];
      StripGeom g = make_geom(ctx, src.w, src.h, src.p, nframes, SS, dst.w, dst.h, 62);
      rc = launch_scaledown(ctx, src.img, g, dst.img, dst.p, SS, k5);
    }
    if (!rc) rc = launch_dog_scan_all(ctx, d_scratch, P, tapsv.data(), thresh, 2, num_octaves);
    ctx->stream = saved;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    scanned = true;
  }
  // --- pyramid (ScaleDown chain of cudaSiftH.cu:153-160), finest to coarsest
  // Small batches (a single frame above all) are bound by the number of DEPENDENT dispatches, not by their work: up to
  // three levels per launch (scaledown_chain_kernel).  Batches keep one streamed launch per level.
  const bool chained = !scanned && !tiny && nframes <= ctx->chain_max_frames;
  // ... and when ONE chain covers every remaining level, it rides in the first workgroups of the scan launch
  ChainGeom embed;
  bool embedded = false;
  // (only where the host reads the counter blocks afterwards — the synchronous calls: that is where an expired wait is
  //  noticed and the call re-run, see read_counts)
  if (chained && ctx->opt.fused && ctx->chain_embed && ctx->want_export && first_down_done && num_octaves >= 3 &&
      num_octaves - 2 <= 3) {
    const int o = num_octaves - 1, nlev = o - 1;
    int dims[4][3];
    long long offs[4];
    for (int k = 0; k <= nlev; k++) {
      dims[k][0] = lv[o - k].w; dims[k][1] = lv[o - k].h; dims[k][2] = lv[o - k].p;
      offs[k] = (long long)(lv[o - k].img - d_scratch);
    }
    rc = make_chain_geom(&embed, SS, dims, offs, nlev, 4);
    if (rc) return rc;
    embedded = true;
  }
  for (int o = num_octaves; o >= 2 && !scanned && !embedded;) {
    if (o == num_octaves && first_down_done) { o--; continue; }
    if (chained) {
      const int nlev = (o - 1) < 3 ? (o - 1) : 3;
      int dims[4][3];
      long long offs[4];
      for (int k = 0; k <= nlev; k++) {
        dims[k][0] = lv[o - k].w; dims[k][1] = lv[o - k].h; dims[k][2] = lv[o - k].p;
        offs[k] = (long long)(lv[o - k].img - d_scratch);
      }
      rc = launch_scaledown_chain(ctx, d_scratch, SS, nframes, dims, offs, nlev, k5);
      if (rc) return rc;
      o -= nlev;
      continue;
    }
    const Level &src = lv[o], &dst = lv[o - 1];
    if (dst.w < 1 || dst.h < 1) break;                 // nothing left to decimate: this and every coarser level is empty
    StripGeom g = make_geom(ctx, src.w, src.h, src.p, nframes, SS, dst.w, dst.h, 62);
    rc = launch_scaledown(ctx, src.img, g, dst.img, dst.p, SS, k5);
    if (rc) return rc;
    o--;
  }
  if (ctx->opt.fused) {
    // scan / refine / orient / descr each run ONCE over all pyramid levels; the final array is laid out in the
    // reference's segment order by descr_all_kernel
    if (!scanned) {
      rc = launch_dog_scan_all(ctx, d_scratch, P, tapsv.data(), thresh, 0, num_octaves, embedded ? &embed : nullptr, k5);
      if (rc) return rc;
    }
    rc = launch_refine_all(ctx, d_scratch, P, tapsv.data(), thresh, 10.0f, 1.0f / NUM_SCALES, max_pts);
    if (rc) return rc;
    // (a counting sort per (octave, frame) is one workgroup each: for a frame or two it is a dependent dispatch that buys
    //  nothing — the keypoints of one frame share the caches anyway)
    const bool binned = (ctx->bin_detections && nframes >= ctx->bin_min_frames) || ctx->opt.deterministic;
    ctx->cur_binned = binned ? 1 : 0;
    ctx->cur_balanced = 0;              // set by launch_bin_detections (its extra workgroup builds the block tables) or, without binning, by build_block_maps
    if (binned) {                       // spatial order for the per-keypoint kernels (L1/L2 reuse between neighbours);
      rc = launch_bin_detections(ctx, P, max_pts);      // deterministic mode: a total order
      if (rc) return rc;
    }
    // a frame or two: orientations and descriptors in one launch (the descriptor pass of a workgroup waits in the kernel for
    // the coarser octaves' orientations, whose duplicate counts decide where its records go)
    // (off by default: measured 12 us slower than the two launches)
    if (ctx->fuse_orient && !binned && !ctx->pack_dst && nframes <= ctx->small_frames && ctx->tile_descr && !ctx->tile_orient &&
        !ctx->in_capture)
      return launch_orient_descr_fused(ctx, d_scratch, P, pts, max_pts);
    rc = launch_orient_all(ctx, d_scratch, P, pts, max_pts);
    if (rc) return rc;
    if (ctx->opt.deterministic) {       // second-orientation slots in keypoint order instead of atomic order
      rc = launch_renumber_dups(ctx, P, max_pts);
      if (rc) return rc;
    }
    if (ctx->pack_dst) {                // counts and offsets are known as soon as the orientations are: pack while writing
      rc = launch_export_counts_staged(ctx, nframes, num_octaves, max_pts, ctx->pack_counts, ctx->pack_offsets);
      if (rc) return rc;
    }
    return launch_descr_all(ctx, d_scratch, P, pts, max_pts, ctx->pack_offsets, ctx->pack_dst);
  }
  // --- octaves, coarsest first (cudaSiftH.cu:161 after the recursion)
  for (int o = 1; o <= num_octaves; o++) {
    const Level &L = lv[o];
    if (L.w < 1 || L.h < 1) continue;                  // an empty level: no pixels, no points (counters stay where they are)
    const float subsampling = (float)(1 << (num_octaves - o));
    const LaplaceTaps taps = octave_taps(table, o);
    {
      StripGeom g = make_geom(ctx, L.w, L.h, L.p, nframes, SS, L.w, L.h, 62);
      rc = launch_laplace(ctx, L.img, g, memoryTmp, SS, taps);
      if (rc) return rc;
      rc = launch_detect(ctx, memoryTmp, g, SS, thresh, o);
      if (rc) return rc;
      if (ctx->opt.reference_cap) {
        rc = launch_refcap(ctx, L.w, L.h, nframes, o);
        if (rc) return rc;
      }
      rc = launch_refine(ctx, memoryTmp, SS, nullptr, 0, nullptr, L.w, L.h, L.p, nframes, thresh, 10.0f,
                         1.0f / NUM_SCALES, lowest_scale / subsampling, subsampling, o, pts, max_pts);
      if (rc) return rc;
    }
    rc = launch_orient(ctx, L.img, SS, L.w, L.h, L.p, nframes, o, pts, max_pts);
    if (rc) return rc;
    rc = launch_descr(ctx, L.img, SS, L.w, L.h, L.p, nframes, subsampling, o, pts, max_pts);
    if (rc) return rc;
  }
  if (ctx->opt.deterministic) {                     // dense path: order fixed after the fact, segment by segment
    rc = launch_sort_segments(ctx, pts, max_pts, nframes, num_octaves);
    if (rc) return rc;
  }
  if (scale_up) return launch_rescale_batch(ctx, pts, max_pts, nframes, num_octaves, 0.5f);      // cudaSiftH.cu:130
  return MISIFT_OK;
}

// Wait until a kernel of the context stream has stored `seq` into a word of pinned host memory: poll it (a few hundred
// ns after the store) instead of a blocking copy + hipStreamSynchronize.  The records themselves are complete for
// every later operation on the context's stream; a consumer on ANOTHER stream or device calls misift_ctx_sync first
// (include/misift.h).  MISIFT_HOST_SPIN=0: plain hipStreamSynchronize.
static int wait_host_flag(misift_ctx *ctx, unsigned *word, unsigned seq)
{
  volatile unsigned *flag = word;
  bool seen = false;
  if (ctx->host_spin) {
    for (unsigned spins = 0; !seen; spins++) {
      seen = *flag == seq;
#if defined(__x86_64__) || defined(__i386__)
      if (!seen) __builtin_ia32_pause();                   // a polite spin: the sibling hyper-thread keeps its issue slots
#endif
      if (!seen && (spins & 0x3fff) == 0x3fff) {           // every ~16 k polls: is the stream still alive?
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) { seen = *flag == seq; break; }
        if (q != hipErrorNotReady) HIP_TRY(q);
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // keep the runtime's bookkeeping of finished commands bounded
    if ((seq & 63u) == 0) HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (!seen) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (*flag != seq) {
      misift_set_error("the kernel's completion flag did not arrive (flag %u, expected %u)", *flag, seq);
      return MISIFT_EHIP;
    }
  }
  return MISIFT_OK;
}

#define MISIFT_RETRY_CHAIN 1000      // internal to this file: never returned to a caller
#define MISIFT_RETRY_FUSE 1001
static int read_counts(misift_ctx *ctx, int nframes, int num_octaves, int max_pts, int *num_pts_out,
                       bool *cand_overflow)
{
  if (ctx->exported) {
    // the last kernel of the call has written the counter blocks into h_counters and stores export_seq behind them
    int rc = wait_host_flag(ctx, ctx->h_counters + (size_t)nframes * CNT_STRIDE, ctx->export_seq);
    if (rc) return rc;
    if (ctx->pending_big.valid) {
      // a folded single call (launch_descr_all): descr_all exported the counters; descr_big runs only if some keypoint was
      // deferred to it — next to never — and exports them again
      bool any = false;
      for (int f = 0; f < nframes; f++) any = any || ctx->h_counters[(size_t)f * CNT_STRIDE + CNT_BIG] != 0;
      if (!any) ctx->pending_big.valid = 0;
      else {
        rc = launch_descr_big_pending(ctx);
        if (!rc) rc = wait_host_flag(ctx, ctx->h_counters + (size_t)nframes * CNT_STRIDE, ctx->export_seq);
        if (rc) return rc;
      }
    }
  } else {
    HIP_TRY(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(unsigned) * CNT_STRIDE * (size_t)nframes,
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (ctx->h_counters[CNT_CHAINTMO]) return MISIFT_RETRY_CHAIN;      // dog_scan_all_kernel: the bounded in-launch wait expired
  if (ctx->h_counters[CNT_FUSETMO]) return MISIFT_RETRY_FUSE;        // orient_descr_fused_kernel: likewise
  const int slot = 2 * num_octaves + (ctx->opt.fix_numpts ? 1 : 0);
  for (int f = 0; f < nframes; f++) {
    const unsigned c = ctx->h_counters[(size_t)f * CNT_STRIDE + slot];
    num_pts_out[f] = (int)(c < (unsigned)max_pts ? c : (unsigned)max_pts);     // cudaSiftH.cu:116
    if (ctx->h_counters[(size_t)f * CNT_STRIDE + CNT_CANDOVF]) *cand_overflow = true;
  }
  return MISIFT_OK;
}

// Run the launch sequence and read the counts back.  If a candidate list overflowed in the fused
// path (possible only for extreme thresh/contrast), redo the batch with the dense unfused kernels,
// whose list holds true 3x3x3 extrema only; an overflow there is reported as an error, never dropped
// silently (the reference silently caps at 32 candidates per 30x8 tile, cudaSiftD.cu:1371).
// Enqueue through a captured hipGraph when this exact call was seen before: the first call runs normally (and
// allocates whatever the context needs), the second is captured while it is enqueued, later ones are one
// hipGraphLaunch instead of ~10 launches.  Returns 1 if the work was queued here, 0 if the caller must enqueue it
// the ordinary way (graphs disabled, profiling on, unsuitable call, or any capture problem — which turns graphs off).
static int enqueue_via_graph(misift_ctx *ctx, const CallKey &key, const void *d_imgs, long long frame_stride,
                             float *d_scratch, SiftPointD *pts)
{
  CtxExtra *x = extra(ctx);
  if (!x->graph_mode || ctx->profile || !d_scratch || key.scale_up || !key.fused) return 0;
  if (x->have_graph && x->graph_key == key) {
    if (hipEventRecord(x->gev_in, ctx->stream) != hipSuccess || hipStreamWaitEvent(x->gstream, x->gev_in, 0) != hipSuccess ||
        hipGraphLaunch(x->gexec, x->gstream) != hipSuccess || hipEventRecord(x->gev_out, x->gstream) != hipSuccess ||
        hipStreamWaitEvent(ctx->stream, x->gev_out, 0) != hipSuccess) {
      (void)hipGetLastError();
      x->graph_mode = 0;
      return 0;
    }
    return 1;
  }
  const bool repeat = x->have_last && x->last_key == key;
  x->last_key = key;
  x->have_last = true;
  if (!repeat) return 0;
  // second identical call: capture it
  if (!x->gstream) {
    if (hipStreamCreateWithFlags(&x->gstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&x->gev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->gev_out, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      x->graph_mode = 0;
      return 0;
    }
  }
  if (x->have_graph) { hipGraphExecDestroy(x->gexec); x->gexec = nullptr; x->have_graph = false; }
  hipStream_t saved = ctx->stream;
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(x->gstream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    x->graph_mode = 0;
    return 0;
  }
  ctx->stream = x->gstream;
  ctx->in_capture = 1;
  const int rc = misift_extract_enqueue(ctx, d_imgs, key.src_u8, key.nframes, frame_stride, key.width, key.height,
                                        key.pitch, key.num_octaves, key.init_blur, key.thresh, key.lowest_scale, 0,
                                        d_scratch, pts, key.max_pts);
  ctx->stream = saved;
  ctx->in_capture = 0;
  const hipError_t e = hipStreamEndCapture(x->gstream, &graph);
  if (rc || e != hipSuccess || !graph || hipGraphInstantiate(&x->gexec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (graph) hipGraphDestroy(graph);
    x->gexec = nullptr;
    x->graph_mode = 0;                 // never try again on this context; the caller enqueues the ordinary way
    return 0;
  }
  hipGraphDestroy(graph);
  x->graph_key = key;
  x->have_graph = true;
  return enqueue_via_graph(ctx, key, d_imgs, frame_stride, d_scratch, pts);      // replay what was just captured
}

// The asynchronous entry points run a tiny call on the dense kernels as well (misift_tiny_call).
struct TinyScope {
  misift_ctx *ctx;
  int saved;
  TinyScope(misift_ctx *c, int w, int h, int noct) : ctx(c), saved(c ? c->opt.fused : 0)
  {
    if (ctx && dense_call(ctx, w, h, noct)) ctx->opt.fused = 0;
  }
  ~TinyScope() { if (ctx) ctx->opt.fused = saved; }
};

int misift_extract_sync(misift_ctx *ctx, const void *d_imgs, int src_u8, int nframes, long long frame_stride, int width,
                        int height, int pitch, int num_octaves, float init_blur, float thresh, float lowest_scale,
                        int scale_up, float *d_scratch, SiftPointD *pts, int max_pts, int *num_pts_out)
{
  ARG_CHECK(ctx != nullptr && num_pts_out != nullptr);
  // the option is restored on EVERY way out (incl. the HIP_TRY returns below) by a scope guard, not by hand
  struct FusedGuard {
    misift_ctx *c; int saved;
    ~FusedGuard() { c->opt.fused = saved; }
  } fused_guard{ctx, ctx->opt.fused};
  const int fused_saved = fused_guard.saved;
  if (dense_call(ctx, width, height, num_octaves)) ctx->opt.fused = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = MISIFT_OK;
    int queued = 0;
    if (attempt == 0 && ctx && d_imgs && pts) {
      CallKey key;
      memset(&key, 0, sizeof(key));
      key.imgs = d_imgs; key.scratch = d_scratch; key.pts = pts;
      key.src_u8 = src_u8; key.nframes = nframes; key.width = width; key.height = height; key.pitch = pitch;
      key.num_octaves = num_octaves; key.scale_up = scale_up; key.max_pts = max_pts; key.fused = ctx->opt.fused;
      key.texfrac = ctx->opt.texfrac_bits; key.fixnum = ctx->opt.fix_numpts; key.determ = ctx->opt.deterministic; key.refcap = ctx->opt.reference_cap; key.alloc_gen = ctx->alloc_gen;
      key.frame_stride = frame_stride; key.init_blur = init_blur; key.thresh = thresh; key.lowest_scale = lowest_scale;
      queued = enqueue_via_graph(ctx, key, d_imgs, frame_stride, d_scratch, pts);
    }
    if (!queued) {
      ctx->want_export = 1;              // the last kernel hands the counters to the host (paths that cannot leave it 0)
      rc = misift_extract_enqueue(ctx, d_imgs, src_u8, nframes, frame_stride, width, height, pitch, num_octaves,
                                  init_blur, thresh, lowest_scale, scale_up, d_scratch, pts, max_pts);
      ctx->want_export = 0;
    } else {
      ctx->exported = 0;                 // a replayed graph carries no fresh sequence number
    }
    bool ovf = false;
    if (!rc) rc = read_counts(ctx, nframes, num_octaves, max_pts, num_pts_out, &ovf);
    if (rc == MISIFT_RETRY_CHAIN && ctx->chain_embed) {
      // Some workgroup's bounded wait for the ScaleDown chain inside the scan launch expired: its items were skipped, the
      // records are incomplete.  Never again on this context: the chain becomes a launch of its own (one dependent
      // dispatch, ~5 us per call), and this call is redone that way.
      if (!ctx->opt.quiet)
        fprintf(stderr, "misift: the in-launch wait for the ScaleDown chain expired (workgroups were not dispatched in index "
                        "order?); this context now launches the chain separately\n");
      ctx->chain_embed = 0;
      ctx->chain_fallbacks++;
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      attempt--;
      continue;
    }
    if (rc == MISIFT_RETRY_FUSE && ctx->fuse_orient) {
      // Some workgroup of the fused orientation + descriptor launch gave up waiting for the coarser octaves' orientations and
      // skipped its descriptors.  Never again on this context: two launches, and this call is redone that way.
      if (!ctx->opt.quiet)
        fprintf(stderr, "misift: the in-launch wait of the fused orientation + descriptor kernel expired; this context now "
                        "launches the two kernels separately\n");
      ctx->fuse_orient = 0;
      ctx->fuse_fallbacks++;
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      attempt--;
      continue;
    }
    if (rc == MISIFT_RETRY_CHAIN || rc == MISIFT_RETRY_FUSE) rc = MISIFT_EHIP;
    if (rc) { ctx->opt.fused = fused_saved; return rc; }
    if (!ovf) break;
    if (ctx->opt.fused && attempt == 0) {
      // Exact re-run with the dense kernels — of the frames whose candidate list overflowed ONLY (r02 redid the whole
      // batch at 0.3x the fused rate for one bad frame).  Every run of consecutive overflowed frames is one sub-call on
      // its own slices of the inputs, the scratch arena and the output; its counters are merged into the batch's.
      std::vector<unsigned> merged(ctx->h_counters, ctx->h_counters + (size_t)CNT_STRIDE * nframes);
      const size_t S = misift_scratch_floats(width, height, num_octaves, scale_up);
      const size_t esz = src_u8 ? 1 : sizeof(float);
      ctx->opt.fused = 0;
      bool ovf2 = false;
      for (int f0 = 0; f0 < nframes && !rc;) {
        if (!merged[(size_t)f0 * CNT_STRIDE + CNT_CANDOVF]) { f0++; continue; }
        int f1 = f0 + 1;
        while (f1 < nframes && merged[(size_t)f1 * CNT_STRIDE + CNT_CANDOVF]) f1++;
        const int nr = f1 - f0;
        rc = misift_extract_enqueue(ctx, (const char *)d_imgs + (size_t)f0 * frame_stride * esz, src_u8, nr, frame_stride, width,
                                    height, pitch, num_octaves, init_blur, thresh, lowest_scale, scale_up,
                                    d_scratch ? d_scratch + (size_t)f0 * S : nullptr, pts + (size_t)f0 * max_pts, max_pts);
        if (!rc) rc = read_counts(ctx, nr, num_octaves, max_pts, num_pts_out + f0, &ovf2);
        if (!rc) memcpy(&merged[(size_t)f0 * CNT_STRIDE], ctx->h_counters, sizeof(unsigned) * CNT_STRIDE * (size_t)nr);
        f0 = f1;
      }
      ctx->opt.fused = fused_saved;
      if (rc) return rc;
      if (ovf2) {
        misift_set_error("candidate list overflow: more than %zu scale-space extrema in one octave of one frame "
                         "(raise thresh)", ctx->cand_cap);
        return MISIFT_ENOMEM;
      }
      // the batch's counters (misift_get_counters) are the merged ones
      memcpy(ctx->h_counters, merged.data(), sizeof(unsigned) * merged.size());
      HIP_TRY(hipMemcpyAsync(ctx->d_counters, ctx->h_counters, sizeof(unsigned) * merged.size(), hipMemcpyHostToDevice,
                             ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      return MISIFT_OK;
    }
    ctx->opt.fused = fused_saved;
    misift_set_error("candidate list overflow: more than %zu scale-space extrema in one octave of one frame "
                     "(raise thresh)", ctx->cand_cap);
    return MISIFT_ENOMEM;
  }
  ctx->opt.fused = fused_saved;
  return MISIFT_OK;
}

extern "C" int misift_extract(misift_ctx *ctx, const float *d_img, int width, int height, int pitch, int num_octaves,
                              float init_blur, float thresh, float lowest_scale, int scale_up, float *d_scratch,
                              void *d_pts, int max_pts, int *num_pts_out)
{
  ARG_CHECK(num_pts_out != nullptr);
  int rc = misift_extract_sync(ctx, d_img, 0, 1, 0, width, height, pitch, num_octaves, init_blur, thresh, lowest_scale,
                        scale_up, d_scratch, (SiftPointD *)d_pts, max_pts, num_pts_out);
  if (rc) return rc;
  return resolve_profile(ctx);                       // RescalePositions (cudaSiftH.cu:130) is part of the launch sequence
}

extern "C" int misift_extract_batch(misift_ctx *ctx, const float *d_imgs, int nframes, size_t frame_stride, int width,
                                    int height, int pitch, int num_octaves, float init_blur, float thresh,
                                    float lowest_scale, float *d_scratch, void *d_pts, int max_pts,
                                    int *num_pts_out)
{
  ARG_CHECK(num_pts_out != nullptr);
  int rc = misift_extract_sync(ctx, d_imgs, 0, nframes, (long long)frame_stride, width, height, pitch, num_octaves,
                        init_blur, thresh, lowest_scale, 0, d_scratch, (SiftPointD *)d_pts, max_pts, num_pts_out);
  if (rc) return rc;
  return resolve_profile(ctx);
}

extern "C" int misift_extract_batch_ex(misift_ctx *ctx, const void *d_imgs, int src_u8, int nframes, size_t frame_stride,
                                       int width, int height, int pitch, int num_octaves, float init_blur, float thresh,
                                       float lowest_scale, int scale_up, float *d_scratch, void *d_pts, int max_pts,
                                       int *num_pts_out)
{
  ARG_CHECK(num_pts_out != nullptr);
  int rc = misift_extract_sync(ctx, d_imgs, src_u8 ? 1 : 0, nframes, (long long)frame_stride, width, height, pitch,
                               num_octaves, init_blur, thresh, lowest_scale, scale_up ? 1 : 0, d_scratch,
                               (SiftPointD *)d_pts, max_pts, num_pts_out);
  if (rc) return rc;
  return resolve_profile(ctx);
}

extern "C" int misift_extract_batch_u8(misift_ctx *ctx, const unsigned char *d_imgs, int nframes, size_t frame_stride,
                                       int width, int height, int pitch, int num_octaves, float init_blur,
                                       float thresh, float lowest_scale, float *d_scratch, void *d_pts, int max_pts,
                                       int *num_pts_out)
{
  ARG_CHECK(num_pts_out != nullptr);
  int rc = misift_extract_sync(ctx, d_imgs, 1, nframes, (long long)frame_stride, width, height, pitch, num_octaves,
                        init_blur, thresh, lowest_scale, 0, d_scratch, (SiftPointD *)d_pts, max_pts, num_pts_out);
  if (rc) return rc;
  return resolve_profile(ctx);
}

extern "C" int misift_extract_batch_async(misift_ctx *ctx, const float *d_imgs, int nframes, size_t frame_stride,
                                          int width, int height, int pitch, int num_octaves, float init_blur,
                                          float thresh, float lowest_scale, float *d_scratch, void *d_pts,
                                          int max_pts, int *d_counts_out)
{
  ARG_CHECK(ctx != nullptr);
  TinyScope tiny(ctx, width, height, num_octaves);
  int rc = misift_extract_enqueue(ctx, d_imgs, 0, nframes, (long long)frame_stride, width, height, pitch,
                                  num_octaves, init_blur, thresh, lowest_scale, 0, d_scratch, (SiftPointD *)d_pts,
                                  max_pts);
  if (rc) return rc;
  if (d_counts_out) return launch_export_counts(ctx, nframes, num_octaves, max_pts, d_counts_out, nullptr);
  return MISIFT_OK;
}

extern "C" int misift_extract_batch_packed_async(misift_ctx *ctx, const float *d_imgs, int nframes,
                                                 size_t frame_stride, int width, int height, int pitch,
                                                 int num_octaves, float init_blur, float thresh, float lowest_scale,
                                                 float *d_scratch, void *d_pts, int max_pts, int *d_counts_out,
                                                 int *d_offsets_out, void *d_packed_out)
{
  ARG_CHECK(ctx && d_counts_out && d_offsets_out && d_packed_out);
  TinyScope tiny(ctx, width, height, num_octaves);
  CtxExtra *px = extra(ctx);
  if (!px->lanes.empty()) {
    // batches in flight: this call goes to the next pipeline of the ring, behind a marker on the caller's stream
    const size_t k = px->lanes.size();
    misift_ctx *lane = px->lanes[px->ticket % k];
    hipEvent_t done = px->lane_done[px->ticket % (2 * k)];
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(px->ev_in, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(lane->stream, px->ev_in, 0));
    lane->opt = ctx->opt;
    lane->profile = ctx->profile;
    const int rc = misift_extract_batch_packed_async(lane, d_imgs, nframes, frame_stride, width, height, pitch, num_octaves,
                                                     init_blur, thresh, lowest_scale, d_scratch, d_pts, max_pts, d_counts_out,
                                                     d_offsets_out, d_packed_out);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(done, lane->stream));
    px->last_lane = lane;
    px->last_done = done;
    px->ticket++;
    return MISIFT_OK;
  }
  if (ctx->opt.fused) {
    // merged-octave path: descr_all writes the packed array itself (no separate packing pass); d_pts may be NULL
    ctx->pack_counts = d_counts_out; ctx->pack_offsets = d_offsets_out; ctx->pack_dst = (SiftPointD *)d_packed_out;
    const int rc = misift_extract_enqueue(ctx, d_imgs, 0, nframes, (long long)frame_stride, width, height, pitch,
                                          num_octaves, init_blur, thresh, lowest_scale, 0, d_scratch,
                                          (SiftPointD *)d_pts, max_pts);
    ctx->pack_counts = nullptr; ctx->pack_offsets = nullptr; ctx->pack_dst = nullptr;
    return rc;
  }
  ARG_CHECK(d_pts != nullptr);
  int rc = misift_extract_enqueue(ctx, d_imgs, 0, nframes, (long long)frame_stride, width, height, pitch,
                                  num_octaves, init_blur, thresh, lowest_scale, 0, d_scratch, (SiftPointD *)d_pts,
                                  max_pts);
  if (rc) return rc;
  rc = launch_export_counts(ctx, nframes, num_octaves, max_pts, d_counts_out, d_offsets_out);
  if (rc) return rc;
  return launch_pack_records(ctx, (const SiftPointD *)d_pts, max_pts, nframes, d_offsets_out,
                             (SiftPointD *)d_packed_out);
}

extern "C" int misift_get_counters(misift_ctx *ctx, int frame, unsigned int *counters17)
{
  ARG_CHECK(ctx != nullptr);
  ctx = result_ctx(ctx);               // batches in flight: the pipeline that took the most recent batch
  ARG_CHECK(counters17 && frame >= 0 && frame < ctx->cap_frames);
  unsigned tmp[CNT_STRIDE];
  HIP_TRY(hipMemcpyAsync(tmp, ctx->d_counters + (size_t)frame * CNT_STRIDE, sizeof(tmp), hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  memcpy(counters17, tmp, sizeof(unsigned) * 17);
  return MISIFT_OK;
}

// Diagnostic: all CNT_STRIDE (64) words of one frame's counter block — the reference's 17 counters, then the library's
// own (candidates / detections / duplicates per octave, overflow flags); frame == number of frames of the last call
// reads the spare block that holds the call's flags.
extern "C" int misift_get_counter_block(misift_ctx *ctx, int frame, unsigned int *words64)
{
  ARG_CHECK(ctx != nullptr);
  ctx = result_ctx(ctx);
  ARG_CHECK(words64 && frame >= 0 && frame <= ctx->cap_frames);
  HIP_TRY(hipMemcpyAsync(words64, ctx->d_counters + (size_t)frame * CNT_STRIDE, sizeof(unsigned) * CNT_STRIDE,
                         hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_set_counters(misift_ctx *ctx, int frame, const unsigned int *counters17)
{
  ARG_CHECK(ctx && counters17 && frame >= 0 && frame < ctx->cap_frames);
  unsigned tmp[CNT_STRIDE];
  memset(tmp, 0, sizeof(tmp));
  memcpy(tmp, counters17, sizeof(unsigned) * 17);
  HIP_TRY(hipMemcpyAsync(ctx->d_counters + (size_t)frame * CNT_STRIDE, tmp, sizeof(tmp), hipMemcpyHostToDevice,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

// ------------------------------------------------------- stage entry points
extern "C" int misift_lowpass(misift_ctx *ctx, const float *d_src, int width, int height, int spitch, float *d_dst,
                              int dpitch, float sigma)
{
  ARG_CHECK(ctx && d_src && d_dst && width > 0 && height > 0 && spitch >= width && dpitch >= width);
  float k9[9];
  lowpass_taps(sigma, k9);
  StripGeom g = make_geom(ctx, width, height, spitch, 1, 0, width, height, 62);
  int rc = launch_lowpass(ctx, d_src, 0, g, d_dst, dpitch, 0, k9, nullptr);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_lowpass_scaledown(misift_ctx *ctx, const float *d_src, int width, int height, int spitch,
                                        float *d_dst, int dpitch, float sigma, float *d_dst2, int dpitch2)
{
  ARG_CHECK(ctx && d_src && d_dst && d_dst2 && width >= 4 && height >= 8);
  HIP_TRY(hipSetDevice(ctx->device));
  float k9[9], k5[5];
  lowpass_taps(sigma, k9);
  scaledown_taps(0.5f, k5);
  StripGeom g = make_geom(ctx, width, height, spitch, 1, 0, width, height, 60);
  int done = 0;
  int rc = launch_lowpass_down(ctx, d_src, 0, g, d_dst, dpitch, 0, k9, d_dst2, dpitch2, 0, k5, &done, nullptr);
  if (rc) return rc;
  if (!done) {
    misift_set_error("misift_lowpass_scaledown: shape not supported by the fused kernel (width %% 4, alignment)");
    return MISIFT_EINVAL;
  }
  return resolve_profile(ctx);
}

extern "C" int misift_scaledown(misift_ctx *ctx, const float *d_src, int width, int height, int spitch, float *d_dst,
                                int dpitch)
{
  ARG_CHECK(ctx && d_src && d_dst && width > 1 && height > 1 && spitch >= width && dpitch >= width / 2);
  float k5[5];
  scaledown_taps(0.5f, k5);
  StripGeom g = make_geom(ctx, width, height, spitch, 1, 0, width / 2, height / 2, 62);
  int rc = launch_scaledown(ctx, d_src, g, d_dst, dpitch, 0, k5);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_scaleup(misift_ctx *ctx, const float *d_src, int width, int height, int spitch, float *d_dst,
                              int dpitch)
{
  ARG_CHECK(ctx && d_src && d_dst && width > 0 && height > 0 && spitch >= width && dpitch >= 2 * width &&
            (dpitch & 1) == 0);
  int rc = launch_scaleup(ctx, d_src, 0, width, height, spitch, 0, 1, d_dst, dpitch, 0);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_laplace(misift_ctx *ctx, const float *d_base, int width, int height, int pitch, int num_octaves,
                              int octave, float *d_dog)
{
  ARG_CHECK(ctx && d_base && d_dog && width > 0 && height > 0 && pitch >= width);
  ARG_CHECK(num_octaves >= 1 && num_octaves <= MISIFT_MAX_OCTAVES && octave >= 1 && octave <= num_octaves);
  float table[8 * 12 * 16];
  misift_laplace_taps(num_octaves, table);
  StripGeom g = make_geom(ctx, width, height, pitch, 1, 0, width, height, 62);
  int rc = launch_laplace(ctx, d_base, g, d_dog, 0, octave_taps(table, octave));
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_reset_counters(misift_ctx *ctx, int max_pts)
{
  ARG_CHECK(ctx && max_pts >= 1);
  int rc = misift_ensure_frames(ctx, 1, 2 * (size_t)max_pts);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned) * CNT_STRIDE, ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_findpoints(misift_ctx *ctx, const float *d_dog, int width, int height, int pitch, float thresh,
                                 float edge_limit, float lowest_scale, float subsampling, int octave, void *d_pts,
                                 int max_pts)
{
  ARG_CHECK(ctx && d_dog && d_pts && width >= 16 && height >= 16 && pitch >= width && width < 16384 && height < 16384);
  ARG_CHECK(octave >= 1 && octave <= MISIFT_MAX_OCTAVES && max_pts >= 1);
  size_t cap = (size_t)width * height / 4;
  if (cap < 65536) cap = 65536;
  int rc = misift_ensure_frames(ctx, 1, cap);
  if (rc) return rc;
  StripGeom g = make_geom(ctx, width, height, pitch, 1, 0, width, height, 62);
  rc = launch_detect(ctx, d_dog, g, 0, thresh, octave);
  if (rc) return rc;
  if (ctx->opt.reference_cap) {
    rc = launch_refcap(ctx, width, height, 1, octave);
    if (rc) return rc;
  }
  rc = launch_refine(ctx, d_dog, 0, nullptr, 0, nullptr, width, height, pitch, 1, thresh, edge_limit, 1.0f / NUM_SCALES,
                     lowest_scale, subsampling, octave, (SiftPointD *)d_pts, max_pts);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_dog_findpoints(misift_ctx *ctx, const float *d_base, int width, int height, int pitch,
                                     int num_octaves, int octave, float thresh, float edge_limit, float lowest_scale,
                                     float subsampling, void *d_pts, int max_pts)
{
  ARG_CHECK(ctx && d_base && d_pts && width >= 16 && height >= 16 && pitch >= width && width < 16384 && height < 16384);
  ARG_CHECK(num_octaves >= 1 && num_octaves <= MISIFT_MAX_OCTAVES && octave >= 1 && octave <= num_octaves);
  size_t cap = (size_t)width * height / 4;
  if (cap < 65536) cap = 65536;
  int rc = misift_ensure_frames(ctx, 1, cap);
  if (rc) return rc;
  float table[8 * 12 * 16];
  misift_laplace_taps(num_octaves, table);
  const LaplaceTaps taps = octave_taps(table, octave);
  StripGeom g = make_geom(ctx, width, height, pitch, 1, 0, width, height, 60);
  rc = launch_dog_scan(ctx, d_base, g, taps, thresh, octave);
  if (rc) return rc;
  rc = launch_refine(ctx, nullptr, 0, d_base, 0, &taps, width, height, pitch, 1, thresh, edge_limit, 1.0f / NUM_SCALES,
                     lowest_scale, subsampling, octave, (SiftPointD *)d_pts, max_pts);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_orientations(misift_ctx *ctx, const float *d_base, int width, int height, int pitch, int octave,
                                   void *d_pts, int max_pts)
{
  ARG_CHECK(ctx && d_base && d_pts && width > 0 && height > 0 && pitch >= width && octave >= 1 &&
            octave <= MISIFT_MAX_OCTAVES);
  int rc = launch_orient(ctx, d_base, 0, width, height, pitch, 1, octave, (SiftPointD *)d_pts, max_pts);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_descriptors(misift_ctx *ctx, const float *d_base, int width, int height, int pitch,
                                  float subsampling, int octave, void *d_pts, int max_pts)
{
  ARG_CHECK(ctx && d_base && d_pts && width > 0 && height > 0 && pitch >= width && octave >= 1 &&
            octave <= MISIFT_MAX_OCTAVES);
  int rc = launch_descr(ctx, d_base, 0, width, height, pitch, 1, subsampling, octave, (SiftPointD *)d_pts, max_pts);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

// Test-only: evaluate the DEVICE copy of one written-out elementary function on n inputs (0 = det_exp2(x),
// 1 = det_atan2(y, x), 2 = det_exp(x), 3 = det_sincos(x) -> out = sin, out2 = cos).  The kernels and the oracle share
// these expressions, so parity tests cannot see an error in them; tests/test_gpu_parity.py checks the bits against the
// oracle's copy and tests/test_oracle_cpu.py the values against float64 libm.
extern "C" int misift_test_elementary(misift_ctx *ctx, int fn, const float *d_x, const float *d_y, float *d_out,
                                      float *d_out2, int n)
{
  ARG_CHECK(ctx && fn >= 0 && fn <= 3 && d_x && d_out && n >= 0 && (fn != 1 || d_y) && (fn != 3 || d_out2));
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return MISIFT_OK;
  int rc = fn == 0 ? launch_test_exp2(ctx, d_x, d_out, n) : launch_test_points_fn(ctx, fn, d_x, d_y, d_out, d_out2, n);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_rescale_positions(misift_ctx *ctx, void *d_pts, int npts, float scale)
{
  ARG_CHECK(ctx && d_pts && npts >= 0);
  int rc = launch_rescale(ctx, (SiftPointD *)d_pts, npts, scale);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MISIFT_OK;
}

// ----------------------------------------------------------------- matching
extern "C" int misift_match_rows(misift_ctx *ctx, void *d_pts1, int row_begin, int row_count, const void *d_pts2, int n2)
{
  ARG_CHECK(ctx && row_begin >= 0 && row_count >= 0 && n2 >= 0);
  if (row_count == 0 || n2 == 0) return MISIFT_OK;       // matching.cu:1095-1096
  ARG_CHECK(d_pts1 && d_pts2);
  RoctxRange range("misift_match");
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->want_match_flag = 1;
  ctx->match_flagged = 0;
  int rc = launch_match(ctx, (SiftPointD *)d_pts1, row_begin, row_count, (const SiftPointD *)d_pts2, n2);
  ctx->want_match_flag = 0;
  if (rc) return rc;
  if (ctx->match_flagged) rc = wait_host_flag(ctx, ctx->h_flags, ctx->match_seq);      // matching.cu:1191
  else HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (rc) return rc;
  return resolve_profile(ctx);
}

// Test-only: the two-launch column split of the sharded matcher (multigpu.hip) on one set-2 buffer — super-tiles
// [own_tile_begin, own_tile_end) first, the rest second, one merge — so its result bits and its cost can be checked
// without a communicator.
extern "C" int misift_test_match_split(misift_ctx *ctx, void *d_pts1, int n1, const void *d_pts2, int n2,
                                       int own_tile_begin, int own_tile_end)
{
  ARG_CHECK(ctx && n1 >= 0 && n2 >= 0 && own_tile_begin >= 0 && own_tile_end >= own_tile_begin);
  if (n1 == 0 || n2 == 0) return MISIFT_OK;
  ARG_CHECK(d_pts1 && d_pts2);
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = launch_match_split(ctx, (SiftPointD *)d_pts1, 0, n1, (const SiftPointD *)d_pts2, n2,
                              (const SiftPointD *)d_pts2, own_tile_begin, own_tile_end, nullptr, MATCH_PHASE_ALL, 0);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return resolve_profile(ctx);
}

extern "C" int misift_match(misift_ctx *ctx, void *d_pts1, int n1, const void *d_pts2, int n2)
{
  return misift_match_rows(ctx, d_pts1, 0, n1, d_pts2, n2);
}

// ------------------------------------------------------------------- timing
extern "C" int misift_timer_start(misift_ctx *ctx)
{
  ARG_CHECK(ctx != nullptr);
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  return MISIFT_OK;
}

extern "C" int misift_timer_stop_ms(misift_ctx *ctx, float *ms_out)
{
  ARG_CHECK(ctx && ms_out);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return MISIFT_OK;
}

extern "C" int misift_profile_enable(misift_ctx *ctx, int on)
{
  ARG_CHECK(ctx != nullptr);
  int rc = resolve_profile(ctx);
  ctx->profile = on != 0;
  for (misift_ctx *l : extra(ctx)->lanes) {
    const int r2 = resolve_profile(l);
    l->profile = on != 0;
    if (!rc) rc = r2;
  }
  return rc;
}

extern "C" int misift_profile_reset(misift_ctx *ctx)
{
  ARG_CHECK(ctx != nullptr);
  int rc = resolve_profile(ctx);
  ctx->nprof = 0;
  for (misift_ctx *l : extra(ctx)->lanes) {
    const int r2 = resolve_profile(l);
    l->nprof = 0;
    if (!rc) rc = r2;
  }
  return rc;
}

extern "C" int misift_profile_read(misift_ctx *ctx, int cap, char (*names)[32], float *total_ms, int *calls, int *n_out)
{
  ARG_CHECK(ctx && names && total_ms && calls && n_out);
  int rc = resolve_profile(ctx);
  if (rc) return rc;
  int n = 0;
  // the context's own launches, then those of its pipelines (batches in flight), merged by kernel name
  std::vector<misift_ctx *> all(1, ctx);
  for (misift_ctx *l : extra(ctx)->lanes) {
    rc = resolve_profile(l);
    if (rc) return rc;
    all.push_back(l);
  }
  for (misift_ctx *c : all)
    for (int i = 0; i < c->nprof; i++) {
      int j = 0;
      while (j < n && strncmp(names[j], c->prof[i].name, 32) != 0) j++;
      if (j == n) {
        if (n == cap) continue;
        memcpy(names[n], c->prof[i].name, 32);
        total_ms[n] = 0.0f;
        calls[n] = 0;
        n++;
      }
      total_ms[j] += c->prof[i].total_ms;
      calls[j] += c->prof[i].calls;
    }
  *n_out = n;
  return MISIFT_OK;
}

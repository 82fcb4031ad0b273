// multigpu.hip — the multi-GPU entry points of the C-ABI (BASELINE configs 4 and 5) on RCCL over xGMI.
//
// The reference is single-GPU (one global device chosen by InitCuda, cudaSiftH.cu:19-37); north_star adds
//   config 4: a batch of frames sharded over the GPUs of a node, SiftData gathered on one rank;
//   config 5: 100k x 100k matching, row blocks of set 1 per GPU, set 2 all-gathered, results all-gathered.
// Both live BEHIND the boundary so that a C++ caller of cudaSift.h (the only kind the reference has,
// mainSift.cpp:25-93) can use 8 GPUs: one misift_ctx per device (one host thread or process each), one
// misift_comm per context.  No collective touches the extraction data path; the matcher has exactly the one
// exchange step it needs before the sweep and one after.
//
// RCCL is bound at run time (dlopen): libmisift.so carries no link dependency on it, a single-GPU user never
// loads it, and inside a process that already holds an RCCL (PyTorch ships its own librccl.so) that copy is
// reused instead of loading a second one.  xGMI is point to point (7 links per GPU): the variable-length
// gather is ONE message per sender (7 senders -> 7 distinct links into the root), never a ring of padded blocks.
//
// Transports.  Everything above the five primitives {all-gather, group start / send / recv / group end} — count
// staging, per-rank record counts and offsets, root placement, the -1 frames, the collective ENOMEM decision, the
// set-2 placement and the result all-gather of the sharded matcher — is ONE code path over a transport table with two
// implementations: RCCL (production) and an in-process LOOPBACK WORLD (misift_comm_create_loopback): N communicators
// owned by N host threads, normally on ONE device, exchanging through a shared rendezvous object with device-to-device
// copies.  The loopback world exists so that the N > 1 branches run on the hardware a developer has (SURVEY section 4:
// "fake N ranks on one GPU"); it is functional, never a scaling measurement.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <type_traits>
#include <vector>
#include <rccl/rccl.h>      // types and prototypes only; every call goes through the table below
#include "common.hpp"

// ------------------------------------------------------------------ RCCL binding
struct RcclApi {
  void *handle;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)(void);
  ncclResult_t (*GroupEnd)(void);
  const char *(*GetErrorString)(ncclResult_t);
};
// the table is filled by dlsym (untyped): pin every slot to the prototype of the header this file was compiled against
#define RCCL_SLOT_MATCHES(field, fn) \
  static_assert(std::is_same<decltype(RcclApi::field), decltype(&fn)>::value, "RcclApi::" #field " != " #fn " of rccl.h")
RCCL_SLOT_MATCHES(GetUniqueId, ncclGetUniqueId);
RCCL_SLOT_MATCHES(CommInitRank, ncclCommInitRank);
RCCL_SLOT_MATCHES(CommDestroy, ncclCommDestroy);
RCCL_SLOT_MATCHES(CommCount, ncclCommCount);
RCCL_SLOT_MATCHES(CommUserRank, ncclCommUserRank);
RCCL_SLOT_MATCHES(AllGather, ncclAllGather);
RCCL_SLOT_MATCHES(Send, ncclSend);
RCCL_SLOT_MATCHES(Recv, ncclRecv);
RCCL_SLOT_MATCHES(GroupStart, ncclGroupStart);
RCCL_SLOT_MATCHES(GroupEnd, ncclGroupEnd);
RCCL_SLOT_MATCHES(GetErrorString, ncclGetErrorString);
#undef RCCL_SLOT_MATCHES
static RcclApi g_rccl;
static int g_rccl_state = 0;      // 0 = untried, 1 = bound, -1 = unavailable

static std::mutex g_rccl_mutex;      // communicators are created from several host threads (one per device)
static int rccl_bind(void)
{
  std::lock_guard<std::mutex> lk(g_rccl_mutex);
  if (g_rccl_state) return g_rccl_state > 0 ? MISIFT_OK : MISIFT_ENODEV;
  void *h = nullptr;
  // an RCCL already mapped into the process wins (PyTorch's own copy has the soname librccl.so)
  const char *names[] = {"librccl.so", "librccl.so.1"};
  for (const char *n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  if (const char *e = getenv("MISIFT_RCCL_LIB"))
    if (!h) h = dlopen(e, RTLD_NOW);
  const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : paths)
    if (!h) h = dlopen(n, RTLD_NOW);
  if (!h) {
    g_rccl_state = -1;
    misift_set_error("RCCL not found (librccl.so.1): %s", dlerror());
    return MISIFT_ENODEV;
  }
  g_rccl.handle = h;
  bool ok = true;
#define BIND(field, sym)                                                          \
  do {                                                                            \
    *(void **)(&g_rccl.field) = dlsym(h, sym);                                    \
    if (!g_rccl.field) { ok = false; misift_set_error("RCCL symbol %s missing", sym); } \
  } while (0)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(CommCount, "ncclCommCount");
  BIND(CommUserRank, "ncclCommUserRank");
  BIND(AllGather, "ncclAllGather");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  g_rccl_state = ok ? 1 : -1;
  return ok ? MISIFT_OK : MISIFT_ENODEV;
}

#define NCCL_TRY(expr)                                                                         \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      misift_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return MISIFT_EHIP;                                                                      \
    }                                                                                          \
  } while (0)

#define MG_CHECK(cond)                                                        \
  do {                                                                        \
    if (!(cond)) {                                                            \
      misift_set_error("%s: invalid argument: %s", __func__, #cond);          \
      return MISIFT_EINVAL;                                                   \
    }                                                                         \
  } while (0)

// ------------------------------------------------------------------ communicator
struct GatherSlot {
  const int *d_counts;
  const void *d_packed;
  int nframes;
  hipEvent_t ready;      // recorded on the context stream when the batch was posted
  bool posted;
};

struct misift_comm;
// stream-ordered like NCCL: an operation is complete for the caller once `stream` has drained
struct Transport {
  int (*allgather)(misift_comm *, const void *send, void *recv, size_t bytes_per_rank, hipStream_t stream);
  int (*group_start)(misift_comm *);
  int (*send)(misift_comm *, const void *buf, size_t bytes, int peer, hipStream_t stream);
  int (*recv)(misift_comm *, void *buf, size_t bytes, int peer, hipStream_t stream);
  int (*group_end)(misift_comm *, hipStream_t stream);
};

struct PendingP2P { bool is_send; void *buf; size_t bytes; int peer; };

struct misift_comm {
  misift_ctx *ctx;
  const Transport *tp;
  ncclComm_t nccl;              // RCCL transport
  bool owns_nccl;
  misift_loopback_world *world; // loopback transport
  std::vector<PendingP2P> pending;
  int rank, nranks;
  hipStream_t stream;           // communication stream: high priority, beside the extraction on the context stream
  int *d_all_counts;            // [nranks][cap_frames] staging of the count all-gather
  int *h_all_counts;            // pinned mirror
  int cap_frames;
  std::vector<GatherSlot> slots;
  hipEvent_t ev_fork, ev_gathered;   // sharded matcher: inputs ready on the context stream / set 2 complete on the communication stream
  unsigned long long wire_bytes = 0, sent_bytes = 0;   // bytes this rank has RECEIVED / SENT over the links (misift_comm_wire_bytes)
  // HOST communicator (misift_comm_create_host): no device, no context — every buffer is host memory, the runtime calls
  // below become memcpy / malloc / nothing, the five primitives are the caller's callbacks.  The gather logic above the
  // transport is the same code either way, which is the point: the CPU-only suite runs IT over gloo, not a mirror.
  bool host = false;
  misift_host_transport hcb = {};
};

// what the exchange code needs from the runtime besides the transport, for device and host communicators alike
static int mg_set_device(misift_comm *c)
{
  if (!c->host) HIP_TRY(hipSetDevice(c->ctx->device));
  return MISIFT_OK;
}
static int mg_copy(misift_comm *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st)
{
  if (c->host) { if (bytes) memcpy(dst, src, bytes); return MISIFT_OK; }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, st));
  return MISIFT_OK;
}
static int mg_sync(misift_comm *c, hipStream_t st)
{
  if (!c->host) HIP_TRY(hipStreamSynchronize(st));
  return MISIFT_OK;
}
static int mg_alloc_counts(misift_comm *c, size_t ints)
{
  if (c->host) {
    c->d_all_counts = (int *)calloc(ints, sizeof(int));
    c->h_all_counts = (int *)calloc(ints, sizeof(int));
    if (!c->d_all_counts || !c->h_all_counts) { misift_set_error("out of host memory"); return MISIFT_ENOMEM; }
    return MISIFT_OK;
  }
  if (!c->d_all_counts) HIP_TRY(misift_dev_alloc((void **)&c->d_all_counts, sizeof(int) * ints, "gather_counts"));
  if (!c->h_all_counts) HIP_TRY(hipHostMalloc((void **)&c->h_all_counts, sizeof(int) * ints, hipHostMallocDefault));
  return MISIFT_OK;
}
static int mg_free_counts(misift_comm *c)
{
  if (c->host) { free(c->d_all_counts); free(c->h_all_counts); }
  else {
    if (c->d_all_counts) HIP_TRY(misift_dev_free(c->d_all_counts));
    if (c->h_all_counts) HIP_TRY(hipHostFree(c->h_all_counts));
  }
  c->d_all_counts = nullptr; c->h_all_counts = nullptr; c->cap_frames = 0;
  return MISIFT_OK;
}

// ---- RCCL transport
static int rccl_allgather(misift_comm *c, const void *send, void *recv, size_t bytes, hipStream_t st)
{
  NCCL_TRY(g_rccl.AllGather(send, recv, bytes, ncclUint8, c->nccl, st));
  return MISIFT_OK;
}
static int rccl_group_start(misift_comm *) { NCCL_TRY(g_rccl.GroupStart()); return MISIFT_OK; }
static int rccl_send(misift_comm *c, const void *buf, size_t bytes, int peer, hipStream_t st)
{
  NCCL_TRY(g_rccl.Send(buf, bytes, ncclUint8, peer, c->nccl, st));
  return MISIFT_OK;
}
static int rccl_recv(misift_comm *c, void *buf, size_t bytes, int peer, hipStream_t st)
{
  NCCL_TRY(g_rccl.Recv(buf, bytes, ncclUint8, peer, c->nccl, st));
  return MISIFT_OK;
}
static int rccl_group_end(misift_comm *, hipStream_t) { NCCL_TRY(g_rccl.GroupEnd()); return MISIFT_OK; }
static const Transport kRcclTransport = {rccl_allgather, rccl_group_start, rccl_send, rccl_recv, rccl_group_end};

// ---- host transport: the caller's callbacks on host memory (tests: torch.distributed / gloo)
#define HCB_TRY(expr, what)                                                                  \
  do {                                                                                       \
    const int r_ = (expr);                                                                   \
    if (r_) { misift_set_error("host transport: %s failed with %d", what, r_); return MISIFT_EHIP; } \
  } while (0)
static int host_allgather(misift_comm *c, const void *send, void *recv, size_t bytes, hipStream_t)
{
  HCB_TRY(c->hcb.allgather(c->hcb.user, send, recv, bytes), "all-gather");
  return MISIFT_OK;
}
static int host_group_start(misift_comm *) { return MISIFT_OK; }
static int host_send(misift_comm *c, const void *buf, size_t bytes, int peer, hipStream_t)
{
  HCB_TRY(c->hcb.send(c->hcb.user, buf, bytes, peer), "send");
  return MISIFT_OK;
}
static int host_recv(misift_comm *c, void *buf, size_t bytes, int peer, hipStream_t)
{
  HCB_TRY(c->hcb.recv(c->hcb.user, buf, bytes, peer), "recv");
  return MISIFT_OK;
}
static int host_group_end(misift_comm *c, hipStream_t)
{
  HCB_TRY(c->hcb.group_end(c->hcb.user), "group end");
  return MISIFT_OK;
}
static const Transport kHostTransport = {host_allgather, host_group_start, host_send, host_recv, host_group_end};

// ---- loopback transport: a rendezvous object shared by the N communicators of one process
struct misift_loopback_world {
  int nranks;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;                       // reusable barrier
  unsigned long long generation = 0;
  bool failed = false;                   // a rank gave up (timeout / size mismatch): everybody fails fast
  std::vector<const void *> ag_send;     // all-gather deposits, valid between the two barriers of one operation
  std::vector<size_t> ag_bytes;
  struct Msg { const void *buf; size_t bytes; bool posted; };
  std::vector<Msg> box;                  // point-to-point mailboxes [src * nranks + dst]
  std::vector<char> attached;
  double timeout_s = 60.0;
};

static bool lw_wait(misift_loopback_world *w, std::unique_lock<std::mutex> &lk, const char *what, bool (*pred)(void *), void *arg)
{
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(w->timeout_s);
  while (!pred(arg) && !w->failed) {
    if (w->cv.wait_until(lk, deadline) == std::cv_status::timeout && !pred(arg)) {
      w->failed = true;
      w->cv.notify_all();
      misift_set_error("loopback world: rank timed out in %s (a rank never made the matching call)", what);
      return false;
    }
  }
  if (w->failed) { misift_set_error("loopback world: a peer failed during %s", what); return false; }
  return true;
}

static int lw_barrier(misift_loopback_world *w, const char *what)
{
  std::unique_lock<std::mutex> lk(w->m);
  if (w->failed) { misift_set_error("loopback world: a peer failed before %s", what); return MISIFT_EHIP; }
  struct A { misift_loopback_world *w; unsigned long long g; } a = {w, w->generation};
  if (++w->arrived == w->nranks) {
    w->arrived = 0;
    w->generation++;
    w->cv.notify_all();
    return MISIFT_OK;
  }
  return lw_wait(w, lk, what, [](void *p) { A *x = (A *)p; return x->w->generation != x->g; }, &a) ? MISIFT_OK : MISIFT_EHIP;
}

static int loop_allgather(misift_comm *c, const void *send, void *recv, size_t bytes, hipStream_t st)
{
  misift_loopback_world *w = c->world;
  HIP_TRY(hipStreamSynchronize(st));                 // my contribution is complete, my receive buffer is free
  {
    std::lock_guard<std::mutex> lk(w->m);
    w->ag_send[c->rank] = send;
    w->ag_bytes[c->rank] = bytes;
  }
  int rc = lw_barrier(w, "all-gather (deposit)");
  if (rc) return rc;
  for (int r = 0; r < c->nranks; r++) {
    if (w->ag_bytes[r] != bytes) {
      misift_set_error("loopback all-gather: rank %d contributes %zu bytes, rank %d %zu", r, w->ag_bytes[r], c->rank, bytes);
      std::lock_guard<std::mutex> lk(w->m);
      w->failed = true;
      w->cv.notify_all();
      return MISIFT_EINVAL;
    }
    char *dst = (char *)recv + (size_t)r * bytes;
    if (bytes && dst != (const char *)w->ag_send[r])       // in-place contribution of this rank: nothing to move
      HIP_TRY(hipMemcpyAsync(dst, w->ag_send[r], bytes, hipMemcpyDefault, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return lw_barrier(w, "all-gather (release)");      // nobody's send buffer is reused before everybody has read it
}

static int loop_group_start(misift_comm *c) { c->pending.clear(); return MISIFT_OK; }
static int loop_send(misift_comm *c, const void *buf, size_t bytes, int peer, hipStream_t)
{
  c->pending.push_back({true, (void *)buf, bytes, peer});
  return MISIFT_OK;
}
static int loop_recv(misift_comm *c, void *buf, size_t bytes, int peer, hipStream_t)
{
  c->pending.push_back({false, buf, bytes, peer});
  return MISIFT_OK;
}
static int loop_group_end(misift_comm *c, hipStream_t st)
{
  misift_loopback_world *w = c->world;
  const int n = c->nranks;
  HIP_TRY(hipStreamSynchronize(st));                 // what I send is complete
  {
    std::lock_guard<std::mutex> lk(w->m);
    for (const PendingP2P &p : c->pending)
      if (p.is_send) w->box[(size_t)c->rank * n + p.peer] = {p.buf, p.bytes, true};
    w->cv.notify_all();
  }
  for (const PendingP2P &p : c->pending) {
    if (p.is_send) continue;
    misift_loopback_world::Msg msg;
    {
      std::unique_lock<std::mutex> lk(w->m);
      struct A { misift_loopback_world *w; size_t i; } a = {w, (size_t)p.peer * n + c->rank};
      if (!lw_wait(w, lk, "recv", [](void *q) { A *x = (A *)q; return x->w->box[x->i].posted; }, &a)) return MISIFT_EHIP;
      msg = w->box[a.i];
    }
    if (msg.bytes != p.bytes) {                      // the offset arithmetic of the two sides disagrees
      misift_set_error("loopback recv: rank %d expects %zu bytes from rank %d, which sends %zu", c->rank, p.bytes, p.peer, msg.bytes);
      std::lock_guard<std::mutex> lk(w->m);
      w->failed = true;
      w->cv.notify_all();
      return MISIFT_EINVAL;
    }
    if (p.bytes) HIP_TRY(hipMemcpyAsync(p.buf, msg.buf, p.bytes, hipMemcpyDefault, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  {
    std::unique_lock<std::mutex> lk(w->m);
    for (const PendingP2P &p : c->pending)           // consumed: the sender may reuse its buffer
      if (!p.is_send) w->box[(size_t)p.peer * n + c->rank].posted = false;
    w->cv.notify_all();
    for (const PendingP2P &p : c->pending) {
      if (!p.is_send) continue;
      struct A { misift_loopback_world *w; size_t i; } a = {w, (size_t)c->rank * n + p.peer};
      if (!lw_wait(w, lk, "send", [](void *q) { A *x = (A *)q; return !x->w->box[x->i].posted; }, &a)) return MISIFT_EHIP;
    }
  }
  c->pending.clear();
  return MISIFT_OK;
}
static const Transport kLoopbackTransport = {loop_allgather, loop_group_start, loop_send, loop_recv, loop_group_end};

extern "C" int misift_loopback_world_create(int nranks, misift_loopback_world **out)
{
  MG_CHECK(out && nranks >= 1 && nranks <= 64);
  misift_loopback_world *w = new misift_loopback_world();
  w->nranks = nranks;
  w->ag_send.assign((size_t)nranks, nullptr);
  w->ag_bytes.assign((size_t)nranks, 0);
  w->box.assign((size_t)nranks * nranks, {nullptr, 0, false});
  w->attached.assign((size_t)nranks, 0);
  if (const char *e = getenv("MISIFT_LOOPBACK_TIMEOUT_S")) w->timeout_s = atof(e) > 0 ? atof(e) : w->timeout_s;
  *out = w;
  return MISIFT_OK;
}

extern "C" void misift_loopback_world_destroy(misift_loopback_world *w) { delete w; }

extern "C" int misift_comm_unique_id(void *id128)
{
  MG_CHECK(id128 != nullptr);
  int rc = rccl_bind();
  if (rc) return rc;
  static_assert(MISIFT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return MISIFT_OK;
}

static int comm_finish_create(misift_ctx *ctx, ncclComm_t nc, bool owns, misift_loopback_world *world, int world_rank,
                              misift_comm **out)
{
  misift_comm *c = new misift_comm();
  c->ctx = ctx; c->nccl = nc; c->owns_nccl = owns; c->world = world;
  c->tp = world ? &kLoopbackTransport : &kRcclTransport;
  c->rank = 0; c->nranks = 1; c->stream = nullptr;
  c->d_all_counts = nullptr; c->h_all_counts = nullptr; c->cap_frames = 0; c->ev_gathered = nullptr; c->ev_fork = nullptr;
  c->slots.resize(MISIFT_GATHER_SLOTS);
  for (GatherSlot &s : c->slots) { memset(&s, 0, sizeof(s)); }
  *out = c;                                           // from here on misift_comm_destroy cleans up after a failure
  if (world) {
    c->nranks = world->nranks;
    c->rank = world_rank;
  } else {
    NCCL_TRY(g_rccl.CommCount(nc, &c->nranks));
    NCCL_TRY(g_rccl.CommUserRank(nc, &c->rank));
  }
  misift_warn_hw_queues("misift_comm_create");
  int lo = 0, hi = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
  for (GatherSlot &s : c->slots) HIP_TRY(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
  return MISIFT_OK;
}

extern "C" void misift_comm_destroy(misift_comm *c)
{
  if (!c) return;
  if (c->host) { mg_free_counts(c); delete c; return; }
  if (c->ctx) hipSetDevice(c->ctx->device);
  if (c->stream) { hipStreamSynchronize(c->stream); }
  if (c->nccl && c->owns_nccl && g_rccl_state > 0) g_rccl.CommDestroy(c->nccl);
  if (c->world) {
    std::lock_guard<std::mutex> lk(c->world->m);
    c->world->attached[c->rank] = 0;
  }
  if (c->stream) hipStreamDestroy(c->stream);
  for (GatherSlot &s : c->slots)
    if (s.ready) hipEventDestroy(s.ready);
  if (c->ev_gathered) hipEventDestroy(c->ev_gathered);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->d_all_counts) misift_dev_free(c->d_all_counts);
  if (c->h_all_counts) hipHostFree(c->h_all_counts);
  delete c;
}

extern "C" int misift_comm_create(misift_ctx *ctx, int nranks, int rank, const void *id128, misift_comm **out)
{
  MG_CHECK(ctx && out && id128 && nranks >= 1 && rank >= 0 && rank < nranks);
  *out = nullptr;
  int rc = rccl_bind();
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t nc = nullptr;
  NCCL_TRY(g_rccl.CommInitRank(&nc, nranks, id, rank));
  rc = comm_finish_create(ctx, nc, true, nullptr, 0, out);
  if (rc) { misift_comm_destroy(*out); *out = nullptr; }
  return rc;
}

extern "C" int misift_comm_adopt(misift_ctx *ctx, void *nccl_comm, misift_comm **out)
{
  MG_CHECK(ctx && out && nccl_comm);
  *out = nullptr;
  int rc = rccl_bind();
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  rc = comm_finish_create(ctx, (ncclComm_t)nccl_comm, false, nullptr, 0, out);
  if (rc) { misift_comm_destroy(*out); *out = nullptr; }
  return rc;
}

extern "C" int misift_comm_create_loopback(misift_ctx *ctx, misift_loopback_world *world, int rank, misift_comm **out)
{
  MG_CHECK(ctx && world && out && rank >= 0 && rank < world->nranks);
  *out = nullptr;
  {
    std::lock_guard<std::mutex> lk(world->m);
    MG_CHECK(!world->attached[rank]);
    world->attached[rank] = 1;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = comm_finish_create(ctx, nullptr, false, world, rank, out);
  if (rc) { misift_comm_destroy(*out); *out = nullptr; }
  return rc;
}

extern "C" int misift_comm_create_host(int nranks, int rank, const misift_host_transport *t, misift_comm **out)
{
  MG_CHECK(out && t && t->allgather && t->send && t->recv && t->group_end && nranks >= 1 && rank >= 0 && rank < nranks);
  misift_comm *c = new misift_comm();
  c->ctx = nullptr; c->nccl = nullptr; c->owns_nccl = false; c->world = nullptr;
  c->tp = &kHostTransport;
  c->rank = rank; c->nranks = nranks; c->stream = nullptr;
  c->d_all_counts = nullptr; c->h_all_counts = nullptr; c->cap_frames = 0; c->ev_gathered = nullptr; c->ev_fork = nullptr;
  c->slots.resize(MISIFT_GATHER_SLOTS);
  for (GatherSlot &s : c->slots) { memset(&s, 0, sizeof(s)); }
  c->host = true;
  c->hcb = *t;
  *out = c;
  return MISIFT_OK;
}

extern "C" int misift_comm_rank(const misift_comm *c) { return c ? c->rank : -1; }
extern "C" int misift_comm_size(const misift_comm *c) { return c ? c->nranks : 0; }

extern "C" int misift_comm_barrier(misift_comm *c)
{
  MG_CHECK(c != nullptr);
  int rc = mg_set_device(c);
  if (rc) return rc;
  // an all-gather of one int per rank on the communication stream, then a host wait: every rank has arrived
  if (c->cap_frames < 1) {                              // (a failed attempt leaves what it did allocate for the next one)
    rc = mg_alloc_counts(c, (size_t)c->nranks * 64);
    if (rc) return rc;
    if (!c->host) HIP_TRY(hipMemsetAsync(c->d_all_counts, 0, sizeof(int) * (size_t)c->nranks * 64, c->stream));
    c->cap_frames = 64;
  }
  rc = c->tp->allgather(c, c->d_all_counts + c->rank, c->d_all_counts, sizeof(int), c->stream);
  if (rc) return rc;
  return mg_sync(c, c->stream);
}

// ------------------------------------------------------------------ config 4: gather of SiftData
extern "C" int misift_gather_post(misift_ctx *ctx, misift_comm *c, int slot, const int *d_counts, int nframes,
                                  const void *d_packed)
{
  MG_CHECK(c && d_counts && d_packed && nframes >= 1);
  MG_CHECK(c->host || (ctx && c->ctx->device == ctx->device));       // a host communicator has no context: pass NULL
  MG_CHECK(slot >= 0 && slot < (int)c->slots.size());
  GatherSlot &s = c->slots[slot];
  s.d_counts = d_counts; s.d_packed = d_packed; s.nframes = nframes;
  if (!c->host) {
    HIP_TRY(hipSetDevice(ctx->device));
    // the batch queued last on the context (on its own stream, or on one of its pipelines with batches in flight)
    // produces these buffers
    HIP_TRY(hipEventRecord(s.ready, misift_ctx_result_stream(ctx)));
  }
  s.posted = true;
  return MISIFT_OK;
}

// Non-blocking companion of misift_gather_complete: *ready = 1 once the posted batch has finished on the GPU, i.e.
// once misift_gather_complete would only wait for the exchange itself (microseconds of counts + one message), not
// for the extraction kernels queued before the post.  A C++ caller polls this instead of parking a thread.
extern "C" int misift_gather_test(misift_comm *c, int slot, int *ready)
{
  MG_CHECK(c && ready && slot >= 0 && slot < (int)c->slots.size());
  GatherSlot &s = c->slots[slot];
  MG_CHECK(s.posted);
  if (c->host) { *ready = 1; return MISIFT_OK; }
  HIP_TRY(hipSetDevice(c->ctx->device));
  const hipError_t e = hipEventQuery(s.ready);
  if (e == hipSuccess) { *ready = 1; return MISIFT_OK; }
  if (e == hipErrorNotReady) { (void)hipGetLastError(); *ready = 0; return MISIFT_OK; }
  misift_set_error("misift_gather_test: hipEventQuery failed: %s", hipGetErrorString(e));
  return MISIFT_EHIP;
}

extern "C" int misift_gather_complete(misift_comm *c, int slot, int root, int *h_all_counts, void *d_recv,
                                      size_t capacity_records, size_t *h_rank_offsets)
{
  MG_CHECK(c && slot >= 0 && slot < (int)c->slots.size() && root >= 0 && root < c->nranks);
  GatherSlot &s = c->slots[slot];
  MG_CHECK(s.posted);
  MG_CHECK(h_all_counts != nullptr);
  MG_CHECK(c->rank != root || c->nranks == 1 || d_recv != nullptr);
  int rc0 = mg_set_device(c);
  if (rc0) return rc0;
  const int nf = s.nframes, nr = c->nranks;
  if (nf > c->cap_frames) {
    rc0 = mg_free_counts(c);
    if (!rc0) rc0 = mg_alloc_counts(c, (size_t)nr * nf);
    if (rc0) return rc0;
    c->cap_frames = nf;
  }
  if (!c->host) HIP_TRY(hipStreamWaitEvent(c->stream, s.ready, 0));
  // 1. per-frame counts of every rank (nframes ints per rank; every rank must post the same nframes)
  if (nr == 1) {         // nothing to gather: a plain copy
    rc0 = mg_copy(c, c->d_all_counts, s.d_counts, sizeof(int) * (size_t)nf, hipMemcpyDeviceToDevice, c->stream);
    if (rc0) return rc0;
  } else {
    int rc = c->tp->allgather(c, s.d_counts, c->d_all_counts, sizeof(int) * (size_t)nf, c->stream);
    if (rc) { s.posted = false; return rc; }
    c->wire_bytes += (unsigned long long)(nr - 1) * nf * sizeof(int);
    c->sent_bytes += (unsigned long long)(nr - 1) * nf * sizeof(int);
  }
  rc0 = mg_copy(c, c->h_all_counts, c->d_all_counts, sizeof(int) * (size_t)nr * nf, hipMemcpyDeviceToHost, c->stream);
  if (!rc0) rc0 = mg_sync(c, c->stream);             // the message sizes must be known on the host (NCCL API)
  if (rc0) return rc0;
  memcpy(h_all_counts, c->h_all_counts, sizeof(int) * (size_t)nr * nf);
  std::vector<size_t> nrec((size_t)nr, 0), off((size_t)nr + 1, 0);
  for (int r = 0; r < nr; r++) {
    for (int f = 0; f < nf; f++) {
      const int v = c->h_all_counts[(size_t)r * nf + f];
      if (v > 0) nrec[r] += (size_t)v;               // -1 marks an overflowed frame: no records
    }
    off[r + 1] = off[r] + nrec[r];
  }
  if (h_rank_offsets) memcpy(h_rank_offsets, off.data(), sizeof(size_t) * ((size_t)nr + 1));
  // Room on the root?  Every rank knows every count and passes the SAME capacity_records (the root's), so every rank
  // takes the same decision and nobody waits for a message that is never posted.
  if (off[nr] > capacity_records) {
    misift_set_error("misift_gather_complete: %zu records but room for %zu on the root", off[nr], capacity_records);
    s.posted = false;
    return MISIFT_ENOMEM;
  }
  // 2. ONE point-to-point message per sender with exactly its valid bytes (xGMI: distinct links into the root)
  int rc = c->tp->group_start(c);
  if (c->rank == root) {
    for (int r = 0; r < nr && !rc; r++) {
      if (r == root || nrec[r] == 0) continue;
      rc = c->tp->recv(c, (char *)d_recv + off[r] * sizeof(SiftPointD), nrec[r] * sizeof(SiftPointD), r, c->stream);
      c->wire_bytes += (unsigned long long)nrec[r] * sizeof(SiftPointD);
    }
  } else if (nrec[c->rank] && !rc) {
    rc = c->tp->send(c, s.d_packed, nrec[c->rank] * sizeof(SiftPointD), root, c->stream);
    c->sent_bytes += (unsigned long long)nrec[c->rank] * sizeof(SiftPointD);
  }
  // a failure inside the group must still close it: an open NCCL group swallows every later collective of this thread
  const int rc_end = c->tp->group_end(c, c->stream);
  if (rc || rc_end) { s.posted = false; return rc ? rc : rc_end; }
  if (c->rank == root && d_recv && nrec[root]) {
    rc = mg_copy(c, (char *)d_recv + off[root] * sizeof(SiftPointD), s.d_packed, nrec[root] * sizeof(SiftPointD),
                 hipMemcpyDeviceToDevice, c->stream);
    if (rc) { s.posted = false; return rc; }
  }
  rc = mg_sync(c, c->stream);                        // the slot's buffers are free again when this returns
  s.posted = false;
  return rc;
}

// Bytes this rank has received / sent over the links since the communicator was created (payload of the collectives
// and point-to-point messages above; what a ring all-gather forwards on behalf of others is not counted).
extern "C" int misift_comm_wire_bytes(misift_comm *c, unsigned long long *received, unsigned long long *sent)
{
  MG_CHECK(c != nullptr);
  if (received) *received = c->wire_bytes;
  if (sent) *sent = c->sent_bytes;
  return MISIFT_OK;
}

// ------------------------------------------------------------------ config 5: row-block matcher
struct MatchResult { float score, ambiguity; int match; };     // the 12 B/row result of SURVEY 8e
static_assert(sizeof(MatchResult) == 12, "result row");

// a 576-byte record -> the 528-byte match column the sharded matcher ships: 33 float4 per column (32 of descriptor, then
// {xpos, ypos, 0, 0}); thread t writes float4 t % 33 of column t / 33
__global__ __launch_bounds__(256) void pack_match_columns_kernel(const SiftPointD *__restrict__ recs, int n,
                                                                 float4 *__restrict__ out)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t colm = t / 33;
  const int q = (int)(t - colm * 33);
  if (colm >= (size_t)n) return;
  const SiftPointD &r = recs[colm];
  out[t] = q < 32 ? reinterpret_cast<const float4 *>(r.data)[q] : make_float4(r.xpos, r.ypos, 0.0f, 0.0f);
}

__global__ void pack_match_results_kernel(const SiftPointD *__restrict__ rows, int n, MatchResult *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    MatchResult r;
    r.score = rows[i].score; r.ambiguity = rows[i].ambiguity; r.match = rows[i].match;
    out[i] = r;
  }
}

extern "C" int misift_match_sharded(misift_ctx *ctx, misift_comm *c, void *d_rows1, int row_count, const void *d_shard2,
                                    int shard_count, void *d_set2_all, void *d_results_all)
{
  MG_CHECK(c && !c->host);                              // the sweep runs on the GPU: device communicators only
  MG_CHECK(ctx && c->ctx == ctx && row_count >= 0 && shard_count >= 0);
  MG_CHECK(row_count == 0 || d_rows1);
  MG_CHECK(shard_count == 0 || (d_shard2 && d_set2_all));
  HIP_TRY(hipSetDevice(ctx->device));
  const int nr = c->nranks;
  const long long n2 = (long long)shard_count * nr;
  MG_CHECK(n2 < (1ll << 31));
  // The shard's 576-byte records are re-packed INTO d_set2_all (528-byte columns): the two buffers must not overlap — an
  // in-place call (legal while the records themselves were all-gathered, r03) would read records the pack has already
  // overwritten.  After the call d_set2_all holds match columns, not SiftPoint records (include/misift.h).
  if (shard_count) {
    const char *s0 = (const char *)d_shard2, *s1 = s0 + (size_t)shard_count * sizeof(SiftPointD);
    const char *a0 = (const char *)d_set2_all, *a1 = a0 + (size_t)n2 * MISIFT_MATCH_COLUMN_BYTES;
    if (s0 < a1 && a0 < s1) {
      misift_set_error("misift_match_sharded: d_shard2 overlaps d_set2_all (the shard is packed into d_set2_all; pass separate buffers)");
      return MISIFT_EINVAL;
    }
  }
  // 1. replicate set 2: the shard is packed into 528-byte match columns (descriptor + position: all the sweep reads of a
  //    record) at its own place in d_set2_all, then all-gathered in place (52.8 MB at 100 k; r03 shipped the 576-byte
  //    records).  With more than one rank the exchange runs on the communication stream while the context stream already
  //    sweeps the super-tiles that lie entirely inside this rank's OWN shard; the rest of the columns follow behind the
  //    all-gather's event.  Same result bits as one sweep (launch_match_split).
  const long long own_begin = (long long)c->rank * shard_count, own_end = own_begin + shard_count;
  int own_t0 = (int)((own_begin + 63) / 64), own_t1 = (int)(own_end / 64);
  const bool split = nr > 1 && shard_count && row_count && own_t1 > own_t0 && !getenv("MISIFT_MATCH_NO_OVERLAP");
  hipEvent_t gathered = nullptr;
  char *cols_all = (char *)d_set2_all;
  if (shard_count) {
    char *mine = cols_all + (size_t)own_begin * MISIFT_MATCH_COLUMN_BYTES;
    hipLaunchKernelGGL(pack_match_columns_kernel, dim3((unsigned)(((size_t)shard_count * 33 + 255) / 256)), dim3(256), 0,
                       ctx->stream, (const SiftPointD *)d_shard2, shard_count, reinterpret_cast<float4 *>(mine));
    HIP_TRY(hipGetLastError());
    if (split) {
      if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(c->ev_fork, ctx->stream));  // the columns are packed, d_set2_all's previous readers are done
      HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_fork, 0));
    }
    hipStream_t ag_stream = split ? c->stream : ctx->stream;
    if (split) {
      // the own-shard sweep is enqueued BEFORE the exchange is posted (the loopback transport blocks the host in it)
      int rc = launch_match_split(ctx, (SiftPointD *)d_rows1, 0, row_count, (const SiftPointD *)cols_all, (int)n2,
                                  (const SiftPointD *)cols_all, own_t0, own_t1, nullptr, MATCH_PHASE_OWN, 1);
      if (rc) return rc;
    }
    int rc = c->tp->allgather(c, mine, cols_all, (size_t)shard_count * MISIFT_MATCH_COLUMN_BYTES, ag_stream);
    if (rc) return rc;
    c->wire_bytes += (unsigned long long)(nr - 1) * shard_count * MISIFT_MATCH_COLUMN_BYTES;
    c->sent_bytes += (unsigned long long)(nr - 1) * shard_count * MISIFT_MATCH_COLUMN_BYTES;
    if (split) {
      if (!c->ev_gathered) HIP_TRY(hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(c->ev_gathered, c->stream));
      gathered = c->ev_gathered;
    }
  }
  // 2. this rank's rows against (the rest of) set 2 (fp32 MFMA sweep, same kernel as misift_match) and the merge
  if (row_count && n2) {
    int rc = launch_match_split(ctx, (SiftPointD *)d_rows1, 0, row_count, (const SiftPointD *)cols_all, (int)n2,
                                (const SiftPointD *)cols_all, split ? own_t0 : 0, split ? own_t1 : 0, gathered,
                                split ? MATCH_PHASE_REST : MATCH_PHASE_ALL, 1);
    if (rc) return rc;
  }
  // 3. results: 12 B per row, all-gathered so every rank holds the whole answer (row blocks in rank order)
  if (d_results_all && row_count) {
    MatchResult *all = (MatchResult *)d_results_all;
    MatchResult *mine = all + (size_t)c->rank * row_count;
    hipLaunchKernelGGL(pack_match_results_kernel, dim3((row_count + 255) / 256), dim3(256), 0, ctx->stream,
                       (const SiftPointD *)d_rows1, row_count, mine);
    HIP_TRY(hipGetLastError());
    int rc = c->tp->allgather(c, mine, all, (size_t)row_count * sizeof(MatchResult), ctx->stream);
    if (rc) return rc;
    c->wire_bytes += (unsigned long long)(nr - 1) * row_count * sizeof(MatchResult);
    c->sent_bytes += (unsigned long long)(nr - 1) * row_count * sizeof(MatchResult);
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));         // matching.cu:1191: MatchSiftData returns with the results in place
  return MISIFT_OK;
}

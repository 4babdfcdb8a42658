// comm.hip -- RCCL called from the C ABI (round 5).
//
// Until round 4 the block-column driver's collectives went through torch.distributed's ProcessGroupNCCL on
// ExternalStreams: every panel chunk paid the process group's event hops, torch was a hard dependency of the sharded
// path, and its bundled HIP runtime next to the library's was the cause of GPUTEST_r04.  Here the library issues
// ncclBroadcast / ncclReduce / ncclAllReduce itself, on ITS OWN streams (the context's main and priority stream), on
// plain device pointers.  The communicator is built from a 128-byte ncclUniqueId that rank 0 creates and the CALLER
// distributes (tinygp_amd/comm.py: a TCP exchange on MASTER_ADDR, a file, or torch.distributed for that one message).
//
// librccl is dlopen'ed on first use -- a single-GPU process never maps it, and the library has no link-time dependency
// on it.  "librccl.so.1" resolves to the copy already in the process (PyTorch-ROCm bundles one) or else to ROCm's;
// either binds to the process's ONE HIP runtime by SONAME (tinygp_amd/_ffi.py::share_hip_runtime).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include <rccl/rccl.h>

#include "tgp_common.h"

using namespace tgp;

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;  // optional (NCCL >= 2.18)
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib != nullptr) return TGP_OK;
  // the copy that SHIPS WITH the HIP runtime this process uses comes first (a PyTorch-ROCm wheel bundles both, and its
  // runtime may be older than ROCm's own librccl expects); then whatever "librccl.so.1" names -- the copy already
  // mapped, else the loader's
  std::string beside;
  {
    Dl_info di{};
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &di) && di.dli_fname) {
      beside = di.dli_fname;
      const size_t slash = beside.rfind('/');
      beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
    }
  }
  const std::string b1 = beside.empty() ? std::string() : beside + "librccl.so.1";
  const std::string b0 = beside.empty() ? std::string() : beside + "librccl.so";
  const char* names[] = {getenv("TGP_RCCL_LIBRARY"), b1.c_str(), b0.c_str(), "librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  std::string tried;
  for (const char* nm : names) {
    if (nm == nullptr || !*nm) continue;
    lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
    tried += std::string(tried.empty() ? "" : "; ") + dlerror();
  }
  if (!lib) {
    set_error("RCCL is not loadable (%s): the multi-GPU path needs librccl", tried.c_str());
    return TGP_E_UNSUPPORTED;
  }
  RcclApi a;
  a.lib = lib;
#define SYM(field, name)                                                         \
  do {                                                                           \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));             \
    if (!a.field) {                                                              \
      set_error("librccl has no symbol %s", name);                               \
      dlclose(lib); /* (advisor r5: the handle leaked on this path) */           \
      return TGP_E_UNSUPPORTED;                                                  \
    }                                                                            \
  } while (0)
  SYM(GetVersion, "ncclGetVersion");
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommAbort, "ncclCommAbort");
  SYM(GetErrorString, "ncclGetErrorString");
  SYM(Broadcast, "ncclBroadcast");
  SYM(Reduce, "ncclReduce");
  SYM(AllReduce, "ncclAllReduce");
#undef SYM
  a.CommSplit = reinterpret_cast<decltype(a.CommSplit)>(dlsym(lib, "ncclCommSplit"));  // (may be absent: see tgp_comm_create)
  g_rccl = a;
  return TGP_OK;
}

#define TGP_NCCL_TRY(expr)                                                                      \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess) {                                                                    \
      set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return TGP_E_HIP;                                                                         \
    }                                                                                           \
  } while (0)

inline ncclDataType_t nccl_dtype(int dtype) { return dtype == TGP_F64 ? ncclFloat64 : ncclFloat32; }

}  // namespace

struct tgp_comm {
  tgp_ctx* ctx = nullptr;
  // ONE communicator PER STREAM (round 6).  RCCL orders the operations of a communicator and inserts
  // cross-stream dependencies when the stream changes: with one communicator on both streams the look-ahead broadcast of a
  // panel (priority stream) and a main-stream reduce / all-reduce would be ordered against each other -- the overlap the
  // block-column driver is built on would silently serialise on a real node (at world size 1 it is invisible).
  // comm[0]: main stream, comm[1]: priority stream (a split of comm[0], or a second ncclCommInitRank whose id rank 0
  // sends through comm[0] when the library has no ncclCommSplit).
  ncclComm_t comm[2] = {nullptr, nullptr};
  int world = 1, rank = 0;
  // stream-to-stream ordering around the collectives (a broadcast issued on the priority stream, consumed on the main
  // stream): a small ring of events; a ticket names the event AND its generation, a stale ticket is an error
  static constexpr int NEV = 64;
  hipEvent_t ev[NEV] = {};
  int64_t issued = 0;
};

#define CTX_GUARD(ctx)                                                \
  TGP_ARG_CHECK((ctx) != nullptr, "null context");                    \
  std::unique_lock<std::recursive_mutex> _tgp_lock((ctx)->mu);        \
  TGP_HIP_TRY(hipSetDevice((ctx)->device))

#define COMM_GUARD(c)                                                          \
  TGP_ARG_CHECK((c) != nullptr && (c)->ctx != nullptr, "null communicator");   \
  std::unique_lock<std::recursive_mutex> _tgp_lock((c)->ctx->mu);            \
  TGP_HIP_TRY(hipSetDevice((c)->ctx->device))

static hipStream_t comm_stream(tgp_comm* c, int which) { return which == 0 ? c->ctx->stream : c->ctx->panel_stream; }
static ncclComm_t comm_of(tgp_comm* c, int which) { return c->comm[which == 0 ? 0 : 1]; }

extern "C" {

int tgp_comm_unique_id(void* id_out, int32_t* version_out) {
  TGP_ARG_CHECK(id_out != nullptr, "null output pointer");
  TGP_TRY(load_rccl());
  ncclUniqueId id;
  TGP_NCCL_TRY(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == TGP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id_out, &id, sizeof(id));
  if (version_out) {
    int v = 0;
    TGP_NCCL_TRY(g_rccl.GetVersion(&v));
    *version_out = v;
  }
  return TGP_OK;
}

int tgp_comm_create(tgp_ctx* ctx, int32_t world, int32_t rank, const void* id, tgp_comm** out) {
  TGP_ARG_CHECK(ctx != nullptr && id != nullptr && out != nullptr, "null argument");
  TGP_ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
  std::unique_lock<std::recursive_mutex> lk(ctx->mu);
  TGP_HIP_TRY(hipSetDevice(ctx->device));
  TGP_ARG_CHECK(ctx->panel_stream != nullptr, "the context has no panel stream");
  TGP_TRY(load_rccl());
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  tgp_comm* c = new tgp_comm();
  c->ctx = ctx;
  c->world = world;
  c->rank = rank;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm[0], world, uid, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, g_rccl.GetErrorString(r));
    delete c;
    return TGP_E_HIP;
  }
  // the priority stream's own communicator
  r = ncclInternalError;
  if (g_rccl.CommSplit != nullptr && getenv("TGP_COMM_NO_SPLIT") == nullptr)
    r = g_rccl.CommSplit(c->comm[0], 0, rank, &c->comm[1], nullptr);
  if (r != ncclSuccess || c->comm[1] == nullptr) {
    // no ncclCommSplit (or it refused): a second id, generated by rank 0 and sent through the first communicator
    c->comm[1] = nullptr;
    ncclUniqueId id2;
    std::memset(&id2, 0, sizeof(id2));
    if (rank == 0) r = g_rccl.GetUniqueId(&id2);
    void* dbuf = nullptr;
    hipError_t he = hipMalloc(&dbuf, sizeof(id2));
    if (he == hipSuccess) he = hipMemcpy(dbuf, &id2, sizeof(id2), hipMemcpyHostToDevice);
    ncclResult_t rb = ncclInternalError;
    if (he == hipSuccess) rb = g_rccl.Broadcast(dbuf, dbuf, sizeof(id2), ncclChar, 0, c->comm[0], ctx->stream);
    if (he == hipSuccess && rb == ncclSuccess) he = hipStreamSynchronize(ctx->stream);
    if (he == hipSuccess && rb == ncclSuccess) he = hipMemcpy(&id2, dbuf, sizeof(id2), hipMemcpyDeviceToHost);
    if (dbuf) hipFree(dbuf);
    ncclResult_t r2 = ncclInternalError;
    if (he == hipSuccess && rb == ncclSuccess) r2 = g_rccl.CommInitRank(&c->comm[1], world, id2, rank);
    if (r2 != ncclSuccess) {
      set_error("second communicator (priority stream) failed: hip %s, broadcast %s, init %s", hipGetErrorString(he),
                g_rccl.GetErrorString(rb), g_rccl.GetErrorString(r2));
      g_rccl.CommDestroy(c->comm[0]);
      delete c;
      return TGP_E_HIP;
    }
  }
  for (auto& e : c->ev) {
    hipError_t he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (he != hipSuccess) {
      set_error("hipEventCreateWithFlags failed: %s", hipGetErrorString(he));
      tgp_comm_destroy(c);
      return TGP_E_HIP;
    }
  }
  *out = c;
  return TGP_OK;
}

int tgp_comm_destroy(tgp_comm* c) {
  if (!c) return TGP_OK;
  {
    // under the context's lock (advisor r5): no other thread may be enqueueing on these streams while they are drained and
    // the communicators go away.  The caller keeps the context alive until this returns (tinygp_amd/comm.py holds a
    // reference; BlockCyclicCholesky.close closes the communicator it created BEFORE its operations and their context).
    std::unique_lock<std::recursive_mutex> lk;
    if (c->ctx) {
      lk = std::unique_lock<std::recursive_mutex>(c->ctx->mu);
      hipSetDevice(c->ctx->device);
      for (hipStream_t q : {c->ctx->panel_stream, c->ctx->stream})
        if (q) hipStreamSynchronize(q);
    }
    for (int q = 1; q >= 0; --q)
      if (c->comm[q] && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm[q]);
    for (auto e : c->ev)
      if (e) hipEventDestroy(e);
  }
  delete c;
  return TGP_OK;
}

int tgp_comm_info(tgp_comm* c, int32_t* world, int32_t* rank) {
  TGP_ARG_CHECK(c != nullptr, "null communicator");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  return TGP_OK;
}

// In-place collectives on `count` elements of `dtype` at the device pointer `buf`, enqueued on stream `which`
// (0: the context's main stream, 1: its priority stream).  Asynchronous: they return once enqueued.
int tgp_comm_broadcast(tgp_comm* c, int which, void* buf, int64_t count, int dtype, int32_t root) {
  COMM_GUARD(c);
  TGP_ARG_CHECK((which == 0 || which == 1) && buf != nullptr && count >= 0 && root >= 0 && root < c->world,
                "broadcast: bad argument");
  if (count == 0) return TGP_OK;
  TGP_NCCL_TRY(g_rccl.Broadcast(buf, buf, size_t(count), nccl_dtype(dtype), root, comm_of(c, which), comm_stream(c, which)));
  return TGP_OK;
}

int tgp_comm_reduce(tgp_comm* c, int which, void* buf, int64_t count, int dtype, int32_t root) {
  COMM_GUARD(c);
  TGP_ARG_CHECK((which == 0 || which == 1) && buf != nullptr && count >= 0 && root >= 0 && root < c->world,
                "reduce: bad argument");
  if (count == 0) return TGP_OK;
  TGP_NCCL_TRY(g_rccl.Reduce(buf, buf, size_t(count), nccl_dtype(dtype), ncclSum, root, comm_of(c, which), comm_stream(c, which)));
  return TGP_OK;
}

// op: 0 sum, 1 min, 2 max
int tgp_comm_all_reduce(tgp_comm* c, int which, void* buf, int64_t count, int dtype, int32_t op) {
  COMM_GUARD(c);
  TGP_ARG_CHECK((which == 0 || which == 1) && buf != nullptr && count >= 0 && op >= 0 && op <= 2, "all_reduce: bad argument");
  if (count == 0) return TGP_OK;
  const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax);
  TGP_NCCL_TRY(g_rccl.AllReduce(buf, buf, size_t(count), nccl_dtype(dtype), rop, comm_of(c, which), comm_stream(c, which)));
  return TGP_OK;
}

// "everything enqueued so far on stream `which`" as a ticket another stream can wait for (no host block)
int tgp_comm_record(tgp_comm* c, int which, int64_t* ticket) {
  COMM_GUARD(c);
  TGP_ARG_CHECK((which == 0 || which == 1) && ticket != nullptr, "record: bad argument");
  const int64_t t = c->issued++;
  TGP_HIP_TRY(hipEventRecord(c->ev[t % tgp_comm::NEV], comm_stream(c, which)));
  *ticket = t;
  return TGP_OK;
}

int tgp_comm_wait(tgp_comm* c, int which, int64_t ticket) {
  COMM_GUARD(c);
  TGP_ARG_CHECK(which == 0 || which == 1, "wait: bad stream");
  TGP_ARG_CHECK(ticket >= 0 && ticket < c->issued && c->issued - ticket <= tgp_comm::NEV,
                "wait: ticket %lld is stale or was never issued (%lld issued, ring of %d)", (long long)ticket,
                (long long)c->issued, tgp_comm::NEV);
  TGP_HIP_TRY(hipStreamWaitEvent(comm_stream(c, which), c->ev[ticket % tgp_comm::NEV], 0));
  return TGP_OK;
}

// ---- stream-addressed transfers (no communicator needed): the driver's Python side owns plain device buffers ----
static hipStream_t ctx_stream(tgp_ctx* ctx, int which) { return which == 0 ? ctx->stream : ctx->panel_stream; }

int tgp_stream_h2d(tgp_ctx* ctx, int which, void* dst_dev, const void* src_host, int64_t bytes) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK((which == 0 || which == 1) && dst_dev && (src_host || bytes == 0) && bytes >= 0, "h2d: bad argument");
  if (bytes == 0) return TGP_OK;
  TGP_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, size_t(bytes), hipMemcpyHostToDevice, ctx_stream(ctx, which)));
  TGP_HIP_TRY(hipStreamSynchronize(ctx_stream(ctx, which)));  // (pageable source: returns when it has been consumed)
  return TGP_OK;
}

int tgp_stream_d2h(tgp_ctx* ctx, int which, void* dst_host, const void* src_dev, int64_t bytes) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK((which == 0 || which == 1) && (dst_host || bytes == 0) && src_dev && bytes >= 0, "d2h: bad argument");
  if (bytes == 0) return TGP_OK;
  TGP_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, size_t(bytes), hipMemcpyDeviceToHost, ctx_stream(ctx, which)));
  TGP_HIP_TRY(hipStreamSynchronize(ctx_stream(ctx, which)));
  return TGP_OK;
}

int tgp_stream_d2d(tgp_ctx* ctx, int which, void* dst_dev, const void* src_dev, int64_t bytes) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK((which == 0 || which == 1) && dst_dev && src_dev && bytes >= 0, "d2d: bad argument");
  if (bytes == 0) return TGP_OK;
  TGP_HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, size_t(bytes), hipMemcpyDeviceToDevice, ctx_stream(ctx, which)));
  return TGP_OK;
}

int tgp_stream_memset(tgp_ctx* ctx, int which, void* dst_dev, int byte, int64_t bytes) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK((which == 0 || which == 1) && dst_dev && bytes >= 0, "memset: bad argument");
  if (bytes == 0) return TGP_OK;
  TGP_HIP_TRY(hipMemsetAsync(dst_dev, byte, size_t(bytes), ctx_stream(ctx, which)));
  return TGP_OK;
}

int tgp_stream_sync(tgp_ctx* ctx, int which) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(which == 0 || which == 1, "sync: bad stream");
  TGP_HIP_TRY(hipStreamSynchronize(ctx_stream(ctx, which)));
  return TGP_OK;
}

}  // extern "C"

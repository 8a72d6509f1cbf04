// mvae_rccl.hip -- the gradient all-reduce of the data-parallel step on librccl DIRECTLY (C ABI: include/mvae_hip.h,
// "Flat all-reduce").  New functionality: the reference is single-device (SURVEY.md section 8e names this export).
//
// One communicator per process (one process per GPU).  ncclAllReduce is enqueued on the CALLER's stream, so the exchange
// is ordered with the step's launches like any other kernel and is captured into the step's HIP graphs natively: no
// torch ProcessGroupNCCL exists on this route, hence no watchdog thread whose event queries can invalidate a capture.
// librccl is opened at run time (dlopen; the path may be given, default: the one already mapped into the process by
// torch, else the system's): the library has no link-time dependency on it, and single-GPU use never touches it.
#include <dlfcn.h>

#include "mvae_common.hpp"

// The handful of librccl declarations this unit needs, stated locally (NCCL's stable public ABI: nccl.h / rccl.h) so that the
// library builds on a ROCm install without the RCCL development headers -- librccl itself is only opened at run time.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;                     // (non-zero values are only printed, through rccl's text)
typedef enum { ncclInt32 = 2, ncclFloat32 = 7 } ncclDataType_t;    // nccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ... ncclFloat32 7
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_fail(ncclResult_t r, const char* what) {  // "<call> failed: <rccl's text> (<code>)"
  static thread_local char text[160];
  snprintf(text, sizeof(text), "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return fail(MVAE_E_SYSTEM, "%s (%lld)", text, (long long)r);
}
}  // namespace

struct mvae_rccl {
  ncclComm_t comm;
  int rank, world;
};

extern "C" int mvae_rccl_load(const char* path) {
  if (g_rccl.handle) return 0;
  const char* tries[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* p : tries) {
    if (!p || !p[0]) continue;
    h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(MVAE_E_UNSUPPORTED, "librccl could not be opened%s (%lld)", "", 0);
#define MV_SYM(field, name)                                                          \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));          \
  if (!g_rccl.field) return fail(MVAE_E_UNSUPPORTED, "librccl lacks %s (%lld)", name, 0)
  MV_SYM(GetUniqueId, "ncclGetUniqueId");
  MV_SYM(CommInitRank, "ncclCommInitRank");
  MV_SYM(CommDestroy, "ncclCommDestroy");
  MV_SYM(CommAbort, "ncclCommAbort");
  MV_SYM(AllReduce, "ncclAllReduce");
  MV_SYM(Broadcast, "ncclBroadcast");
  MV_SYM(ReduceScatter, "ncclReduceScatter");
  MV_SYM(AllGather, "ncclAllGather");
  MV_SYM(GroupStart, "ncclGroupStart");
  MV_SYM(GroupEnd, "ncclGroupEnd");
  MV_SYM(GetErrorString, "ncclGetErrorString");
#undef MV_SYM
  g_rccl.handle = h;
  return 0;
}

extern "C" int mvae_rccl_unique_id(uint8_t id[MVAE_RCCL_ID_BYTES]) {
  if (!id) return fail(MVAE_E_BADARG, "null pointer%s", "");
  int rc = mvae_rccl_load(nullptr);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == MVAE_RCCL_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int mvae_rccl_create(const uint8_t id[MVAE_RCCL_ID_BYTES], int rank, int world, mvae_rccl** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(MVAE_E_BADARG, "bad rank / world%s", "");
  int rc = mvae_rccl_load(nullptr);
  if (rc) return rc;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm;
  ncclResult_t r = g_rccl.CommInitRank(&comm, world, u, rank);  // collective: every rank of the job calls it
  if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
  *out = new mvae_rccl{comm, rank, world};
  return 0;
}

extern "C" void mvae_rccl_destroy(mvae_rccl* c) {
  if (!c) return;
  if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

// buf[i] <- sum over ranks of buf[i], in place, float32, on `stream` (graph-capturable).  Every rank calls it with the
// same n in the same order.  Several calls may be bracketed by mvae_rccl_group(1) / (0) to be launched as one.
extern "C" int mvae_flat_allreduce(mvae_rccl* c, float* buf, int64_t n, void* stream) {
  if (!c || !buf || n < 0) return fail(MVAE_E_BADARG, "null pointer / negative count%s", "");
  if (n == 0) return 0;
  ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
  return 0;
}

// The two halves of the all-reduce, for the SHARDED optimizer (mvae_step_optimizer_slice between them), both in place on the
// flat buffer: after the reduce-scatter rank r holds the sums of floats [r n / world, (r + 1) n / world) in that range of
// `buf` (the rest of `buf` is unspecified); the all-gather hands every rank's range of `buf` to everybody.  n must be a
// multiple of the world size (MVAE_E_UNSUPPORTED otherwise: the caller keeps the replicated optimizer).
extern "C" int mvae_flat_reduce_scatter(mvae_rccl* c, float* buf, int64_t n, void* stream) {
  if (!c || !buf || n < 0) return fail(MVAE_E_BADARG, "null pointer / negative count%s", "");
  if (n % c->world) return fail(MVAE_E_UNSUPPORTED, "reduce-scatter: the count must be a multiple of the world size%s", "");
  if (n == 0) return 0;
  const size_t cnt = (size_t)(n / c->world);
  ncclResult_t r = g_rccl.ReduceScatter(buf, buf + (size_t)c->rank * cnt, cnt, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail(r, "ncclReduceScatter");
  return 0;
}

extern "C" int mvae_flat_allgather(mvae_rccl* c, float* buf, int64_t n, void* stream) {
  if (!c || !buf || n < 0) return fail(MVAE_E_BADARG, "null pointer / negative count%s", "");
  if (n % c->world) return fail(MVAE_E_UNSUPPORTED, "all-gather: the count must be a multiple of the world size%s", "");
  if (n == 0) return 0;
  const size_t cnt = (size_t)(n / c->world);
  ncclResult_t r = g_rccl.AllGather(buf + (size_t)c->rank * cnt, buf, cnt, ncclFloat32, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
  return 0;
}

// buf of rank `root` -> every rank (the initial parameter / optimizer-state broadcast), float32 or int32 words
extern "C" int mvae_flat_broadcast(mvae_rccl* c, void* buf, int64_t n_words, int root, void* stream) {
  if (!c || !buf || n_words < 0 || root < 0 || root >= c->world) return fail(MVAE_E_BADARG, "bad argument%s", "");
  if (n_words == 0) return 0;
  ncclResult_t r = g_rccl.Broadcast(buf, buf, (size_t)n_words, ncclInt32, root, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail(r, "ncclBroadcast");
  return 0;
}

extern "C" int mvae_rccl_group(int begin) {
  int rc = mvae_rccl_load(nullptr);
  if (rc) return rc;
  ncclResult_t r = begin ? g_rccl.GroupStart() : g_rccl.GroupEnd();
  if (r != ncclSuccess) return rccl_fail(r, begin ? "ncclGroupStart" : "ncclGroupEnd");
  return 0;
}

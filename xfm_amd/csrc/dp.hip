// Data-parallel exchange over RCCL (xGMI) behind the C-ABI: xfm_dp_* (include/xfm_hip.h).  What the reference does through
// torch.distributed -- the gradient all-reduce of DistributedDataParallel (accelerators/ddp_accelerator.py:34-98), the feature AllGather of
// the contrastive loss (models/xfm.py:17-50), the parameter broadcast at set-up -- for a host that has no torch.distributed: one
// communicator per process (one process per GPU), every call enqueued on the caller's HIP stream, nothing allocated or synchronised here.
// Included by capi.hip.
//
// librccl is resolved at the first xfm_dp_* call with dlopen / dlsym, not at link time: libxfm_hip.so must load on a box without RCCL
// (the build check runs without a GPU), and inside a PyTorch process the loader hands back the librccl.so.1 torch has already mapped
// -- one RCCL per process, whoever asked first.
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace xfm_dp {
struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static Api& api() {
  static Api a;   // (initialised once, thread-safe by the language; a failed load is remembered and reported on every call)
  static const bool once = [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.handle != nullptr) break;
    }
    if (a.handle == nullptr) return false;
#define XFM_DP_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, sym))
    XFM_DP_SYM(GetUniqueId, "ncclGetUniqueId");
    XFM_DP_SYM(CommInitRank, "ncclCommInitRank");
    XFM_DP_SYM(CommDestroy, "ncclCommDestroy");
    XFM_DP_SYM(AllReduce, "ncclAllReduce");
    XFM_DP_SYM(AllGather, "ncclAllGather");
    XFM_DP_SYM(Broadcast, "ncclBroadcast");
    XFM_DP_SYM(GetErrorString, "ncclGetErrorString");
#undef XFM_DP_SYM
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.Broadcast && a.GetErrorString;
    return a.ok;
  }();
  (void)once;
  return a;
}
static int check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return XFM_OK;
  xfm_set_error("%s: RCCL error %d (%s)", what, (int)r, api().GetErrorString ? api().GetErrorString(r) : "?");
  return XFM_E_LAUNCH;
}
static bool dtype_of(int dtype, ncclDataType_t& t) {
  if (dtype == XFM_DP_F32) { t = ncclFloat32; return true; }
  if (dtype == XFM_DP_BF16) { t = ncclBfloat16; return true; }
  if (dtype == XFM_DP_I32) { t = ncclInt32; return true; }
  return false;
}
}  // namespace xfm_dp

#define XFM_DP_READY()                                                                                                  \
  do {                                                                                                                  \
    if (!xfm_dp::api().ok) {                                                                                            \
      xfm_set_error("xfm_dp: librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");          \
      return XFM_E_UNSUPPORTED;                                                                                         \
    }                                                                                                                   \
  } while (0)

int xfm_dp_unique_id_impl(void* id) {
  XFM_DP_READY();
  XFM_REQUIRE(id != nullptr, "xfm_dp_unique_id: NULL buffer");
  static_assert(sizeof(ncclUniqueId) == XFM_DP_ID_BYTES, "XFM_DP_ID_BYTES is RCCL's unique-id size");
  return xfm_dp::check(xfm_dp::api().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id)), "xfm_dp_unique_id");
}
int xfm_dp_init_impl(const void* id, int rank, int world, void** comm) {
  XFM_DP_READY();
  XFM_REQUIRE(id != nullptr && comm != nullptr && world >= 1 && rank >= 0 && rank < world, "xfm_dp_init: bad arguments (rank %d of %d)", rank, world);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  const int rc = xfm_dp::check(xfm_dp::api().CommInitRank(&c, world, uid, rank), "xfm_dp_init");
  *comm = rc == XFM_OK ? reinterpret_cast<void*>(c) : nullptr;
  return rc;
}
int xfm_dp_finalize_impl(void* comm) {
  XFM_DP_READY();
  XFM_REQUIRE(comm != nullptr, "xfm_dp_finalize: NULL communicator");
  return xfm_dp::check(xfm_dp::api().CommDestroy(reinterpret_cast<ncclComm_t>(comm)), "xfm_dp_finalize");
}
int xfm_dp_bucket_allreduce_impl(void* comm, void* buf, long n, int dtype, int op, hipStream_t st) {
  XFM_DP_READY();
  ncclDataType_t t;
  XFM_REQUIRE(comm != nullptr && (n == 0 || buf != nullptr) && n >= 0 && xfm_dp::dtype_of(dtype, t), "xfm_dp_bucket_allreduce: bad arguments");
  XFM_REQUIRE(op == XFM_DP_SUM || op == XFM_DP_AVG || op == XFM_DP_MAX, "xfm_dp_bucket_allreduce: op %d", op);
  if (n == 0) return XFM_OK;
  const ncclRedOp_t o = op == XFM_DP_SUM ? ncclSum : (op == XFM_DP_AVG ? ncclAvg : ncclMax);
  return xfm_dp::check(xfm_dp::api().AllReduce(buf, buf, (size_t)n, t, o, reinterpret_cast<ncclComm_t>(comm), st), "xfm_dp_bucket_allreduce");
}
int xfm_dp_allgather_impl(void* comm, const void* send, void* recv, long n_per_rank, int dtype, hipStream_t st) {
  XFM_DP_READY();
  ncclDataType_t t;
  XFM_REQUIRE(comm != nullptr && send != nullptr && recv != nullptr && n_per_rank > 0 && xfm_dp::dtype_of(dtype, t), "xfm_dp_allgather: bad arguments");
  return xfm_dp::check(xfm_dp::api().AllGather(send, recv, (size_t)n_per_rank, t, reinterpret_cast<ncclComm_t>(comm), st), "xfm_dp_allgather");
}
int xfm_dp_broadcast_impl(void* comm, void* buf, long n, int dtype, int root, hipStream_t st) {
  XFM_DP_READY();
  ncclDataType_t t;
  XFM_REQUIRE(comm != nullptr && (n == 0 || buf != nullptr) && n >= 0 && root >= 0 && xfm_dp::dtype_of(dtype, t), "xfm_dp_broadcast: bad arguments");
  if (n == 0) return XFM_OK;
  return xfm_dp::check(xfm_dp::api().Broadcast(buf, buf, (size_t)n, t, root, reinterpret_cast<ncclComm_t>(comm), st), "xfm_dp_broadcast");
}

// RCCL over xGMI, called straight from the executor: one communicator per process (= per GPU), all-reduce(sum) of
// the Sync-BN statistics vectors and of the flat gradient buffer enqueued on the stream the kernels run on -- no
// Python in the loop (≈310 statistics exchanges per Mean-Teacher step), no extra stream hop.
//   reference: the thread-based master/slave exchange of sync_batchnorm/comm.py + nn.DataParallel's gradient
//   reduction (pixelssl/nn/func.py:54-62).
// librccl is resolved at run time (dlopen by SONAME: the copy PyTorch already loaded is reused), so libpixelhip.so
// itself has no link-time dependency on it; every entry point fails loudly when it is unavailable.
#include <dlfcn.h>
#include <cstring>
#include <new>

#include "common.h"

namespace {

typedef void* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclFloat = 7, kNcclSum = 0;

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) return r;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
  return r;
}

}  // namespace

struct pxl_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

#define PXL_CHECK_RCCL(expr)                                                                                   \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != 0) return pxl_set_error(PXL_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(_r));         \
  } while (0)

extern "C" int pxl_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int pxl_comm_unique_id(void* id128) {
  PXL_REQUIRE(id128, "comm_unique_id: null argument");
  if (!rccl().ok) return pxl_set_error(PXL_ERR_UNSUPPORTED, "comm: librccl could not be loaded");
  ncclUniqueId id;
  PXL_CHECK_RCCL(rccl().GetUniqueId(&id));
  std::memcpy(id128, id.internal, 128);
  return PXL_OK;
}

extern "C" int pxl_comm_init(const void* id128, int rank, int world, pxl_comm** out) {
  PXL_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad argument");
  if (!rccl().ok) return pxl_set_error(PXL_ERR_UNSUPPORTED, "comm: librccl could not be loaded");
  pxl_comm* c = new (std::nothrow) pxl_comm();
  PXL_REQUIRE(c != nullptr, "comm_init: out of host memory");
  ncclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  const ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
  if (r != 0) {
    delete c;
    return pxl_set_error(PXL_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
  }
  c->rank = rank;
  c->world = world;
  *out = c;
  return PXL_OK;
}

extern "C" void pxl_comm_destroy(pxl_comm* c) {
  if (!c) return;
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  delete c;
}

extern "C" int pxl_comm_allreduce_sum(pxl_comm* c, float* buf, long n, void* stream) {
  PXL_REQUIRE(c && c->comm && buf && n > 0, "comm_allreduce_sum: bad argument");
  PXL_CHECK_RCCL(rccl().AllReduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, c->comm, reinterpret_cast<hipStream_t>(stream)));
  return PXL_OK;
}

// pxl_allreduce_fn-compatible entry: pxl_net_set_sync(net, pxl_comm_allreduce_hook, comm, world)
extern "C" int pxl_comm_allreduce_hook(void* user, float* buf, int n, void* stream) {
  return pxl_comm_allreduce_sum(reinterpret_cast<pxl_comm*>(user), buf, (long)n, stream) == PXL_OK ? 0 : 1;
}

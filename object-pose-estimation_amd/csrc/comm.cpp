// comm.cpp — native RCCL path: one all-reduce of the 17 fp64 sums per ICP iteration.
//
// The reference is single-process and has no collective; this is the one exchange step of the
// sharded design: source points are sharded across ranks, the target index is
// replicated, and {n, Σs, Σt, Σ t sᵀ, Σd²} is summed over ranks so that every rank computes the
// same incremental transform redundantly (no broadcast).  136 bytes per iteration: latency-bound.
//
// RCCL is dlopen'ed so that (a) the library has no link-time RCCL dependency for single-GPU users
// and (b) inside a torch process we bind to the librccl that torch already loaded (same SONAME).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "ope_internal.hpp"

namespace ope {

namespace {
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl &rccl() {
  static Rccl r;
  if (r.h) return r;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return r;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce) r.h = h;
  return r;
}
}  // namespace

int comm_allreduce_sums(ope_ctx *ctx, double *d_sums, int count) {
  Rccl &r = rccl();
  if (!r.h || !ctx->nccl_comm) return set_err(ctx, OPE_ECOMM, "comm_allreduce_sums: communicator not initialised");
  ncclResult_t rc = r.AllReduce(d_sums, d_sums, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)ctx->nccl_comm, ctx->stream);
  if (rc != ncclSuccess)
    return set_err(ctx, OPE_ECOMM, std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
  return OPE_OK;
}

}  // namespace ope

using namespace ope;

extern "C" {

int ope_comm_get_unique_id(char id[OPE_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= OPE_COMM_ID_BYTES, "ncclUniqueId larger than OPE_COMM_ID_BYTES");
  Rccl &r = rccl();
  if (!id) return OPE_EINVAL;
  if (!r.h) return set_err(nullptr, OPE_ECOMM, "librccl.so.1 could not be loaded");
  ncclUniqueId uid;
  if (r.GetUniqueId(&uid) != ncclSuccess) return set_err(nullptr, OPE_ECOMM, "ncclGetUniqueId failed");
  std::memset(id, 0, OPE_COMM_ID_BYTES);
  std::memcpy(id, &uid, sizeof uid);
  return OPE_OK;
}

int ope_comm_init_rank(ope_ctx *ctx, const char id[OPE_COMM_ID_BYTES], int nranks, int rank) {
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return set_err(ctx, OPE_EINVAL, "ope_comm_init_rank: bad argument");
  Rccl &r = rccl();
  if (!r.h) return set_err(ctx, OPE_ECOMM, "librccl.so.1 could not be loaded");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  ope_comm_destroy(ctx);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclComm_t comm;
  ncclResult_t rc = r.CommInitRank(&comm, nranks, uid, rank);
  if (rc != ncclSuccess)
    return set_err(ctx, OPE_ECOMM, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
  ctx->nccl_comm = comm;
  ctx->comm_nranks = nranks;
  ctx->comm_rank = rank;
  return OPE_OK;
}

int ope_comm_destroy(ope_ctx *ctx) {
  if (!ctx) return OPE_EINVAL;
  if (ctx->nccl_comm) {
    Rccl &r = rccl();
    if (r.h) r.CommDestroy((ncclComm_t)ctx->nccl_comm);
    ctx->nccl_comm = nullptr;
  }
  ctx->comm_nranks = 1;
  ctx->comm_rank = 0;
  return OPE_OK;
}

}  // extern "C"

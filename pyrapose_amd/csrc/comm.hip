// Gradient all-reduce on RCCL from INSIDE the library (pp_allreduce_bucket; 8e: per-image batch shards, weights
// replicated, SUM all-reduce of the flat gradient buffer in buckets on a stream the engine owns, overlapped with the backward).
// The reference has no collective path (bin/train.py:82-89 is a disabled multi_gpu_model branch): this is the engine's own layer.
//
// RCCL is reached through dlopen -- librccl.so is not a link-time dependency of libpyrapose_hip.so, a box without it still loads the
// library and only these entry points fail (PP_ERR_UNSUPPORTED).  In a Python process torch has usually mapped its own librccl.so
// already; dlopen by soname then returns THAT copy, so both layers share one RCCL.  One communicator = one rank of one job; the
// 128-byte unique id travels between the ranks by whatever channel the host layer has (pyrapose_amd/parallel.py: the process
// group's object broadcast, or a file).
#include <dlfcn.h>
#include <stdlib.h>

#include "pp_internal.h"

namespace {

// the slice of rccl.h this file uses (ABI of RCCL 2.x: ncclResult_t / ncclDataType_t / ncclRedOp_t are ints)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclInt32 = 2, kNcclFloat32 = 7 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char why[256] = {0};
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  // (the soname first: in a process that has torch's bundled RCCL mapped, dlopen by soname returns THAT copy, not a second library)
  const char* names[] = {getenv("PP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !n[0]) continue;
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
    snprintf(r.why, sizeof(r.why), "%s", dlerror());
  }
  if (!r.handle) return nullptr;
  r.GetUniqueId = (int (*)(ncclUniqueId*))dlsym(r.handle, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(r.handle, "ncclCommInitRank");
  r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.handle, "ncclCommDestroy");
  r.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(r.handle, "ncclAllReduce");
  r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
    snprintf(r.why, sizeof(r.why), "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
    dlclose(r.handle);
    r.handle = nullptr;
    return nullptr;
  }
  return &r;
}

const char* why_not() { return "librccl.so could not be loaded (PP_RCCL_LIB names another path)"; }

}  // namespace

struct pp_comm {
  ncclComm_t comm;
  int world, rank;
};

#define PP_NCCL(ctx, r, call)                                                                                       \
  do {                                                                                                              \
    const int rc__ = (call);                                                                                        \
    if (rc__ != kNcclSuccess)                                                                                       \
      return pp_fail(ctx, PP_ERR_COMM, "%s: %s", #call, (r)->GetErrorString ? (r)->GetErrorString(rc__) : "RCCL error"); \
  } while (0)

extern "C" int pp_comm_available(void) { return rccl() != nullptr; }

extern "C" int pp_comm_unique_id(pp_ctx* ctx, void* id128) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, id128 != nullptr, PP_ERR_ARG, "pp_comm_unique_id: null id buffer (128 bytes)");
  Rccl* r = rccl();
  PP_CHECK_ARG(ctx, r != nullptr, PP_ERR_UNSUPPORTED, "pp_comm_unique_id: %s", why_not());
  ncclUniqueId id;
  PP_NCCL(ctx, r, r->GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return PP_OK;
}

extern "C" int pp_comm_init(pp_ctx* ctx, int world, int rank, const void* id128, pp_comm** out) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, out && id128 && world >= 1 && rank >= 0 && rank < world, PP_ERR_ARG, "pp_comm_init: world %d rank %d", world, rank);
  *out = nullptr;
  Rccl* r = rccl();
  PP_CHECK_ARG(ctx, r != nullptr, PP_ERR_UNSUPPORTED, "pp_comm_init: %s", why_not());
  PP_HIP(ctx, hipSetDevice(ctx->device));  // (the communicator binds to the current device)
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  PP_NCCL(ctx, r, r->CommInitRank(&c, world, id, rank));
  pp_comm* p = (pp_comm*)calloc(1, sizeof(pp_comm));
  if (!p) {
    r->CommDestroy(c);
    return pp_fail(ctx, PP_ERR_ARG, "pp_comm_init: out of memory");
  }
  p->comm = c;
  p->world = world;
  p->rank = rank;
  *out = p;
  return PP_OK;
}

extern "C" int pp_comm_destroy(pp_comm* comm) {
  if (!comm) return PP_OK;
  Rccl* r = rccl();
  if (r && comm->comm) r->CommDestroy(comm->comm);
  free(comm);
  return PP_OK;
}

extern "C" int pp_allreduce_bucket(pp_ctx* ctx, pp_comm* comm, float* buf, size_t count) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, comm && comm->comm && (buf || count == 0), PP_ERR_ARG, "pp_allreduce_bucket: null communicator / buffer");
  if (count == 0) return PP_OK;
  Rccl* r = rccl();
  PP_CHECK_ARG(ctx, r != nullptr, PP_ERR_UNSUPPORTED, "pp_allreduce_bucket: %s", why_not());
  PP_NCCL(ctx, r, r->AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, comm->comm, ctx->stream));
  return PP_OK;
}

extern "C" int pp_allreduce_counts(pp_ctx* ctx, pp_comm* comm, int* counts, int n) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, comm && comm->comm && counts && n > 0, PP_ERR_ARG, "pp_allreduce_counts: bad arguments");
  Rccl* r = rccl();
  PP_CHECK_ARG(ctx, r != nullptr, PP_ERR_UNSUPPORTED, "pp_allreduce_counts: %s", why_not());
  PP_NCCL(ctx, r, r->AllReduce(counts, counts, (size_t)n, kNcclInt32, kNcclSum, comm->comm, ctx->stream));
  return PP_OK;
}

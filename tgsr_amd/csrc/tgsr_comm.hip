// The gradient collective behind the C ABI (SURVEY.md 8b op list: `allreduce_flat (RCCL wrapper)`): sum over the ranks of a flat fp32
// buffer, in place, then x scale (1 / world for the mean) - what tgsr_amd.parallel.FlatGradBucket issues once (twice with the early
// range) per training step.  The reference has no collective at all (single process, trainer_objective.py:31); the default data-
// parallel path goes through torch.distributed's "nccl" backend = RCCL, which SURVEY 7.5 allows; this is the same collective
// without torch in the way, for a host that is not Python.
//
// librccl is opened lazily with dlopen: the library must load - and every other entry point work - on a box without RCCL; the
// tgsr_comm_* functions then return TGSR_EUNSUPPORTED.  No ncclCommInitRank happens behind the caller's back: the caller moves the
// 128-byte id from rank 0 to the other ranks by whatever channel it has (a file, MPI, torch.distributed's store).
#include "tgsr_common.h"

#include <dlfcn.h>
#include <cstring>

namespace tgsr {

struct RcclApi {
  void* so = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, const void*, int) = nullptr;     // (comm*, nranks, id BY VALUE: see comm_init), rank
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

static RcclApi* rccl() {
  // (a function-local static with an initialiser: C++11 runs it once, also under concurrent first calls)
  static RcclApi* const loaded = []() -> RcclApi* {
    static RcclApi api;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      api.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.so) break;
    }
    if (!api.so) return nullptr;
    api.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(api.so, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<int (*)(void**, int, const void*, int)>(dlsym(api.so, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(api.so, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(api.so, "ncclCommDestroy"));
    api.CommCount = reinterpret_cast<int (*)(void*, int*)>(dlsym(api.so, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<int (*)(void*, int*)>(dlsym(api.so, "ncclCommUserRank"));
    api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(api.so, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) return nullptr;
    return &api;
  }();
  return loaded;
}

struct UniqueId { char bytes[128]; };                       // ncclUniqueId: passed BY VALUE to ncclCommInitRank
typedef int (*comm_init_by_value_t)(void**, int, UniqueId, int);

__global__ void scale_flat_kernel(float* __restrict__ x, int64_t n, float s) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s;
}

static int rccl_fail(RcclApi* a, int rc, const char* what) {
  return note_error(what, a->GetErrorString ? a->GetErrorString(rc) : "rccl error", rc);
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int tgsr_comm_unique_id(void* id128) {
  if (!id128) return TGSR_EINVAL;
  RcclApi* a = rccl();
  if (!a) return TGSR_EUNSUPPORTED;
  const int rc = a->GetUniqueId(id128);
  return rc ? rccl_fail(a, rc, "ncclGetUniqueId") : TGSR_OK;
}

extern "C" int tgsr_comm_init(void** comm, const void* id128, int rank, int world) {
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return TGSR_EINVAL;
  RcclApi* a = rccl();
  if (!a) return TGSR_EUNSUPPORTED;
  UniqueId id;
  memcpy(id.bytes, id128, sizeof(id.bytes));
  const int rc = reinterpret_cast<comm_init_by_value_t>(a->CommInitRank)(comm, world, id, rank);
  return rc ? rccl_fail(a, rc, "ncclCommInitRank") : TGSR_OK;
}

extern "C" int tgsr_allreduce_flat(void* comm, float* buf, int64_t n, float scale, void* stream) {
  if (!comm || !buf || n < 1) return TGSR_EINVAL;
  RcclApi* a = rccl();
  if (!a) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int rc = a->AllReduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, s);
  if (rc) return rccl_fail(a, rc, "ncclAllReduce");
  if (scale != 1.f) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(scale_flat_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, buf, n, scale);
    return note_launch(hipGetLastError(), "scale_flat_kernel");
  }
  return TGSR_OK;
}

extern "C" int tgsr_comm_count(void* comm, int* world, int* rank) {
  if (!comm || !world || !rank) return TGSR_EINVAL;
  RcclApi* a = rccl();
  if (!a || !a->CommCount || !a->CommUserRank) return TGSR_EUNSUPPORTED;
  int rc = a->CommCount(comm, world);
  if (rc) return rccl_fail(a, rc, "ncclCommCount");
  rc = a->CommUserRank(comm, rank);
  return rc ? rccl_fail(a, rc, "ncclCommUserRank") : TGSR_OK;
}

extern "C" int tgsr_comm_destroy(void* comm) {
  if (!comm) return TGSR_EINVAL;
  RcclApi* a = rccl();
  if (!a) return TGSR_EUNSUPPORTED;
  const int rc = a->CommDestroy(comm);
  return rc ? rccl_fail(a, rc, "ncclCommDestroy") : TGSR_OK;
}

// RCCL in the C ABI (flk_allreduce_sum_f32): the data-parallel attack's only exchange -- one sum all-reduce of the
// (T*3 + 3)-float payload (or of the dense gradient) over xGMI -- issued on the CALLER's stream, so it is ordered with the kernels
// around it without any event hop.  RCCL is loaded at run time (dlopen of the librccl already in the process -- torch's -- or of
// the system one), so the library has no link-time dependency on it and a process that never creates a communicator never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "flk_internal.h"

namespace {
typedef struct { char internal[128]; } UniqueId;      // ncclUniqueId (rccl.h): 128 opaque bytes, passed by value
typedef void* Comm;                                   // ncclComm_t
constexpr int kFloat32 = 7, kSum = 0;                 // ncclFloat32, ncclSum (rccl.h:448,466)
struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

int load_api() {
  if (g_api.lib) return FLK_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  FLK_REQUIRE(h, "flk_comm: cannot load librccl (%s)", dlerror());
  Api a;
  a.lib = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  FLK_REQUIRE(a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy, "flk_comm: librccl lacks an expected symbol");
  g_api = a;
  return FLK_OK;
}
const char* errstr(int rc) { return g_api.GetErrorString ? g_api.GetErrorString(rc) : "rccl error"; }
}  // namespace

struct flk_comm {
  Comm comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" int flk_comm_unique_id(void* id128_out) {
  FLK_REQUIRE(id128_out, "flk_comm_unique_id: null argument");
  int rc = load_api();
  if (rc) return rc;
  UniqueId id;
  const int nr = g_api.GetUniqueId(&id);
  FLK_REQUIRE(nr == 0, "ncclGetUniqueId: %s", errstr(nr));
  memcpy(id128_out, &id, sizeof id);
  return FLK_OK;
}

extern "C" int flk_comm_create(const void* id128, int rank, int world, int device, flk_comm** out) {
  FLK_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "flk_comm_create: bad argument (rank %d of %d)", rank, world);
  int rc = load_api();
  if (rc) return rc;
  FLK_CHECK_HIP(hipSetDevice(device));
  UniqueId id;
  memcpy(&id, id128, sizeof id);
  flk_comm* c = new flk_comm;
  c->rank = rank; c->world = world; c->device = device;
  const int nr = g_api.CommInitRank(&c->comm, world, id, rank);
  if (nr != 0) { delete c; flk_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, errstr(nr)); return FLK_EHIP; }
  *out = c;
  return FLK_OK;
}

extern "C" int flk_allreduce_sum_f32(flk_comm* c, float* buf, int64_t n, void* stream) {
  FLK_REQUIRE(c && c->comm && buf && n > 0, "flk_allreduce_sum_f32: bad argument");
  const int nr = g_api.AllReduce(buf, buf, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream);
  FLK_REQUIRE(nr == 0, "ncclAllReduce(%lld floats): %s", (long long)n, errstr(nr));
  return FLK_OK;
}

extern "C" int flk_comm_destroy(flk_comm* c) {
  if (!c) return FLK_OK;
  if (c->comm && g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
  delete c;
  return FLK_OK;
}

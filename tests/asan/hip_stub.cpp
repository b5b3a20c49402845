// Host stand-in for the HIP runtime, used ONLY by the AddressSanitizer build of libflicker_hip's HOST code
// (tests/test_asan_cpu.py; there is no GPU ASan on this pool, and no GPU in the build container).
// "Device" memory is ordinary heap memory, so every hipMemcpy / hipMemset the library issues -- weight uploads, table
// uploads, activation read-backs -- is an ASan-checked memcpy / memset with the library's own sizes; kernel launches
// are accepted and dropped (nothing runs on a device), streams and events are opaque tokens.  The library's host halves
// -- argument validation, weight packing, plan construction, tile / grid selection, the launch wrappers -- run for real.
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

static std::atomic<long> g_live_allocs{0}, g_launches{0};
extern "C" long flk_stub_live_allocs() { return g_live_allocs.load(); }
extern "C" long flk_stub_launches() { return g_launches.load(); }

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); if (!*p) return hipErrorOutOfMemory; ++g_live_allocs; return hipSuccess; }
hipError_t hipFree(void* p) { if (p) { free(p); --g_live_allocs; } return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }      // (multiprocessor count: conv_pc.hip sizes its persistent grid by it)
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t lds, hipStream_t) {
  // the limits a real launch would be rejected for
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024) return hipErrorInvalidConfiguration;
  if (grid.y > 65535 || grid.z > 65535 || lds > 160 * 1024) return hipErrorInvalidValue;
  ++g_launches;
  return hipSuccess;
}
// hipExtLaunchKernelGGL (fork / join events riding on kernels): same checks, the events are ignored
hipError_t hipExtLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t lds, hipStream_t s, hipEvent_t, hipEvent_t, int) {
  return hipLaunchKernel(f, grid, block, args, lds, s);
}
// kernel-launch plumbing emitted by clang for <<< >>> / hipLaunchKernelGGL and the module constructor
struct CallCfg { dim3 g, b; size_t lds; hipStream_t s; };
static thread_local CallCfg g_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t lds, hipStream_t s) { g_cfg = {g, b, lds, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* lds, hipStream_t* s) { *g = g_cfg.g; *b = g_cfg.b; *lds = g_cfg.lds; *s = g_cfg.s; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
}

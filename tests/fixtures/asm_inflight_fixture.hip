// Fixture for tools/audit_asm_loads.py (tests/test_asm_audit_cpu.py): two tiny kernels with a hand-issued asynchronous load.
// The audit must PASS the first and FLAG the second.  Never launched; compiled to assembly only.
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// correct: the destination is only touched after the wait statement that names it
extern "C" __global__ void conv_igemm_kernel_fixture_ok(const u32x4* src, u32x4* dst, int n) {
  const u32x4* p = src + threadIdx.x;
  u32x4 q;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
  int acc = 0;
  for (int i = 0; i < n; ++i) acc += i * (int)threadIdx.x;          // independent work while the load is in flight
  asm volatile("s_waitcnt vmcnt(0) ; release %0" : "+v"(q) : : "memory");
  q.x += (unsigned)acc;
  dst[threadIdx.x] = q;
}

// the hazard of cdna_hip_programming.md 5.7 item 1: hipcc believes the asm statement has already written `q`, so it is free to
// read (copy, store, reuse) the register BEFORE the wait -- here the value is consumed between the load and its release
extern "C" __global__ void conv_igemm_kernel_fixture_bad(const u32x4* src, u32x4* dst, int n) {
  const u32x4* p = src + threadIdx.x;
  u32x4 q;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
  u32x4 early = q;                                                   // compiler-generated read of an in-flight register
  early.x += (unsigned)n;
  asm volatile("s_waitcnt vmcnt(0) ; release %0" : "+v"(q) : : "memory");
  q.y += early.x;
  dst[threadIdx.x] = q;
}

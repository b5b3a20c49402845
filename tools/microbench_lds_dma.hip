// Microbenchmark (gfx950): issue rate and per-CU throughput of the LDS-DMA loads the kernels of this library stream their operands with
// (global_load_lds_dwordx4, new on gfx950, against the 4-byte global_load_lds_dword), source resident in L2, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/microbench_lds_dma tools/microbench_lds_dma.hip && /tmp/microbench_lds_dma
// Measured on MI355X (round 3): dwordx4 from ONE wave 17 ns (41 cycles) per instruction = 60 B/ns per CU, saturating at 130 B/ns per CU
// (54 B/clk, the L1 rate) from four waves on -- 33 TB/s over the chip, far above HBM; the 4-byte form stops at 18.7 B/ns per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ inline void glds16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ inline void glds4(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(1024) void k(const char* src, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 4096;
  const char* p = src + ((blockIdx.x * 16 + wave) % 64) * 16384 + lane * 16;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const char* q = p + ((it * 4 + u) & 15) * 1024;
      if (MODE == 0) glds16(q, lds0 + u * 1024);
      else glds4(q, lds0 + u * 256);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc = *(float*)(smem + threadIdx.x * 4);
  if (acc == 12345.f) out[0] = acc;
}
template <int MODE> void run(int nw, const char* src, float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  k<MODE><<<256, nw * 64, 65536>>>(src, 10, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, nw * 64, 65536>>>(src, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 4;                       // per wave
  const double bytes_per = MODE == 2 ? 256.0 : 1024.0;
  printf("%s waves/CU %2d: %.1f ns per instruction per wave, %.1f B/ns per CU (%.2f TB/s chip)\n", name, nw, ms * 1e6 / instr,
         nw * instr * bytes_per / (ms * 1e6), 256.0 * nw * instr * bytes_per / (ms * 1e6) / 1e3);
}
int main() {
  char* src; float* out; hipMalloc(&src, 64 * 16384 + 65536); hipMalloc(&out, 64); hipMemset(src, 1, 64 * 16384 + 65536);
  for (int nw : {1, 2, 4, 8, 16}) run<0>(nw, src, out, "global_load_lds_dwordx4");
  for (int nw : {1, 2, 4, 8, 16}) run<2>(nw, src, out, "global_load_lds_dword  ");
  return 0;
}

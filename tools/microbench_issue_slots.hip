// how many non-MFMA instructions per K step does the ring-loop skeleton tolerate?  16 MFMAs 16x16x32 + 8 ds_read_b128 + barrier per step, three
// workgroups per CU, plus EXTRA independent v_add_u32 (VALU) and SX extra s_add (SALU) per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int EXTRA, int LDSW>
__global__ __launch_bounds__(256, 2) void k(const unsigned* seed, int steps, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 48 * 1024 / 4; i += 256) ((unsigned*)smem)[i] = (seed[(i * 7 + blockIdx.x) & 4095] & 0x7fff7fffu) | 0x30003000u;
  __syncthreads();
  int off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) off[i] = ((tid >> 6) * 4096 + i * 1024 + lane * 16) & (32768 - 1);
  f32x4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[f][i] = f32x4{0, 0, 0, 0};
  int tap = 0;
  unsigned x[8] = {1u + tid, 2u, 3u, 4u, 5u, 6u, 7u, 8u};
  for (int s = 0; s < steps; ++s) {
    if (LDSW) { *(uint4*)(smem + 40960 + ((s & 1) * 4096) + tid * 16) = make_uint4(x[0], x[1], x[2], x[3]); }
    __syncthreads();
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(smem + ((off[i] + tap) & 32767));
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(smem + 32768 + ((off[4 + i] + (s & 1) * 4096) & 8191));
    tap = (tap + 16) & 4095;
#pragma unroll
    for (int e = 0; e < EXTRA; ++e) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[e & 7]) : "v"(x[(e + 1) & 7]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f], b[i], acc[f][i], 0, 0, 0);
  }
  float r = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc[f][i][0] + acc[f][i][3];
  if (r == 12345.f || x[0] + x[3] == 77u) out[0] = r;
}
template <int EXTRA, int LDSW> double run(const unsigned* seed, float* out, int wgs, int steps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<EXTRA, LDSW>, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024);
  k<EXTRA, LDSW><<<wgs, 256, 52 * 1024>>>(seed, 64, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<EXTRA, LDSW><<<wgs, 256, 52 * 1024>>>(seed, steps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  unsigned* seed; float* out; hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 64);
  unsigned h[4096]; srand(1); for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand() * 2654435761u; hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  const int wgs = 256 * 3 * 4, steps = 2000;
  const double fl = (double)wgs * 4 * steps * 64.0 * 64 * 32 * 2;
  for (int rep = 0; rep < 2; ++rep) {
    printf("extra VALU per step   0: %.0f   8: %.0f  16: %.0f  32: %.0f  64: %.0f TFLOP/s | with a ring write per step: 0: %.0f  16: %.0f  32: %.0f\n",
           fl / run<0, 0>(seed, out, wgs, steps) / 1e9, fl / run<8, 0>(seed, out, wgs, steps) / 1e9, fl / run<16, 0>(seed, out, wgs, steps) / 1e9,
           fl / run<32, 0>(seed, out, wgs, steps) / 1e9, fl / run<64, 0>(seed, out, wgs, steps) / 1e9,
           fl / run<0, 1>(seed, out, wgs, steps) / 1e9, fl / run<16, 1>(seed, out, wgs, steps) / 1e9, fl / run<32, 1>(seed, out, wgs, steps) / 1e9);
  }
  return 0;
}

// Microbenchmark (gfx950): does the MFMA SHAPE matter in the issue mix of conv_igemm_kernel's ring loop?  One K step of that loop is 14
// non-MFMA vector instructions (8 ds_read_b128 of fragments, 6 address adds; the ring write and weight load are left out) + a workgroup barrier
// per 16 v_mfma_f32_16x16x32_bf16; a 16x16x32 MFMA holds the SIMD's vector issue port for 8 of its 16 cycles, a 32x32x16 for 8 of its 32
// (MI355X_MICROARCH.md), so the same flops as 8 MFMAs of 32x32x16 leave twice the issue slots -- at a clock that the chip holds ~13 % lower
// for that shape in bare loops.  Both loops here: 256 threads, three workgroups per CU (LDS-limited, like the nf = 4 ring kernels), random
// bf16 operands in LDS, fragments re-read every step.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/microbench_mfma_shape tools/microbench_mfma_shape.hip && /tmp/microbench_mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>      // 0: 16 x mfma 16x16x32 per step, 1: 8 x mfma 32x32x16 per step (same flops: 64 x 64 x 32 per wave and step)
__global__ __launch_bounds__(256, 2) void k(const unsigned* seed, int steps, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 48 KiB: three workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 48 * 1024 / 4; i += 256) ((unsigned*)smem)[i] = (seed[(i * 7 + blockIdx.x) & 4095] & 0x7fff7fffu) | 0x30003000u;   // random bf16 of moderate size
  __syncthreads();
  int off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) off[i] = ((tid >> 6) * 4096 + i * 1024 + lane * 16) & (32768 - 1);
  if (SHAPE == 0) {
    f32x4 acc[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[f][i] = f32x4{0, 0, 0, 0};
    int tap = 0;
    for (int s = 0; s < steps; ++s) {
      __syncthreads();
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(smem + ((off[i] + tap) & 32767));
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(smem + 32768 + ((off[4 + i] + (s & 1) * 4096) & 8191));
      tap = (tap + 16) & 4095;
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
    if (r == 12345.f) out[0] = r;
  } else {
    f32x16 acc[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][i][e] = 0;
    int tap = 0;
    for (int s = 0; s < steps; ++s) {
      __syncthreads();
      bf16x8 a[4], b[4];            // [row block 0/1][K half 0/1]: the same 8 x 16-byte fragment reads per step
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(smem + ((off[i] + tap) & 32767));
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(smem + 32768 + ((off[4 + i] + (s & 1) * 4096) & 8191));
      tap = (tap + 16) & 4095;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[f][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f * 2 + kh], b[i * 2 + kh], acc[f][i], 0, 0, 0);
    }
    float r = 0;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int i = 0; i < 2; ++i) r += acc[f][i][0] + acc[f][i][15];
    if (r == 12345.f) out[0] = r;
  }
}

template <int SHAPE> double run(const unsigned* seed, float* out, int wgs, int steps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
  k<SHAPE><<<wgs, 256, 48 * 1024>>>(seed, 64, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<SHAPE><<<wgs, 256, 48 * 1024>>>(seed, steps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  unsigned* seed; float* out; hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 64);
  unsigned h[4096]; srand(1); for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand() * 2654435761u; hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  const int wgs = 256 * 3 * 4, steps = 2000;                      // four rounds of three workgroups per CU
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = run<0>(seed, out, wgs, steps), t1 = run<1>(seed, out, wgs, steps);
    const double fl = (double)wgs * 4 * steps * 64.0 * 64 * 32 * 2;
    printf("16 x 16x16x32 per step: %.3f ms = %.0f TFLOP/s | 8 x 32x32x16 per step: %.3f ms = %.0f TFLOP/s  (ratio %.3f)\n", t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t0 / t1);
  }
  return 0;
}

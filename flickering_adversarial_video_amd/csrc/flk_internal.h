// Internal declarations shared by the HIP translation units of libflicker_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/flicker_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing -------------------------------------------------------------------------
void flk_set_error(const char* fmt, ...);
#define FLK_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      flk_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return FLK_EHIP;                                                                   \
    }                                                                                    \
  } while (0)
#define FLK_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      flk_set_error(__VA_ARGS__);       \
      return FLK_EINVAL;                \
    }                                   \
  } while (0)

// Every kernel launch of the library goes through FLK_LAUNCH_KERNEL.  When the plan (net.cpp) has armed flk_stop_event -- the launch is
// the last one a stream does before another stream waits for it -- the kernel is launched with that event as its STOP event
// (hipExtLaunchKernelGGL: the event is bound to the kernel's own completion signal), so the fork / join needs no separate
// hipEventRecord marker packet behind the kernel.  Armed by the plan for single-launch operators only; the launch disarms it.
extern thread_local hipEvent_t flk_stop_event;
extern thread_local unsigned flk_launch_count;      // kernel launches of this thread (the plan counts the launches of each operator)
#define FLK_LAUNCH_KERNEL(kernel, grid, block, lds, stream, ...)                                         \
  do {                                                                                                   \
    ++flk_launch_count;                                                                                  \
    hipEvent_t _flk_ev = flk_stop_event;                                                                 \
    if (_flk_ev) {                                                                                       \
      flk_stop_event = nullptr;                                                                          \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, _flk_ev, 0, __VA_ARGS__);         \
    } else {                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    }                                                                                                    \
  } while (0)

// name of the kernel the last convolution-class launch of this thread went to (per-launch profile: flk_net_profile_read's "kernel")
extern thread_local const char* flk_last_kernel_tag;

static inline int flk_esize(int dtype) { return dtype == FLK_BF16 ? 2 : 4; }

// experiment knobs of the timing-only builds (-DFLK_ABLATE / -DSF_ABLATE: kernels with phases switched off, WRONG results; the shipped
// library contains none of them -- tests/test_abi.py): read here and nowhere else
static inline int flk_ablate_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel and one process may drive several GPUs:
// `done` is the call site's `static bool done[FLK_MAX_DEVICES]`, indexed by the current device (set once per device, always
// set for a device id outside the table).
constexpr int FLK_MAX_DEVICES = 16;
static inline int flk_raise_lds_limit(const void* kernel, int bytes, bool* done) {
  int dev = 0;
  FLK_CHECK_HIP(hipGetDevice(&dev));
  const bool tracked = dev >= 0 && dev < FLK_MAX_DEVICES;
  if (tracked && done[dev]) return FLK_OK;
  FLK_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (tracked) done[dev] = true;
  return FLK_OK;
}

// ---- packed convolution weights ---------------------------------------------------------------
// Layout on device: [nslab][ntaps][cout_frags][64 lanes][EPL elems] where EPL = 16 B / elem size,
// a slab = 4*EPL input channels (one 64-byte run per position) and fragment F = ntile*nf + f of
// lane l=(q=l>>4, m=l&15) holds, for j < EPL,
//     W[tap][cin = slab*4*EPL + q*EPL + j][cout = ntile*16*nf + G*4*EPL + (m>>2)*EPL + (f % (EPL/4))*4 + (m&3)],
//     G = f / (EPL/4)
// (zero outside cin/cout).  With this permutation accumulator lane group q' = m>>2 ... of store group G owns EPL
// consecutive output channels (one 16-byte access) and the four lane groups of a wave cover 64 contiguous bytes of a
// position in ONE instruction.
struct flk_conv_weights {
  void* dev = nullptr;
  int kt = 0, kh = 0, kw = 0, cin = 0, cout = 0;   // of THIS operator (after optional transpose)
  int dtype = 0, nf = 0, nslab = 0, ntaps = 0, cout_frags = 0;
  int cin_split = 0, nslab1 = 0;   // two-segment K order: slabs [0,nslab1) = channels [0,cin_split)
  int stem4 = 0;                   // folded-stem K-step packing (49 steps of non-zero chunks; conv_igemm.hip mode 4)
  struct Tuned { int B, To, Ho, Wo, wn, da; };
  mutable std::vector<Tuned> tuned;   // cache of autotuned launch layouts per call geometry (conv_igemm.hip: flk_conv_set_autotune):
                                      // not part of the operator's value, filled by the tuning pass of the owning thread
  size_t bytes = 0;
};

int flk_conv_weights_create_impl(const float* w_dhwio, int kt, int kh, int kw, int cin, int cout,
                                 const float* row_scale, int transpose, int dtype, int nf, int cin_split,
                                 flk_conv_weights** out);

// choose the (Tt,Ht,Wt) tile for a logical output grid; returns rows used (<=256)
struct flk_tile {
  int Tt, Ht, Wt;
};
flk_tile flk_choose_tile(int To, int Ho, int Wo, int kt, int kh, int kw, int st, int sh, int sw, int max_rows = 256,
                         int max_halo = 1008);

constexpr int FLK_MAX_HALO = 1008; // halo positions per tile (LDS: 4 planes x 1008 x 16 B = 63 KiB: 2 workgroups per CU with a 16 KiB weight ring)
constexpr int FLK_ROWS = 256;      // output positions per workgroup tile

// Perturbation kernels (HBM-bound): apply + clip fused with the stem's space-to-depth staging, the
// delta-gradient reduction with the clip masks, and regulariser-gradient + Adam.
//
// Reference maths: kinetics_i3d_utils.py:100-142 (apply), :172-200 (regularisers/metrics),
// i3d_adversarial_main_single_video_npy.py:56-59,79-84 (loss, TF Adam); torch dialect model.py:80-101,
// 198-209, 868.  Closed forms: SURVEY Appendix C.
#include "flk_internal.h"

// Space-to-depth layouts (template FT = flk_apply_args.fold_t, 0 -> 2):
//   FT = 1: fold (h,w) only,   16 channels: (qh*2+qw)*3 + c, 12..15 zero          (VideoResNet stems)
//   FT = 2: fold (t,h,w),      32 channels: (qt*4+qh*2+qw)*3 + c, 24..31 zero
//   FT = 3: fold (t,h,w),      32 channels: (qt*2+qh)*8 + qw*3 + c, 6 and 7 of every 8 zero -- every 16-byte chunk holds
//           ONE (qt,qh) parity, so the structurally zero (tap, parity) chunks of the folded 7x7x7 stem can be skipped
//           (conv_igemm.hip mode 4)
template <int FT> struct S2D {
  static constexpr int F = FT == 1 ? 1 : 2;                 // frames folded into one position
  static constexpr int NCH = 16 * F;
  static constexpr int NUSED = FT == 3 ? 32 : 12 * F;       // leading channels that may be non-zero
  __device__ static constexpr int ch(int qt, int qh, int k) {   // k = qw*3 + c
    return FT == 3 ? (qt * 2 + qh) * 8 + k : (qt * 4 + qh * 2) * 3 + k;
  }
};
template <typename TO, int NCH> __device__ static inline void store_ch(char* dst, const float* v) {
  if constexpr (sizeof(TO) == 4) {
#pragma unroll
    for (int i = 0; i < NCH / 4; ++i) ((float4*)dst)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < NCH / 8; ++i) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[8 * i + e];
      ((bf16x8*)dst)[i] = o;
    }
  }
}
template <typename TI, int NUSED> __device__ static inline void load_ch(const char* src, float* v) {   // NUSED = 24 | 12
  if constexpr (sizeof(TI) == 4) {
#pragma unroll
    for (int i = 0; i < NUSED / 4; ++i) { const float4 f = ((const float4*)src)[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
  } else {
    float t[(NUSED + 7) / 8 * 8];
#pragma unroll
    for (int i = 0; i < (NUSED + 7) / 8; ++i) {
      const uint4 u = ((const uint4*)src)[i];
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { t[8 * i + 2 * k] = __uint_as_float(w[k] << 16); t[8 * i + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
#pragma unroll
    for (int i = 0; i < NUSED; ++i) v[i] = t[i];
  }
}

// six consecutive values x[b,t,h,2*w2 .. 2*w2+1, 0..2]
__device__ static inline void load6(const flk_apply_args& a, size_t off, float* x) {
  if (a.x_is_u8) {
    const uint16_t* p = (const uint16_t*)((const uint8_t*)a.x + off);   // off = 6*k: 2-byte aligned
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    const uint32_t by[6] = {w0 & 255, w0 >> 8, w1 & 255, w1 >> 8, w2 & 255, w2 >> 8};
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = (float)by[i] * a.x_scale + a.x_bias;
  } else {
    const float2* p = (const float2*)((const float*)a.x + off);        // off = 6*k: 8-byte aligned
    const float2 a0 = p[0], a1 = p[1], a2 = p[2];
    x[0] = a0.x; x[1] = a0.y; x[2] = a1.x; x[3] = a1.y; x[4] = a2.x; x[5] = a2.y;
  }
}

__device__ static inline float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
// value written for a pixel x with perturbation pv: the clipped sum, or (flk_apply_args.center) the sum minus pv -- i.e. x itself
// wherever the clip is inactive
__device__ static inline float applied(const flk_apply_args& a, float x, float pv) {
  const float u = x + pv;
  if (!a.center) return clipf(u, a.lo, a.hi);
  return (u < a.lo || u > a.hi) ? clipf(u, a.lo, a.hi) - pv : x;
}
__device__ static inline int wrap(int t, int T) { t %= T; return t < 0 ? t + T : t; }

// perturbation added at frame t (before adv_flag): p'[t] = p[(t - shift_p) mod T] (tf.roll), p = clip(delta)/std
// (delta_per_clip: clip b has its own [T,3] perturbation)
__device__ static inline float pert_at(const flk_apply_args& a, int b, int t, int h, int w, int c) {
  const int ts = wrap(t - a.shift_p, a.T);
  float d = a.delta_dense ? a.delta[(((size_t)ts * a.H + h) * a.W + w) * 3 + c] : a.delta[(a.delta_per_clip ? b * a.T : 0) * 3 + ts * 3 + c];
  const float dc = a.dclip_dev ? a.dclip_dev[b] : a.dclip;       // (per-clip clamp bound)
  if (dc > 0.f) d = clipf(d, -dc, dc);
  return d * a.inv_std[c];
}

// ---- apply: one thread = one space-to-depth output position (FT x 2 x 2 input cells x 3 channels) ----
template <typename TO, int FTL>
__global__ __launch_bounds__(256) void apply_s2d_kernel(const flk_apply_args a, char* out) {
  constexpr int FT = S2D<FTL>::F, NCH = S2D<FTL>::NCH;
  const int T2 = a.T / FT, H2 = a.H / 2, W2 = a.W / 2;
  const long total = (long)a.B * T2 * H2 * W2;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    long r = gid;
    const int w2 = r % W2; r /= W2;
    const int h2 = r % H2; r /= H2;
    const int t2 = r % T2;
    const int b = r / T2;
    float v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = 0.f;
#pragma unroll
    for (int qt = 0; qt < FT; ++qt) {
      const int t = FT * t2 + qt;
      const int tx = wrap(t - a.shift_x, a.T);      // x'[t] = x[(t - shift_x) mod T]
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        const int h = 2 * h2 + qh;
        float x[6];
        load6(a, ((((size_t)b * a.T + tx) * a.H + h) * a.W + 2 * w2) * 3, x);
#pragma unroll
        for (int qw = 0; qw < 2; ++qw)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float pv = a.adv_flag != 0.f ? a.adv_flag * pert_at(a, b, t, h, 2 * w2 + qw, c) : 0.f;
            v[S2D<FTL>::ch(qt, qh, qw * 3 + c)] = applied(a, x[qw * 3 + c], pv);
          }
      }
    }
    store_ch<TO, NCH>(out + (size_t)gid * NCH * sizeof(TO), v);
  }
}

// fold_t = 4: the (h,w) fold of fold_t = 1 with every value split into TWO bf16 numbers, [B,T,H/2,W/2,32]: channel k = bf16(x_adv),
// channel 16 + k = bf16(x_adv - channel k).  The VideoResNet stems (bf16 plans) read both halves against the same weights, i.e. they
// see the perturbed clip to ~16 mantissa bits at the MFMA cost of the 16 -> 32 channel padding their K step had anyway: rounding
// x + delta/std to ONE bf16 swallows a small flicker perturbation (|delta| = 1e-4 moved the logits with the wrong sign), and the
// centred-clip + position-bias form of the I3D stem is exact only where the clamp bounds are bf16 numbers -- here 5-8 % of the
// values of a clip sit AT a bound, and bf16(bound - p) jumps by a whole ulp for all of them at once (measured: wrong sign at
// |delta| = 5e-4).
__global__ __launch_bounds__(256) void apply_s2d_hilo_kernel(const flk_apply_args a, char* out) {
  const int H2 = a.H / 2, W2 = a.W / 2;
  const long total = (long)a.B * a.T * H2 * W2;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    long r = gid;
    const int w2 = r % W2; r /= W2;
    const int h2 = r % H2; r /= H2;
    const int t = r % a.T;
    const int b = r / a.T;
    const int tx = wrap(t - a.shift_x, a.T);
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      const int h = 2 * h2 + qh;
      float x[6];
      load6(a, ((((size_t)b * a.T + tx) * a.H + h) * a.W + 2 * w2) * 3, x);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float pv = a.adv_flag != 0.f ? a.adv_flag * pert_at(a, b, t, h, 2 * w2 + k / 3, k % 3) : 0.f;
        const float u = applied(a, x[k], pv);
        const float hi = (float)(bf16_t)u;
        v[S2D<1>::ch(0, qh, k)] = hi;                    // (stored again as bf16: exact)
        v[16 + S2D<1>::ch(0, qh, k)] = u - hi;
      }
    }
    store_ch<bf16_t, 32>(out + (size_t)gid * 32 * sizeof(bf16_t), v);
  }
}

// Fast path of the headline configuration (uint8 clip, flicker delta [T,3], FT = 2, W % 8 == 0): one thread = 4
// consecutive output positions.  It reads its 24 source bytes of each of the 4 (frame, row) pairs as three aligned
// 8-byte loads (the generic kernel above issues 2-byte loads), evaluates the 6 perturbation values (2 frames x RGB) once,
// and decodes its position with 32-bit arithmetic from a 3-D grid.  Same arithmetic per element as the generic kernel.
template <typename TO, int FTL>
__global__ __launch_bounds__(256) void apply_s2d_u8_flicker_kernel(const flk_apply_args a, char* out) {
  constexpr int NCH = 32;
  const int H2 = a.H / 2, W2 = a.W / 2, WG = W2 / 4;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)(H2 * WG)) return;
  const int h2 = i / WG, wg = i - h2 * WG;
  const int t2 = blockIdx.y, b = blockIdx.z;
  float pv[2][3];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int c = 0; c < 3; ++c) pv[qt][c] = a.adv_flag != 0.f ? a.adv_flag * pert_at(a, b, 2 * t2 + qt, 0, 0, c) : 0.f;
  uint2 raw[2][2][3];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int tx = wrap(2 * t2 + qt - a.shift_x, a.T);
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      const uint2* src = (const uint2*)((const uint8_t*)a.x + ((((size_t)b * a.T + tx) * a.H + 2 * h2 + qh) * a.W + 8 * wg) * 3);
#pragma unroll
      for (int k = 0; k < 3; ++k) raw[qt][qh][k] = src[k];
    }
  }
  char* dst = out + ((((size_t)b * (a.T / 2) + t2) * H2 + h2) * W2 + 4 * wg) * NCH * sizeof(TO);
#pragma unroll
  for (int j = 0; j < 4; ++j) {                   // output position 4*wg + j: source pixels 2j, 2j+1 of the 8-pixel run
    float v[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) v[k] = 0.f;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        const uint32_t w[6] = {raw[qt][qh][0].x, raw[qt][qh][0].y, raw[qt][qh][1].x, raw[qt][qh][1].y, raw[qt][qh][2].x, raw[qt][qh][2].y};
#pragma unroll
        for (int e = 0; e < 6; ++e) {              // byte 6j + e of the 24-byte run
          const int bi = 6 * j + e;
          const float x = (float)((w[bi >> 2] >> (8 * (bi & 3))) & 255u) * a.x_scale + a.x_bias;
          v[S2D<FTL>::ch(qt, qh, e)] = applied(a, x, pv[qt][e % 3]);
        }
      }
    store_ch<TO, NCH>(dst + (size_t)j * NCH * sizeof(TO), v);
  }
}

static int check_apply(const flk_apply_args* a) {
  FLK_REQUIRE(a && a->x && a->delta, "flk_perturb: null argument");
  FLK_REQUIRE(a->fold_t >= 0 && a->fold_t <= 4, "flk_perturb: fold_t must be 0 .. 4");
  FLK_REQUIRE(a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && (a->fold_t == 1 || a->fold_t == 4 || a->T % 2 == 0) && a->H % 2 == 0 && a->W % 2 == 0,
              "flk_perturb: H,W (and T when folded) must be positive and even (got %d,%d,%d)", a->T, a->H, a->W);
  FLK_REQUIRE(a->lo <= a->hi, "flk_perturb: lo > hi");
  FLK_REQUIRE(!(a->center && a->delta_dense), "flk_perturb: center = 1 is defined for the flicker perturbation [T,3] only");
  FLK_REQUIRE(!(a->delta_per_clip && a->delta_dense), "flk_perturb: delta_per_clip is defined for the flicker perturbation only");
  FLK_REQUIRE(!a->dclip_dev || a->delta_per_clip, "flk_perturb: dclip_dev (per-clip clamp bounds) needs delta_per_clip");
  return FLK_OK;
}

extern "C" int flk_perturb_apply_s2d(const flk_apply_args* a, void* out, int dtype, void* stream) {
  int rc = check_apply(a);
  if (rc) return rc;
  FLK_REQUIRE(out, "flk_perturb_apply_s2d: null out");
  const int ftl = (a->fold_t == 1 || a->fold_t == 4) ? 1 : a->fold_t == 3 ? 3 : 2, ft = ftl == 1 ? 1 : 2;
  const long total = (long)a->B * (a->T / ft) * (a->H / 2) * (a->W / 2);
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  FLK_REQUIRE(dtype == FLK_BF16 || dtype == FLK_F32, "flk_perturb_apply_s2d: bad dtype");
  if (a->fold_t == 4) {
    FLK_REQUIRE(dtype == FLK_BF16 && !a->center, "flk_perturb_apply_s2d: fold_t = 4 (two bf16 numbers per value) writes bf16, uncentred");
    FLK_LAUNCH_KERNEL(apply_s2d_hilo_kernel, dim3(grid), dim3(256), 0, st, *a, (char*)out);
    FLK_CHECK_HIP(hipGetLastError());
    return FLK_OK;
  }
  const bool bf = dtype == FLK_BF16;
  if (ft == 2 && a->x_is_u8 && !a->delta_dense && a->W % 8 == 0 && a->T / 2 < 65536 && a->B < 65536) {
    const dim3 g3((unsigned)(((a->H / 2) * (a->W / 8) + 255) / 256), (unsigned)(a->T / 2), (unsigned)a->B);
    if (bf && ftl == 2) FLK_LAUNCH_KERNEL((apply_s2d_u8_flicker_kernel<bf16_t, 2>), g3, dim3(256), 0, st, *a, (char*)out);
    else if (bf) FLK_LAUNCH_KERNEL((apply_s2d_u8_flicker_kernel<bf16_t, 3>), g3, dim3(256), 0, st, *a, (char*)out);
    else if (ftl == 2) FLK_LAUNCH_KERNEL((apply_s2d_u8_flicker_kernel<float, 2>), g3, dim3(256), 0, st, *a, (char*)out);
    else FLK_LAUNCH_KERNEL((apply_s2d_u8_flicker_kernel<float, 3>), g3, dim3(256), 0, st, *a, (char*)out);
    FLK_CHECK_HIP(hipGetLastError());
    return FLK_OK;
  }
#define FLK_APPLY(TT, L) FLK_LAUNCH_KERNEL((apply_s2d_kernel<TT, L>), dim3(grid), dim3(256), 0, st, *a, (char*)out)
  if (bf) { if (ftl == 1) FLK_APPLY(bf16_t, 1); else if (ftl == 2) FLK_APPLY(bf16_t, 2); else FLK_APPLY(bf16_t, 3); }
  else { if (ftl == 1) FLK_APPLY(float, 1); else if (ftl == 2) FLK_APPLY(float, 2); else FLK_APPLY(float, 3); }
#undef FLK_APPLY
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// ---- delta gradient --------------------------------------------------------------------------
// stage 1: workgroup (b, t2, chunk of h2 rows) -> partials[wg][qt][c] = sum over its cells of
//          g * 1[lo <= x' + a p' <= hi]      (both clip gradients are inclusive at the bounds)
static inline int grad_nchunk(int B, int T, int H) {
  const int bt = B * (T / 2 > 0 ? T / 2 : 1), H2 = H / 2;
  int n = (1024 + bt - 1) / bt;
  if (n > H2) n = H2;
  if (n < 1) n = 1;
  return n;
}

template <typename TI, int FTL>
__global__ __launch_bounds__(256) void grad_reduce_stage1(const flk_apply_args a, const char* gx, int nchunk, float* partials) {
  constexpr int FT = S2D<FTL>::F, NCH = S2D<FTL>::NCH, NUSED = S2D<FTL>::NUSED;
  const int T2 = a.T / FT, H2 = a.H / 2, W2 = a.W / 2;
  int bid = blockIdx.x;
  const int chunk = bid % nchunk; bid /= nchunk;
  const int t2 = bid % T2;
  const int b = bid / T2;
  const int h_lo = (int)((long)H2 * chunk / nchunk), h_hi = (int)((long)H2 * (chunk + 1) / nchunk);
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ncell = (h_hi - h_lo) * W2;
  for (int i = threadIdx.x; i < ncell; i += 256) {
    const int h2 = h_lo + i / W2, w2 = i % W2;
    float g[NUSED];
    load_ch<TI, NUSED>(gx + ((((size_t)b * T2 + t2) * H2 + h2) * W2 + w2) * NCH * sizeof(TI), g);
#pragma unroll
    for (int qt = 0; qt < FT; ++qt) {
      const int t = FT * t2 + qt, tx = wrap(t - a.shift_x, a.T);
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        const int h = 2 * h2 + qh;
        float x[6];
        load6(a, ((((size_t)b * a.T + tx) * a.H + h) * a.W + 2 * w2) * 3, x);
#pragma unroll
        for (int qw = 0; qw < 2; ++qw)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float u = x[qw * 3 + c] + a.adv_flag * pert_at(a, b, t, h, 2 * w2 + qw, c);
            if (u >= a.lo && u <= a.hi) acc[qt * 3 + c] += g[S2D<FTL>::ch(qt, qh, qw * 3 + c)];
          }
      }
    }
  }
  __shared__ float red[4][6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6)
    partials[(size_t)blockIdx.x * 6 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// stage 2: thread (t,c) sums partials over (b, chunk) in a fixed order; maps frame t back to the delta
// index it was rolled from; applies adv_flag, 1/std and the delta clip mask.  delta_per_clip: one (t,c) row per clip
// (blockIdx.y = clip), summed over that clip's chunks only -- the order of a batch-1 call with the same chunk count.
__global__ void grad_reduce_stage2(const flk_apply_args a, int nchunk, const float* partials, float* gdelta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.T * 3) return;
  const int FT = (a.fold_t == 1 || a.fold_t == 4) ? 1 : 2;      // frames per folded position (fold_t 0, 2, 3: two)
  const int t = i / 3, c = i % 3, t2 = t / FT, qt = t % FT, T2 = a.T / FT;
  const int b_lo = a.delta_per_clip ? (int)blockIdx.y : 0, b_hi = a.delta_per_clip ? b_lo + 1 : a.B;
  float s = 0.f;
  for (int b = b_lo; b < b_hi; ++b)
    for (int k = 0; k < nchunk; ++k) s += partials[(((size_t)b * T2 + t2) * nchunk + k) * 6 + qt * 3 + c];
  const int ts = wrap(t - a.shift_p, a.T);
  const size_t dbase = a.delta_per_clip ? (size_t)b_lo * a.T * 3 : 0;
  const float d = a.delta[dbase + ts * 3 + c];
  const float dc = a.dclip_dev ? a.dclip_dev[b_lo] : a.dclip;
  const bool pass = !(dc > 0.f) || (d >= -dc && d <= dc);
  gdelta[dbase + ts * 3 + c] = pass ? s * a.adv_flag * a.inv_std[c] : 0.f;
}

// stage 2 alone, for producers of stage-1 partials outside this file (stem_grad.hip)
int flk_grad_reduce_stage2_launch(const flk_apply_args* a, int nchunk, const float* partials, float* gdelta, hipStream_t s) {
  FLK_LAUNCH_KERNEL(grad_reduce_stage2, dim3((a->T * 3 + 127) / 128, a->delta_per_clip ? a->B : 1), dim3(128), 0, s, *a, nchunk, partials, gdelta);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// dense delta ("L12" baseline, kinetics_i3d_utils.py:308-521): no spatial reduction, sum over the batch.
template <typename TI, int FTL>
__global__ __launch_bounds__(256) void grad_dense_kernel(const flk_apply_args a, const char* gx, float* gdelta) {
  constexpr int FT = S2D<FTL>::F, NCH = S2D<FTL>::NCH, NUSED = S2D<FTL>::NUSED;
  const int T2 = a.T / FT, H2 = a.H / 2, W2 = a.W / 2;
  const long total = (long)T2 * H2 * W2;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    long r = gid;
    const int w2 = r % W2; r /= W2;
    const int h2 = r % H2;
    const int t2 = r / H2;
    float acc[NUSED];
#pragma unroll
    for (int k = 0; k < NUSED; ++k) acc[k] = 0.f;
    for (int b = 0; b < a.B; ++b) {
      float g[NUSED];
      load_ch<TI, NUSED>(gx + ((((size_t)b * T2 + t2) * H2 + h2) * W2 + w2) * NCH * sizeof(TI), g);
#pragma unroll
      for (int qt = 0; qt < FT; ++qt) {
        const int t = FT * t2 + qt, tx = wrap(t - a.shift_x, a.T);
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
          float x[6];
          load6(a, ((((size_t)b * a.T + tx) * a.H + 2 * h2 + qh) * a.W + 2 * w2) * 3, x);
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const float u = x[k] + a.adv_flag * pert_at(a, b, t, 2 * h2 + qh, 2 * w2 + k / 3, k % 3);
            if (u >= a.lo && u <= a.hi) acc[S2D<FTL>::ch(qt, qh, k)] += g[S2D<FTL>::ch(qt, qh, k)];
          }
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < FT; ++qt)
#pragma unroll
      for (int qh = 0; qh < 2; ++qh)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int t = FT * t2 + qt, ts = wrap(t - a.shift_p, a.T), c = k % 3;
          const size_t di = (((size_t)ts * a.H + 2 * h2 + qh) * a.W + 2 * w2 + k / 3) * 3 + c;
          const float d = a.delta[di];
          const bool pass = !(a.dclip > 0.f) || (d >= -a.dclip && d <= a.dclip);
          gdelta[di] = pass ? acc[S2D<FTL>::ch(qt, qh, k)] * a.adv_flag * a.inv_std[c] : 0.f;
        }
  }
}

extern "C" int64_t flk_perturb_grad_scratch_bytes(int B, int T, int H, int W) {
  (void)W;
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  return (int64_t)B * T * grad_nchunk(1, T, H) * 6 * sizeof(float);   // covers both fold_t settings and the per-clip chunking (batch-1 chunk count)
}

extern "C" int flk_perturb_grad_reduce(const flk_apply_args* a, const void* gx_s2d, int dtype, float* gdelta,
                                       float* partials, void* stream) {
  int rc = check_apply(a);
  if (rc) return rc;
  FLK_REQUIRE(gx_s2d && gdelta, "flk_perturb_grad_reduce: null argument");
  FLK_REQUIRE(dtype == FLK_BF16 || dtype == FLK_F32, "flk_perturb_grad_reduce: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  // (fold_t = 4: the clip went in as two bf16 numbers per value; its gradient comes back in the 16-channel fold_t = 1 layout)
  const int ftl = (a->fold_t == 1 || a->fold_t == 4) ? 1 : a->fold_t == 3 ? 3 : 2, ft = ftl == 1 ? 1 : 2;
  const bool bf = dtype == FLK_BF16;
  if (a->delta_dense) {
    const long total = (long)(a->T / ft) * (a->H / 2) * (a->W / 2);
    const unsigned grid = (unsigned)((total + 255) / 256);
#define FLK_GD(TT, L) FLK_LAUNCH_KERNEL((grad_dense_kernel<TT, L>), dim3(grid), dim3(256), 0, s, *a, (const char*)gx_s2d, gdelta)
    if (bf) { if (ftl == 1) FLK_GD(bf16_t, 1); else if (ftl == 2) FLK_GD(bf16_t, 2); else FLK_GD(bf16_t, 3); }
    else { if (ftl == 1) FLK_GD(float, 1); else if (ftl == 2) FLK_GD(float, 2); else FLK_GD(float, 3); }
#undef FLK_GD
  } else {
    FLK_REQUIRE(partials, "flk_perturb_grad_reduce: null scratch");
    // per-clip perturbations: the chunking (= summation order) of a batch-1 call, so that clip b of a batch follows the same
    // trajectory, bit for bit in fp32, as the same clip attacked alone
    const int nchunk = grad_nchunk(a->delta_per_clip ? 1 : a->B, a->T, a->H);
    const unsigned grid = (unsigned)(a->B * (a->T / ft) * nchunk);
#define FLK_GR(TT, L) FLK_LAUNCH_KERNEL((grad_reduce_stage1<TT, L>), dim3(grid), dim3(256), 0, s, *a, (const char*)gx_s2d, nchunk, partials)
    if (bf) { if (ftl == 1) FLK_GR(bf16_t, 1); else if (ftl == 2) FLK_GR(bf16_t, 2); else FLK_GR(bf16_t, 3); }
    else { if (ftl == 1) FLK_GR(float, 1); else if (ftl == 2) FLK_GR(float, 2); else FLK_GR(float, 3); }
#undef FLK_GR
    FLK_LAUNCH_KERNEL(grad_reduce_stage2, dim3((a->T * 3 + 127) / 128, a->delta_per_clip ? a->B : 1), dim3(128), 0, s, *a, nchunk, partials, gdelta);
  }
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

__global__ void pack_batch_sums_kernel(const float* per_clip, int B, float prob_scale, float* out3) {
  const int k = threadIdx.x;
  if (k >= 3) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += per_clip[b * 4 + k];
  out3[k] = k == 0 ? s : s * prob_scale;
}

extern "C" int flk_pack_batch_sums(const float* per_clip, int B, float prob_scale, float* out3, void* stream) {
  FLK_REQUIRE(per_clip && out3 && B > 0, "flk_pack_batch_sums: bad arguments");
  FLK_LAUNCH_KERNEL(pack_batch_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, per_clip, B, prob_scale, out3);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// ---- regulariser gradient + Adam: one workgroup, delta is [T,3] ----------------------------------
__device__ static inline float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ static inline float block_max(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

constexpr int ADAM_PER = 8;  // 256 threads x 8 >= 3*T  (T <= 682)

// blockIdx.x = clip (flk_perturb_reg_adam_batched: B independent perturbations, each with its own Adam state, step counter and
// "still attacking" flag, all on the device -- no host value changes between iterations, so the loop can be replayed as a graph)
__global__ __launch_bounds__(256) void reg_adam_kernel(const flk_adam_args a, const float* g_adv, float* delta, float* m, float* v,
                                                       float* scalars, int* steps, const int* active, const float* dyn_dev) {
  __shared__ float sh[4];
  const int T = a.T, N = 3 * T;
  {
    const size_t o = (size_t)blockIdx.x * N;
    g_adv += o; delta += o; m += o; v += o;
    if (scalars) scalars += (size_t)blockIdx.x * 8;
  }
  const int step = steps ? steps[blockIdx.x] + 1 : a.step;        // 1-based step of THIS update
  const bool update = !active || active[blockIdx.x] != 0;         // a retired clip keeps its state; its scalars are still reported
  const float dyn = dyn_dev ? dyn_dev[blockIdx.x] : a.dyn_max_norm;
  // value the regulariser sees: raw delta (TF, kinetics_i3d_utils.py:172) or clamp(delta) (torch, model.py:1078)
  auto rv = [&](int t, int c) -> float {
    const float d = delta[wrap(t, T) * 3 + c];
    return a.torch_dialect ? clipf(d, -dyn, dyn) : d;
  };
  float nd[ADAM_PER], nm[ADAM_PER], nv[ADAM_PER];
  float s_norm = 0.f, s_diff = 0.f, s_lap = 0.f, s_abs = 0.f, s_rough = 0.f, s_max = -INFINITY, s_min = INFINITY;
  const float lr_tf = a.lr * sqrtf(1.f - powf(a.adam_b2, (float)step)) / (1.f - powf(a.adam_b1, (float)step));
  const float bc1 = 1.f - powf(a.adam_b1, (float)step), bc2s = sqrtf(1.f - powf(a.adam_b2, (float)step));
#pragma unroll
  for (int k = 0; k < ADAM_PER; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i >= N) continue;
    const int t = i / 3, c = i % 3;
    const float x0 = rv(t, c), xm1 = rv(t - 1, c), xp1 = rv(t + 1, c), xm2 = rv(t - 2, c), xp2 = rv(t + 2, c);
    const float l0 = -2.f * x0 + xm1 + xp1;         // laplacian at t, t-1, t+1
    const float lm = -2.f * xm1 + xm2 + x0;
    const float lp = -2.f * xp1 + x0 + xp2;
    const float df = x0 - xm1;
    s_norm += x0 * x0; s_diff += df * df; s_lap += l0 * l0;
    const float raw = delta[i], rawm1 = delta[wrap(t - 1, T) * 3 + c];
    s_abs += fabsf(raw); s_rough += fabsf(raw - rawm1);
    s_max = fmaxf(s_max, raw); s_min = fminf(s_min, raw);
    float greg = a.beta1 * 2.f * x0 / N + a.beta2 * 2.f * (-l0) / N + a.beta3 * 2.f * (-2.f * l0 + lm + lp) / N;
    if (a.torch_dialect && !(raw >= -dyn && raw <= dyn)) greg = 0.f;   // clamp gradient, inclusive
    const float g = a.g_scale * g_adv[i] + a.beta0 * greg;
    const float mi = a.adam_b1 * m[i] + (1.f - a.adam_b1) * g;
    const float vi = a.adam_b2 * v[i] + (1.f - a.adam_b2) * g * g;
    nm[k] = mi; nv[k] = vi;
    if (a.torch_dialect) nd[k] = raw - (a.lr / bc1) * mi / (sqrtf(vi) / bc2s + a.adam_eps);   // torch-1.4 Adam
    else nd[k] = raw - lr_tf * mi / (sqrtf(vi) + a.adam_eps);                                   // TF-1.15 AdamOptimizer
  }
  const float t_norm = block_sum(s_norm, sh), t_diff = block_sum(s_diff, sh), t_lap = block_sum(s_lap, sh);
  const float t_abs = block_sum(s_abs, sh), t_rough = block_sum(s_rough, sh);
  const float t_max = block_max(s_max, sh), t_min = -block_max(-s_min, sh);
  __syncthreads();   // every neighbour read of delta is done
  if (update) {
#pragma unroll
    for (int k = 0; k < ADAM_PER; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i >= N) continue;
      delta[i] = nd[k]; m[i] = nm[k]; v[i] = nv[k];
    }
    if (threadIdx.x == 0 && steps) steps[blockIdx.x] = step;
  }
  if (threadIdx.x == 0 && scalars) {
    const float norm = t_norm / N + 1e-12f, diff = t_diff / N + 1e-12f, lap = t_lap / N + 1e-12f;
    scalars[0] = a.beta1 * norm + a.beta2 * diff + a.beta3 * lap;
    scalars[1] = norm; scalars[2] = diff; scalars[3] = lap;
    scalars[4] = t_abs / N; scalars[5] = t_rough / N; scalars[6] = t_max; scalars[7] = t_min;
  }
}

extern "C" int flk_perturb_reg_adam(const flk_adam_args* a, const float* g_adv, float* delta, float* m, float* v,
                                    float* scalars, void* stream) {
  FLK_REQUIRE(a && g_adv && delta && m && v, "flk_perturb_reg_adam: null argument");
  FLK_REQUIRE(a->T > 0 && 3 * a->T <= 256 * ADAM_PER, "flk_perturb_reg_adam: T out of range (%d)", a->T);
  FLK_REQUIRE(a->step >= 1, "flk_perturb_reg_adam: step is 1-based");
  FLK_LAUNCH_KERNEL(reg_adam_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *a, g_adv, delta, m, v, scalars, (int*)nullptr, (const int*)nullptr, (const float*)nullptr);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

extern "C" int flk_perturb_reg_adam_batched(const flk_adam_args* a, int nclip, const float* g_adv, float* delta, float* m, float* v,
                                            int* steps_dev, const int* active_dev, const float* dyn_max_norm_dev, float* scalars, void* stream) {
  FLK_REQUIRE(a && g_adv && delta && m && v && steps_dev, "flk_perturb_reg_adam_batched: null argument");
  FLK_REQUIRE(nclip > 0 && nclip < 65536, "flk_perturb_reg_adam_batched: bad clip count %d", nclip);
  FLK_REQUIRE(a->T > 0 && 3 * a->T <= 256 * ADAM_PER, "flk_perturb_reg_adam_batched: T out of range (%d)", a->T);
  FLK_LAUNCH_KERNEL(reg_adam_kernel, dim3((unsigned)nclip), dim3(256), 0, (hipStream_t)stream, *a, g_adv, delta, m, v, scalars, steps_dev, active_dev, dyn_max_norm_dev);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// ---- dense delta: L12 regulariser + Adam (kinetics_i3d_L12) ------------------------------------------------------------
// d/d(delta_t) sqrt(mean_{hwc} delta_t^2) = delta_t / (N_f * sqrt(mean_t)),  N_f = H*W*3.
constexpr int DENSE_CHUNKS = 64;     // workgroups per frame in the reduction pass

__global__ __launch_bounds__(256) void dense_frame_stats(const float* delta, int frame_elems, float* part, float clampv) {
  // part[(t*DENSE_CHUNKS + c)*4 + {0,1,2,3}] = {sum clamp(d)^2, sum |d|, sum |d - d_prev_frame|, max |d|} over this chunk
  // (clampv > 0: the regulariser sees the clamped perturbation; the metrics are taken on the raw one, model.py:113-118)
  const int t = blockIdx.y, c = blockIdx.x, T = gridDim.y;
  const float4* d4 = (const float4*)(delta + (size_t)t * frame_elems);
  const float4* p4 = (const float4*)(delta + (size_t)((t + T - 1) % T) * frame_elems);
  const int n4 = frame_elems / 4;
  float sq = 0.f, ab = 0.f, ro = 0.f, mx = 0.f;
  for (int i = c * 256 + threadIdx.x; i < n4; i += DENSE_CHUNKS * 256) {
    const float4 a = d4[i], b = p4[i];
    {
      const float cx = fminf(fmaxf(a.x, -clampv), clampv), cy = fminf(fmaxf(a.y, -clampv), clampv);
      const float cz = fminf(fmaxf(a.z, -clampv), clampv), cw = fminf(fmaxf(a.w, -clampv), clampv);
      sq += cx * cx + cy * cy + cz * cz + cw * cw;
    }
    ab += fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w);
    ro += fabsf(a.x - b.x) + fabsf(a.y - b.y) + fabsf(a.z - b.z) + fabsf(a.w - b.w);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
  }
  __shared__ float sh[4];
  const float s0 = block_sum(sq, sh), s1 = block_sum(ab, sh), s2 = block_sum(ro, sh), s3 = block_max(mx, sh);
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)t * DENSE_CHUNKS + c) * 4;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
  }
}

__global__ void dense_frame_finish(const float* part, int T, int frame_elems, float* frame_rms, float* scalars) {
  // one thread per frame sums its chunks in a fixed order; thread 0 then folds the frames (T <= 1024)
  __shared__ float s_l12[1024], s_ab[1024], s_ro[1024], s_mx[1024];
  const int t = threadIdx.x;
  float sq = 0.f, ab = 0.f, ro = 0.f, mx = 0.f;
  if (t < T)
    for (int c = 0; c < DENSE_CHUNKS; ++c) {
      const float* o = part + ((size_t)t * DENSE_CHUNKS + c) * 4;
      sq += o[0]; ab += o[1]; ro += o[2]; mx = fmaxf(mx, o[3]);
    }
  const float rms = sqrtf(sq / (float)frame_elems);
  if (t < T) frame_rms[t] = rms;
  s_l12[t] = t < T ? rms : 0.f; s_ab[t] = ab; s_ro[t] = ro; s_mx[t] = mx;
  __syncthreads();
  if (t == 0 && scalars) {
    float l12 = 0.f, a = 0.f, r = 0.f, m = 0.f;
    for (int i = 0; i < T; ++i) { l12 += s_l12[i]; a += s_ab[i]; r += s_ro[i]; m = fmaxf(m, s_mx[i]); }
    const float N = (float)T * (float)frame_elems;
    scalars[0] = l12 + 1e-12f; scalars[1] = a / N; scalars[2] = r / N; scalars[3] = m;
  }
}

__global__ __launch_bounds__(256) void dense_adam_kernel(const flk_dense_adam_args a, int frame_elems, const float* frame_rms,
                                                         const float* g_adv, float* delta, float* m, float* v) {
  const int t = blockIdx.y;
  const float rms = frame_rms[t];
  // sqrt'(0) is unbounded in the reference too (tf.sqrt gradient at 0 -> inf); delta is initialised to 1e-8 for that reason
  const float rcoef = a.beta / ((float)frame_elems * rms);
  const float bc1 = 1.f - powf(a.adam_b1, (float)a.step), bc2s = sqrtf(1.f - powf(a.adam_b2, (float)a.step));
  const float lr_tf = a.lr * bc2s / bc1;
  const size_t base = (size_t)t * frame_elems;
  const int n4 = frame_elems / 4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const float4 d = ((const float4*)(delta + base))[i], g4 = ((const float4*)(g_adv + base))[i];
    float4 mm = ((const float4*)(m + base))[i], vv = ((const float4*)(v + base))[i];
    float dd[4] = {d.x, d.y, d.z, d.w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float mv[4] = {mm.x, mm.y, mm.z, mm.w}, vvv[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // d(L12)/d(delta) through the clamp: passes where |delta| <= dyn_max_norm (torch.clamp gradient, bounds inclusive)
      const float g = a.g_scale * gg[k] + ((a.dyn_max_norm > 0.f && fabsf(dd[k]) > a.dyn_max_norm) ? 0.f : rcoef * dd[k]);
      mv[k] = a.adam_b1 * mv[k] + (1.f - a.adam_b1) * g;
      vvv[k] = a.adam_b2 * vvv[k] + (1.f - a.adam_b2) * g * g;
      dd[k] = a.torch_dialect ? dd[k] - (a.lr / bc1) * mv[k] / (sqrtf(vvv[k]) / bc2s + a.adam_eps)
                              : dd[k] - lr_tf * mv[k] / (sqrtf(vvv[k]) + a.adam_eps);
    }
    ((float4*)(delta + base))[i] = make_float4(dd[0], dd[1], dd[2], dd[3]);
    ((float4*)(m + base))[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    ((float4*)(v + base))[i] = make_float4(vvv[0], vvv[1], vvv[2], vvv[3]);
  }
}

extern "C" int64_t flk_dense_adam_scratch_bytes(int T, int H, int W) {
  (void)H; (void)W;
  return T > 0 ? (int64_t)(T * DENSE_CHUNKS * 4 + T) * sizeof(float) : 0;
}

extern "C" int flk_perturb_dense_l12_adam(const flk_dense_adam_args* a, const float* g_adv, float* delta, float* m, float* v,
                                          float* scalars, float* scratch, void* stream) {
  FLK_REQUIRE(a && g_adv && delta && m && v && scratch, "flk_perturb_dense_l12_adam: null argument");
  FLK_REQUIRE(a->T > 0 && a->T <= 1024 && a->H > 0 && a->W > 0 && (a->H * a->W * 3) % 4 == 0, "flk_perturb_dense_l12_adam: bad dims");
  FLK_REQUIRE(a->step >= 1, "flk_perturb_dense_l12_adam: step is 1-based");
  const int fe = a->H * a->W * 3;
  float* part = scratch;
  float* frame_rms = scratch + (size_t)a->T * DENSE_CHUNKS * 4;
  hipStream_t s = (hipStream_t)stream;
  FLK_LAUNCH_KERNEL(dense_frame_stats, dim3(DENSE_CHUNKS, a->T), dim3(256), 0, s, delta, fe, part,
                     a->dyn_max_norm > 0.f ? a->dyn_max_norm : INFINITY);
  FLK_LAUNCH_KERNEL(dense_frame_finish, dim3(1), dim3(1024), 0, s, part, a->T, fe, frame_rms, scalars);
  FLK_LAUNCH_KERNEL(dense_adam_kernel, dim3(64, a->T), dim3(256), 0, s, *a, fe, frame_rms, g_adv, delta, m, v);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

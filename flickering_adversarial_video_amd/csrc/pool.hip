// tf.nn.max_pool3d (SAME) forward and MaxPool3DGrad, channels-last, 16-byte vectorised over channels.
// HBM-bound elementwise kernels: one thread = one position x one 16-byte channel group.
//   forward : first maximum in (t,h,w) scan order wins (strict >), padded cells never win
//             (i3d.py:174,189,212,252,398); the winning window index is kept as one byte.
//   backward: gather form -- every INPUT position sums the output gradients whose saved argmax points
//             at it.  No atomics, bitwise reproducible.  Optional accumulate (+add) and relu mask of the
//             producing layer (mask > 0) fused in.
#include "flk_internal.h"

struct PoolKP {
  const char* in; char* out; uint8_t* idx;
  const char* gout; char* gin; const char* mask; const char* add;
  int in_ld, in_coff, out_ld, out_coff, C;
  int gout_ld, gout_coff, gin_ld, gin_coff, mask_ld, mask_coff;
  int B, Ti, Hi, Wi, To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
};

template <typename T> struct PV;
template <> struct PV<bf16_t> {
  static constexpr int EPL = 8;
  __device__ static inline void ld(const char* p, float* f) {
    const uint4 u = *(const uint4*)p;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  __device__ static inline void st(char* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
    *(bf16x8*)p = v;
  }
  __device__ static inline void ldidx(const uint8_t* p, int* i) {
    const uint2 u = *(const uint2*)p;
#pragma unroll
    for (int k = 0; k < 4; ++k) { i[k] = (u.x >> (8 * k)) & 255; i[4 + k] = (u.y >> (8 * k)) & 255; }
  }
  __device__ static inline void stidx(uint8_t* p, const int* i) {
    uint2 u;
    u.x = i[0] | (i[1] << 8) | (i[2] << 16) | (i[3] << 24);
    u.y = i[4] | (i[5] << 8) | (i[6] << 16) | (i[7] << 24);
    *(uint2*)p = u;
  }
};
template <> struct PV<float> {
  static constexpr int EPL = 4;
  __device__ static inline void ld(const char* p, float* f) {
    const float4 u = *(const float4*)p; f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w;
  }
  __device__ static inline void st(char* p, const float* f) { *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); }
  __device__ static inline void ldidx(const uint8_t* p, int* i) {
    const uint32_t u = *(const uint32_t*)p;
#pragma unroll
    for (int k = 0; k < 4; ++k) i[k] = (u >> (8 * k)) & 255;
  }
  __device__ static inline void stidx(uint8_t* p, const int* i) {
    *(uint32_t*)p = i[0] | (i[1] << 8) | (i[2] << 16) | (i[3] << 24);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const PoolKP p) {
  constexpr int EPL = PV<T>::EPL;
  const int ng = p.C / EPL;
  const long total = (long)p.B * p.To * p.Ho * p.Wo * ng;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int cg = gid % ng;
    long pos = gid / ng;
    const int ow = pos % p.Wo; pos /= p.Wo;
    const int oh = pos % p.Ho; pos /= p.Ho;
    const int ot = pos % p.To;
    const int b = pos / p.To;
    float best[EPL];
    int bi[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    int tap = 0;
    for (int dt = 0; dt < p.kt; ++dt) {
      const int it = ot * p.st - p.pt + dt;
      for (int dh = 0; dh < p.kh; ++dh) {
        const int ih = oh * p.sh - p.ph + dh;
        for (int dw = 0; dw < p.kw; ++dw, ++tap) {
          const int iw = ow * p.sw - p.pw + dw;
          if ((unsigned)it >= (unsigned)p.Ti || (unsigned)ih >= (unsigned)p.Hi || (unsigned)iw >= (unsigned)p.Wi) continue;
          float v[EPL];
          PV<T>::ld(p.in + ((((size_t)(b * p.Ti + it) * p.Hi + ih) * p.Wi + iw) * p.in_ld + p.in_coff + cg * EPL) * sizeof(T), v);
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            if (v[e] > best[e]) { best[e] = v[e]; bi[e] = tap; }
        }
      }
    }
    const size_t opos = (((size_t)(b * p.To + ot) * p.Ho + oh) * p.Wo + ow);
    PV<T>::st(p.out + (opos * p.out_ld + p.out_coff + cg * EPL) * sizeof(T), best);
    PV<T>::stidx(p.idx + opos * p.C + cg * EPL, bi);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolKP p) {
  constexpr int EPL = PV<T>::EPL;
  const int ng = p.C / EPL;
  const long total = (long)p.B * p.Ti * p.Hi * p.Wi * ng;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int cg = gid % ng;
    long pos = gid / ng;
    const int iw = pos % p.Wi; pos /= p.Wi;
    const int ih = pos % p.Hi; pos /= p.Hi;
    const int it = pos % p.Ti;
    const int b = pos / p.Ti;
    const size_t ipos = (((size_t)(b * p.Ti + it) * p.Hi + ih) * p.Wi + iw);
    float g[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) g[e] = 0.f;
    if (p.add) PV<T>::ld(p.add + (ipos * p.gin_ld + p.gin_coff + cg * EPL) * sizeof(T), g);
    // windows containing this cell: o in [ceil((i+p-k+1)/s), floor((i+p)/s)] per dim; its tap index d = i+p-o*s
    const int ot_lo = max(0, (it + p.pt - p.kt + p.st) / p.st), ot_hi = min(p.To - 1, (it + p.pt) / p.st);
    const int oh_lo = max(0, (ih + p.ph - p.kh + p.sh) / p.sh), oh_hi = min(p.Ho - 1, (ih + p.ph) / p.sh);
    const int ow_lo = max(0, (iw + p.pw - p.kw + p.sw) / p.sw), ow_hi = min(p.Wo - 1, (iw + p.pw) / p.sw);
    for (int ot = ot_lo; ot <= ot_hi; ++ot)
      for (int oh = oh_lo; oh <= oh_hi; ++oh)
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
          const int tap = ((it + p.pt - ot * p.st) * p.kh + (ih + p.ph - oh * p.sh)) * p.kw + (iw + p.pw - ow * p.sw);
          const size_t opos = (((size_t)(b * p.To + ot) * p.Ho + oh) * p.Wo + ow);
          int id[EPL];
          PV<T>::ldidx(p.idx + opos * p.C + cg * EPL, id);
          bool any = false;
#pragma unroll
          for (int e = 0; e < EPL; ++e) any |= id[e] == tap;
          if (!any) continue;
          float go[EPL];
          PV<T>::ld(p.gout + (opos * p.gout_ld + p.gout_coff + cg * EPL) * sizeof(T), go);
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            if (id[e] == tap) g[e] += go[e];
        }
    if (p.mask) {
      float mk[EPL];
      PV<T>::ld(p.mask + (ipos * p.mask_ld + p.mask_coff + cg * EPL) * sizeof(T), mk);
#pragma unroll
      for (int e = 0; e < EPL; ++e) g[e] = mk[e] > 0.f ? g[e] : 0.f;
    }
    PV<T>::st(p.gin + (ipos * p.gin_ld + p.gin_coff + cg * EPL) * sizeof(T), g);
  }
}

static int check_pool(const flk_pool_args* a) {
  FLK_REQUIRE(a && a->in && a->idx, "flk_maxpool3d: null argument");
  FLK_REQUIRE(a->C % 8 == 0 && a->in_ld % 8 == 0 && a->in_coff % 8 == 0 && a->out_ld % 8 == 0 && a->out_coff % 8 == 0,
              "flk_maxpool3d: channel counts / strides / offsets must be multiples of 8");
  FLK_REQUIRE(a->kt * a->kh * a->kw <= 255 && a->kt > 0 && a->kh > 0 && a->kw > 0, "flk_maxpool3d: window too large");
  FLK_REQUIRE(a->B > 0 && a->To > 0 && a->Ho > 0 && a->Wo > 0 && a->st > 0 && a->sh > 0 && a->sw > 0, "flk_maxpool3d: bad dims");
  // every window must contain at least one in-bounds cell
  FLK_REQUIRE(a->pt < a->kt && a->ph < a->kh && a->pw < a->kw &&
                  (a->To - 1) * a->st - a->pt < a->Ti && (a->Ho - 1) * a->sh - a->ph < a->Hi &&
                  (a->Wo - 1) * a->sw - a->pw < a->Wi, "flk_maxpool3d: a window lies entirely in the padding");
  return FLK_OK;
}

static void fill(PoolKP& kp, const flk_pool_args* a) {
  kp.in = (const char*)a->in; kp.out = (char*)a->out; kp.idx = a->idx;
  kp.in_ld = a->in_ld; kp.in_coff = a->in_coff; kp.out_ld = a->out_ld; kp.out_coff = a->out_coff; kp.C = a->C;
  kp.B = a->B; kp.Ti = a->Ti; kp.Hi = a->Hi; kp.Wi = a->Wi; kp.To = a->To; kp.Ho = a->Ho; kp.Wo = a->Wo;
  kp.kt = a->kt; kp.kh = a->kh; kp.kw = a->kw; kp.st = a->st; kp.sh = a->sh; kp.sw = a->sw;
  kp.pt = a->pt; kp.ph = a->ph; kp.pw = a->pw;
}

static unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

extern "C" int flk_maxpool3d_fwd(const flk_pool_args* a, int dtype, void* stream) {
  int rc = check_pool(a);
  if (rc) return rc;
  FLK_REQUIRE(a->out, "flk_maxpool3d_fwd: null out");
  PoolKP kp{};
  fill(kp, a);
  const int epl = dtype == FLK_BF16 ? 8 : 4;
  const long total = (long)a->B * a->To * a->Ho * a->Wo * (a->C / epl);
  if (dtype == FLK_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, kp);
  else if (dtype == FLK_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, kp);
  else { flk_set_error("flk_maxpool3d_fwd: bad dtype"); return FLK_EINVAL; }
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

extern "C" int flk_maxpool3d_bwd(const flk_pool_args* a, const void* gout, int gout_ld, int gout_coff,
                                 void* gin, int gin_ld, int gin_coff, const void* mask, int mask_ld, int mask_coff,
                                 int dtype, void* stream) {
  int rc = check_pool(a);
  if (rc) return rc;
  FLK_REQUIRE(gout && gin, "flk_maxpool3d_bwd: null gradient");
  FLK_REQUIRE(gout_ld % 8 == 0 && gout_coff % 8 == 0 && gin_ld % 8 == 0 && gin_coff % 8 == 0 && mask_ld % 8 == 0 && mask_coff % 8 == 0,
              "flk_maxpool3d_bwd: ld/coff must be multiples of 8");
  PoolKP kp{};
  fill(kp, a);
  kp.gout = (const char*)gout; kp.gout_ld = gout_ld; kp.gout_coff = gout_coff;
  kp.gin = (char*)gin; kp.gin_ld = gin_ld; kp.gin_coff = gin_coff;
  kp.mask = (const char*)mask; kp.mask_ld = mask_ld; kp.mask_coff = mask_coff;
  kp.add = nullptr;
  const int epl = dtype == FLK_BF16 ? 8 : 4;
  const long total = (long)a->B * a->Ti * a->Hi * a->Wi * (a->C / epl);
  if (dtype == FLK_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, kp);
  else if (dtype == FLK_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, kp);
  else { flk_set_error("flk_maxpool3d_bwd: bad dtype"); return FLK_EINVAL; }
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

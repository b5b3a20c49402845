// Classifier head and loss head (small fp32 kernels).
//
// I3D Logits endpoint (i3d.py:459-474): avg_pool3d 2x7x7 VALID s1 -> dropout(keep 1.0) -> 1x1x1 conv
// with bias -> squeeze -> reduce_mean over T'.  Every stage is linear, so
//     logits[b,n] = bias[n] + sum_c W[c,n] * feat[b,c],   feat[b,c] = sum_{t,h,w} wt[t] * y[b,t,h,w,c]
// with wt[t] = (#pool windows covering frame t) / (2*7*7*T').  VideoResNet (AdaptiveAvgPool3d(1) +
// Linear, torchvision 0.5.0) is the same form with wt = 1/(T*H*W).  Backward:
//     gy[b,t,h,w,c] = wt[t] * sum_n dlogits[b,n] W[c,n], masked by y > 0 (ReluGrad of the producer).
//
// Loss head (kinetics_i3d_utils.py:152-169,253-307; model.py:177-250): one workgroup per clip computes
// softmax, label / max-non-label statistics, the adversarial loss and d(loss)/d(logits) in closed form.
#include "flk_internal.h"

template <typename T> __device__ static inline float ldf(const char* p, size_t i);
template <> __device__ inline float ldf<float>(const char* p, size_t i) { return ((const float*)p)[i]; }
template <> __device__ inline float ldf<bf16_t>(const char* p, size_t i) {
  return __uint_as_float((uint32_t)((const uint16_t*)p)[i] << 16);
}
template <typename T> __device__ static inline void stf(char* p, size_t i, float v);
template <> __device__ inline void stf<float>(char* p, size_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ inline void stf<bf16_t>(char* p, size_t i, float v) { ((bf16_t*)p)[i] = (bf16_t)v; }

// feat[b,c] = sum_pos wt[t(pos)] * y[b,pos,c]      grid (C/64, B); 4 position groups x 64 channels per workgroup
template <typename T>
__global__ __launch_bounds__(256) void head_pool_kernel(const char* y, int ld, int coff, int C, int Tn, int HW,
                                                        const float* wt, float* feat) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
  float acc = 0.f;
  if (c < C) {
    const int npos = Tn * HW;
    int i = grp;
    for (; i + 28 < npos; i += 32) {          // 8 independent loads in flight (the loop is latency-bound)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ldf<T>(y, ((size_t)b * npos + i + 4 * u) * ld + coff + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += wt[(i + 4 * u) / HW] * v[u];
    }
    for (; i < npos; i += 4) acc += wt[i / HW] * ldf<T>(y, ((size_t)b * npos + i) * ld + coff + c);
  }
  part[grp][cl] = acc;
  __syncthreads();
  if (grp == 0 && c < C) feat[(size_t)b * C + c] = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
}

// logits[b,n] = bias[n] + sum_c feat[b,c] * W[c,n]     grid (ceil(N/64), B): 64 outputs x 16 channel groups per workgroup,
// 8 weight loads in flight per thread (the loop is latency-bound: W is read once, 1.6 MB for I3D)
__global__ __launch_bounds__(1024) void head_fc_kernel(const float* feat, const float* W, const float* bias, int C, int N,
                                                       float* logits) {
  __shared__ float sf[2048];
  __shared__ float part[16][64];
  const int nl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl, b = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 1024) sf[c] = feat[(size_t)b * C + c];
  __syncthreads();
  float acc = 0.f;
  if (n < N) {
    int c = grp;
    for (; c + 7 * 16 < C; c += 8 * 16) {
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(c + u * 16) * N + n];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += sf[c + u * 16] * w[u];
    }
    for (; c < C; c += 16) acc += sf[c] * W[(size_t)c * N + n];
  }
  part[grp][nl] = acc;
  __syncthreads();
  if (grp == 0 && n < N) {
    float v = bias ? bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += part[g][nl];
    logits[(size_t)b * N + n] = v;
  }
}

// dfeat[b,c] = sum_n dlogits[b,n] * W[c,n]      grid (ceil(C/16), B): a workgroup = 16 channels x 16 lanes along n, so the 16 lanes
// of a channel read 64 contiguous bytes of its weight row (the one-thread-per-channel form read rows 1600 bytes apart: 18 us);
// fixed summation order (lane-strided partial sums, then a fixed shuffle tree)
__global__ __launch_bounds__(256) void head_fc_bwd_kernel(const float* dlogits, const float* W, int C, int N, float* dfeat) {
  const int cl = threadIdx.x >> 4, nl = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + cl, b = blockIdx.y;
  __shared__ float sd[1024];
  for (int n = threadIdx.x; n < N; n += 256) sd[n] = dlogits[(size_t)b * N + n];
  __syncthreads();
  float acc = 0.f;
  if (c < C)
    for (int n = nl; n < N; n += 16) acc += sd[n] * W[(size_t)c * N + n];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
  if (c < C && nl == 0) dfeat[(size_t)b * C + c] = acc;
}

// gy[b,pos,c] = wt[t] * dfeat[b,c] * (y > 0): one thread = one position x EPL channels (16-byte accesses, 32-bit index arithmetic;
// the scalar form with 64-bit div / mod per element took 68 us for the 3.2 M elements of the I3D head)
template <typename T>
__global__ __launch_bounds__(256) void head_pool_bwd_kernel(const char* y, int ld, int coff, char* gy, int gld, int gcoff,
                                                            int C, int Tn, int HW, int B, const float* wt, const float* dfeat, int use_mask) {
  constexpr int EPL = 16 / (int)sizeof(T);
  const int ng = C / EPL;
  const unsigned total = (unsigned)(B * Tn * HW) * (unsigned)ng;
  for (unsigned gid = blockIdx.x * 256u + threadIdx.x; gid < total; gid += gridDim.x * 256u) {
    const unsigned pos = gid / (unsigned)ng, cg = gid - pos * (unsigned)ng;
    const unsigned bt = pos / (unsigned)HW, b = bt / (unsigned)Tn, t = bt - b * (unsigned)Tn;
    const int c = (int)cg * EPL;
    const float w = wt[t];
    float g[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) g[e] = w * dfeat[(size_t)b * C + c + e];
    if (use_mask) {
      const uint4 u = *(const uint4*)(y + ((size_t)pos * ld + coff + c) * sizeof(T));
      if constexpr (sizeof(T) == 2) {
        const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!(__uint_as_float(wv[i] << 16) > 0.f)) g[2 * i] = 0.f;
          if (!(__uint_as_float(wv[i] & 0xffff0000u) > 0.f)) g[2 * i + 1] = 0.f;
        }
      } else {
        const float f[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!(f[e] > 0.f)) g[e] = 0.f;
      }
    }
    if constexpr (sizeof(T) == 2) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)g[e];
      *(bf16x8*)(gy + ((size_t)pos * gld + gcoff + c) * 2) = o;
    } else {
      *(float4*)(gy + ((size_t)pos * gld + gcoff + c) * 4) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

int flk_head_forward(const void* y, int ld, int coff, int C, int B, int Tn, int HW, const float* wt, const float* W,
                     const float* bias, int N, float* feat, float* logits, int dtype, hipStream_t s) {
  FLK_REQUIRE(C <= 2048 && N <= 1024, "head: C<=2048, N<=1024 supported");
  dim3 g1((C + 63) / 64, B);
  if (dtype == FLK_BF16) FLK_LAUNCH_KERNEL(head_pool_kernel<bf16_t>, g1, dim3(256), 0, s, (const char*)y, ld, coff, C, Tn, HW, wt, feat);
  else FLK_LAUNCH_KERNEL(head_pool_kernel<float>, g1, dim3(256), 0, s, (const char*)y, ld, coff, C, Tn, HW, wt, feat);
  FLK_LAUNCH_KERNEL(head_fc_kernel, dim3((N + 63) / 64, B), dim3(1024), 0, s, feat, W, bias, C, N, logits);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

int flk_head_backward(const void* y, int ld, int coff, void* gy, int gld, int gcoff, int C, int B, int Tn, int HW,
                      const float* wt, const float* W, int N, const float* dlogits, float* dfeat, int use_mask, int dtype,
                      hipStream_t s) {
  const int epl = dtype == FLK_BF16 ? 8 : 4;
  FLK_REQUIRE(C % epl == 0 && ld % epl == 0 && coff % epl == 0 && gld % epl == 0 && gcoff % epl == 0 && N <= 1024 &&
              (long)B * Tn * HW * (C / epl) < (1l << 31), "head backward: channel counts / strides must be multiples of %d", epl);
  FLK_LAUNCH_KERNEL(head_fc_bwd_kernel, dim3((C + 15) / 16, B), dim3(256), 0, s, dlogits, W, C, N, dfeat);
  const long total = (long)B * Tn * HW * (C / epl);
  const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (dtype == FLK_BF16)
    FLK_LAUNCH_KERNEL(head_pool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const char*)y, ld, coff, (char*)gy, gld, gcoff, C, Tn, HW, B, wt, dfeat, use_mask);
  else
    FLK_LAUNCH_KERNEL(head_pool_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const char*)y, ld, coff, (char*)gy, gld, gcoff, C, Tn, HW, B, wt, dfeat, use_mask);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// ------------------------------------------------------------------------------------------------
// softmax + adversarial loss + dlogits.  One 256-thread workgroup per clip.
// d(loss_b)/dz_j = alpha [j==y] + beta [j==mz] + gy * p_y ([j==y] - p_j) + gP * p_mp ([j==mp] - p_j)
// with (alpha, beta, gy, gP) derived per variant below.
__device__ static inline void block_argmax(float v, int i, float* sv, int* si, float& ov, int& oi) {
  // first-index-wins argmax over the block (ties -> smallest index, like np.argmax / tf.reduce_max's value)
  const int tid = threadIdx.x;
  sv[tid] = v; si[tid] = i;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float a = sv[tid], b2 = sv[tid + s];
      const int ia = si[tid], ib = si[tid + s];
      if (b2 > a || (b2 == a && ib < ia)) { sv[tid] = b2; si[tid] = ib; }
    }
    __syncthreads();
  }
  ov = sv[0]; oi = si[0];
  __syncthreads();
}

__global__ __launch_bounds__(256) void softmax_adv_loss_kernel(const flk_loss_args a, const float* logits, const int64_t* labels,
                                                               float* softmax, float* dlogits, float* per_clip) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ float coef[8];
  const int b = blockIdx.x, tid = threadIdx.x, C = a.C;
  const float* z = logits + (size_t)b * C;
  // a label outside [0, C) must not index z[]: clamp it for the reads and poison this clip's outputs with NaN so that the
  // error is visible in the loss instead of being a silent out-of-bounds read (the host wrapper validates the range too)
  const int64_t y_raw = labels[b];
  const bool y_bad = y_raw < 0 || y_raw >= (int64_t)C;
  const int y = y_bad ? 0 : (int)y_raw;
  constexpr int PER = 4;  // C <= 1024
  float zl[PER], pl[PER];
  float mx = -INFINITY; int mxi = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int j = tid + 256 * k;
    zl[k] = j < C ? z[j] : -INFINITY;
    if (zl[k] > mx) { mx = zl[k]; mxi = j; }
  }
  float zmax; int amax;
  block_argmax(mx, mxi, sv, si, zmax, amax);
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) { pl[k] = tid + 256 * k < C ? __expf(zl[k] - zmax) : 0.f; se += pl[k]; }
  sv[tid] = se;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) sv[tid] += sv[tid + s]; __syncthreads(); }
  const float inv = 1.f / sv[0];
  __syncthreads();
  // max non-label prob / logit.  TF dialect: max_k(v_k - onehot_k) (label NOT excluded);
  // torch dialect: true exclusion (model.py:218-219,235).
  float bp = -INFINITY, bz = -INFINITY; int bpi = 0x7fffffff, bzi = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int j = tid + 256 * k;
    pl[k] *= inv;
    if (j >= C) continue;
    float vp = pl[k], vz = zl[k];
    if (j == y) {
      if (a.torch_dialect) { vp = -INFINITY; vz = -INFINITY; } else { vp -= 1.f; vz -= 1.f; }
    }
    if (vp > bp) { bp = vp; bpi = j; }
    if (vz > bz) { bz = vz; bzi = j; }
  }
  float P, Z; int mp, mz;
  block_argmax(bp, bpi, sv, si, P, mp);
  block_argmax(bz, bzi, sv, si, Z, mz);
  // NaN logits never win a comparison, so the arg-max indices keep their "none" value: they must not index z[] (a NaN anywhere in
  // the network -- corrupt weights -- would otherwise end in a memory fault here instead of in a NaN loss)
  if ((unsigned)mp >= (unsigned)C) mp = 0;
  if ((unsigned)mz >= (unsigned)C) mz = 0;
  if ((unsigned)amax >= (unsigned)C) amax = 0;
  if (tid == 0) {
    const float zy = z[y], py = __expf(zy - zmax) * inv;
    const float Pm = __expf(z[mp] - zmax) * inv;   // p at the arg of the max (P may carry the TF "-1")
    float alpha = 0.f, beta = 0.f, gy = 0.f, gP = 0.f, loss = 0.f;
    const float mg = a.margin;
    if (a.improve_loss) {
      float u, M, dM_dpy = 0.f, dM_dP = 0.f, su_y = 0.f, su_m = 0.f, sp_y = 0.f, sp_P = 0.f;
      // u = to_min - to_max + M ; record d(u)/d(z_y), d(u)/d(z_mz), d(u)/d(p_y), d(u)/d(P) excluding M
      if (!a.use_logits) {
        M = mg;
        if (!a.targeted) { u = py - P + M; sp_y = 1.f; sp_P = -1.f; }
        else { u = P - py + M; sp_y = -1.f; sp_P = 1.f; }
      } else if (!a.targeted) {
        const float q = a.torch_dialect ? py : P;          // model.py:236 uses the LABEL prob
        M = logf(1.f + mg / (1e-5f + q));
        const float dM = -mg / ((1e-5f + q) * (1e-5f + q + mg));
        if (a.torch_dialect) dM_dpy = dM; else dM_dP = dM;
        u = zy - Z + M; su_y = 1.f; su_m = -1.f;
      } else {                                             // TF targeted, logits (kinetics_i3d_utils.py:256-259)
        M = logf(1.f + mg / py);
        dM_dpy = -mg / (py * (py + mg));
        u = Z - zy + M; su_y = -1.f; su_m = 1.f;
      }
      float dl_du = 0.f, dl_dM = 0.f;
      if (u > 0.f) {
        if (u * u / M <= u) { loss = u * u / M; dl_du = 2.f * u / M; dl_dM = -u * u / (M * M); }
        else { loss = u; dl_du = 1.f; }
      }
      const float cM = dl_du + dl_dM;                      // total derivative through M
      alpha = dl_du * su_y; beta = dl_du * su_m;
      gy = dl_du * sp_y + cM * dM_dpy;
      gP = dl_du * sp_P + cM * dM_dP;
    } else {
      const float ms = a.mean_scale;
      if (!a.targeted) { loss = -logf(1.f - py + 1e-6f) * ms; gy = ms / (1.f - py + 1e-6f); }
      else if (a.torch_dialect) { loss = -logf(py + 1e-6f) * ms; gy = -ms / (py + 1e-6f); }
      else { loss = -(zy - zmax - logf(1.f / inv)) * ms; gy = -ms / py; }
    }
    if (y_bad) { loss = NAN; alpha = NAN; gy = NAN; }
    coef[0] = alpha; coef[1] = beta; coef[2] = gy * py; coef[3] = gP * Pm;
    per_clip[b * 4 + 0] = loss; per_clip[b * 4 + 1] = py; per_clip[b * 4 + 2] = Pm; per_clip[b * 4 + 3] = (float)amax;
  }
  __syncthreads();
  const float alpha = coef[0], beta = coef[1], cy = coef[2], cP = coef[3];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int j = tid + 256 * k;
    if (j >= C) continue;
    float d = -(cy + cP) * pl[k];
    if (j == y) d += alpha + cy;
    if (j == mz) d += beta;
    if (j == mp) d += cP;
    if (softmax) softmax[(size_t)b * C + j] = pl[k];
    dlogits[(size_t)b * C + j] = d;
  }
}

extern "C" int flk_softmax_adv_loss(const flk_loss_args* a, const float* logits, const int64_t* labels, float* softmax,
                                    float* dlogits, float* per_clip, void* stream) {
  FLK_REQUIRE(a && logits && labels && dlogits && per_clip, "flk_softmax_adv_loss: null argument");
  FLK_REQUIRE(a->B > 0 && a->C > 1 && a->C <= 1024, "flk_softmax_adv_loss: need 1 < C <= 1024");
  FLK_REQUIRE(!(a->torch_dialect && a->improve_loss && a->targeted),
              "flk_softmax_adv_loss: the reference's targeted improve-loss is non-functional in the torch dialect "
              "(model.py:223-225 references undefined names); refusing to guess");
  FLK_REQUIRE(a->margin > 0.f || !a->improve_loss, "flk_softmax_adv_loss: margin must be > 0");
  FLK_LAUNCH_KERNEL(softmax_adv_loss_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a, logits, labels, softmax, dlogits, per_clip);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

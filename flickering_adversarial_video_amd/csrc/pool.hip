// tf.nn.max_pool3d (SAME) forward and MaxPool3DGrad, channels-last, 16-byte vectorised over channels.
// HBM-bound kernels: one thread = one position x one 16-byte channel group.
//   forward : first maximum in (t,h,w) scan order wins (strict >), padded cells never win
//             (i3d.py:174,189,212,252,398); the winning window index is kept as one byte.
//   backward: default = scatter form (maxpool_scatter_bwd): a workgroup owns a tile of INPUT cells in LDS and every
//             window that reaches the tile adds its gradient to the cell its saved argmax names.  bf16 mode sums in
//             32-bit fixed point with integer LDS atomics (order-independent, bitwise reproducible).  fp32 (the parity mode)
//             takes the gather forms (every input cell scans the windows containing it; fixed summation order), as does
//             FLK_POOL_GATHER=1 in bf16; the scatter form in fp32 (float LDS atomics, last-ulp run-to-run differences) only
//             with FLK_POOL_SCATTER_F32=1.  Optional relu mask of the producing layer (mask > 0) fused in.
#include <stdlib.h>
#include <string.h>
#include "flk_internal.h"

// Timing ablations (skip loads / atomics / stores: WRONG results) exist only in -DFLK_ABLATE builds (tools/*_time.py pass it through
// FLK_HIPCC_EXTRA); in the product library the switches are compile-time zeros and the kernels carry no such branch.
#ifdef FLK_ABLATE
#define PF_DBG(bit) (p.dbg & (bit))
#define PG_DBG(bit) (pg.dbg & (bit))
#else
#define PF_DBG(bit) (0)
#define PG_DBG(bit) (0)
#endif

struct PoolKP {
  const char* in; char* out; uint8_t* idx;
  const char* gout; char* gin; const char* mask; const char* add;
  int in_ld, in_coff, out_ld, out_coff, C;
  int gout_ld, gout_coff, gin_ld, gin_coff, mask_ld, mask_coff;
  int B, Ti, Hi, Wi, To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int relu_input;
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

// grid: x = chunks of 256 over (w, channel group) of one output row; y = h; z = b*T + t  (32-bit index math only)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const PoolKP p) {
  constexpr int EPL = PV<T>::EPL;
  const int ng = p.C / EPL;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)(p.Wo * ng)) return;
  const int ow = i / ng, cg = i - ow * ng;
  const int oh = blockIdx.y, ot = blockIdx.z % p.To, b = blockIdx.z / p.To;
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
  if (p.relu_input) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) bi[e] = best[e] > 0.f ? bi[e] : 255;
  }
  const size_t opos = (((size_t)(b * p.To + ot) * p.Ho + oh) * p.Wo + ow);
  PV<T>::st(p.out + (opos * p.out_ld + p.out_coff + cg * EPL) * sizeof(T), best);
  PV<T>::stidx(p.idx + opos * p.C + cg * EPL, bi);
}

// The same with the window as template arguments (the I3D pools: 1x3x3, 3x3x3, 2x2x2): the tap loop unrolls and ALL tap loads of a thread
// are requested before the first compare -- with run-time loop bounds hipcc emits load -> wait -> compare per tap, KT*KH*KW exposed memory
// round trips per thread.  Same scan order, same tie rule (strict >), same bits.
template <typename T, int KT, int KH, int KW>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel_k(const PoolKP p) {
  constexpr int EPL = PV<T>::EPL, NT = KT * KH * KW;
  const int ng = p.C / EPL;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)(p.Wo * ng)) return;
  const int ow = i / ng, cg = i - ow * ng;
  const int oh = blockIdx.y, ot = blockIdx.z % p.To, b = blockIdx.z / p.To;
  uint4 raw[NT];
  bool ok[NT];
#pragma unroll
  for (int dt = 0; dt < KT; ++dt)
#pragma unroll
    for (int dh = 0; dh < KH; ++dh)
#pragma unroll
      for (int dw = 0; dw < KW; ++dw) {
        const int tap = (dt * KH + dh) * KW + dw;
        const int it = ot * p.st - p.pt + dt, ih = oh * p.sh - p.ph + dh, iw = ow * p.sw - p.pw + dw;
        ok[tap] = (unsigned)it < (unsigned)p.Ti && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        const size_t pos = ok[tap] ? (((size_t)(b * p.Ti + it) * p.Hi + ih) * p.Wi + iw) : 0;      // (position 0 is always readable)
        if constexpr (sizeof(T) == 2) raw[tap] = *(const uint4*)(p.in + (pos * p.in_ld + p.in_coff + cg * EPL) * sizeof(T));
        else raw[tap] = *(const uint4*)(p.in + (pos * p.in_ld + p.in_coff + cg * EPL) * sizeof(T));
      }
  float best[EPL];
  int bi[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) {
    float v[EPL];
    if constexpr (sizeof(T) == 2) {
      const uint32_t w[4] = {raw[tap].x, raw[tap].y, raw[tap].z, raw[tap].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    } else {
      v[0] = __uint_as_float(raw[tap].x); v[1] = __uint_as_float(raw[tap].y); v[2] = __uint_as_float(raw[tap].z); v[3] = __uint_as_float(raw[tap].w);
    }
    if (ok[tap]) {
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (v[e] > best[e]) { best[e] = v[e]; bi[e] = tap; }
    }
  }
  if (p.relu_input) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) bi[e] = best[e] > 0.f ? bi[e] : 255;
  }
  const size_t opos = (((size_t)(b * p.To + ot) * p.Ho + oh) * p.Wo + ow);
  PV<T>::st(p.out + (opos * p.out_ld + p.out_coff + cg * EPL) * sizeof(T), best);
  PV<T>::stidx(p.idx + opos * p.C + cg * EPL, bi);
}

// grid: x = chunks of 256 over (w, channel group) of one INPUT row; y = h; z = b*T + t
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolKP p) {
  constexpr int EPL = PV<T>::EPL;
  const int ng = p.C / EPL;
  {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)(p.Wi * ng)) return;
    const int iw = i / ng, cg = i - iw * ng;
    const int ih = blockIdx.y, it = blockIdx.z % p.Ti, b = blockIdx.z / p.Ti;
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

// Backward of the STRIDED pools (MaxPool3d_2a/3a/4a/5a, i3d.py:174,189,252,398; window <= 2 * stride; SAME pad-before p < stride per
// dimension: 0 on even sizes, 1 on the odd ones the reference's 90-frame clips produce -- T/2 = 45, 23):
// one thread = one output window x one 16-byte channel group, and it OWNS the stride^3 input cells o*s - p + a, a in [0,s).
// A cell is covered by its own window (tap a) and, where a + s < k, by the previous window of that dimension (tap
// a + s): every thread loads the index bytes and gradients of its <= 8 candidate windows up front (independent loads,
// shared with its neighbours through L1/L2), then resolves its cells in registers in a fixed order -- no atomics,
// no dependent load chains, and each gradient cell is written exactly once (64-byte runs per thread).
template <typename T, int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool_strided_bwd(const PoolKP p) {
  constexpr int EPL = PV<T>::EPL;
  constexpr int NT = KT > ST ? 2 : 1, NH = KH > SH ? 2 : 1, NW = KW > SW ? 2 : 1;     // candidate windows per dimension
  static_assert(KT <= 2 * ST && KH <= 2 * SH && KW <= 2 * SW, "a cell may see at most two windows per dimension");
  const int ng = p.C / EPL;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)(p.Wo * ng)) return;
  const int ow = i / ng, cg = i - ow * ng;
  const int oh = blockIdx.y, ot = blockIdx.z % p.To, b = blockIdx.z / p.To;
  int id[NT * NH * NW][EPL];
  float go[NT * NH * NW][EPL];
  bool ok[NT * NH * NW];
#pragma unroll
  for (int dt = 0; dt < NT; ++dt)
#pragma unroll
    for (int dh = 0; dh < NH; ++dh)
#pragma unroll
      for (int dw = 0; dw < NW; ++dw) {
        const int w = (dt * NH + dh) * NW + dw;
        ok[w] = ot - dt >= 0 && oh - dh >= 0 && ow - dw >= 0;
        const size_t opos = (((size_t)(b * p.To + max(ot - dt, 0)) * p.Ho + max(oh - dh, 0)) * p.Wo + max(ow - dw, 0));
        PV<T>::ldidx(p.idx + opos * p.C + cg * EPL, id[w]);
        PV<T>::ld(p.gout + (opos * p.gout_ld + p.gout_coff + cg * EPL) * sizeof(T), go[w]);
      }
  // the ReLU-mask operand of every owned cell, requested with the loads above (inside the cell loop each was load -> wait -> store):
  // 1x3x3 / 2 (4 cells) 57.9 -> 54.4 us; with 8 cells the registers cost more than the round trips (3x3x3 / 2: 64.1 -> 67.6 us): not there
  constexpr bool HOIST = ST * SH * SW <= 4;
  float mk[HOIST ? ST * SH * SW : 1][EPL];
  if (HOIST && p.mask) {
#pragma unroll
    for (int at = 0; at < ST; ++at)
#pragma unroll
      for (int ah = 0; ah < SH; ++ah)
#pragma unroll
        for (int aw = 0; aw < SW; ++aw) {
          const int it = min(max(ot * ST - p.pt + at, 0), p.Ti - 1), ih = min(max(oh * SH - p.ph + ah, 0), p.Hi - 1), iw = min(max(ow * SW - p.pw + aw, 0), p.Wi - 1);
          const size_t ipos = (((size_t)(b * p.Ti + it) * p.Hi + ih) * p.Wi + iw);
          PV<T>::ld(p.mask + (ipos * p.mask_ld + p.mask_coff + cg * EPL) * sizeof(T), mk[(at * SH + ah) * SW + aw]);
        }
  }
#pragma unroll
  for (int at = 0; at < ST; ++at)
#pragma unroll
    for (int ah = 0; ah < SH; ++ah)
#pragma unroll
      for (int aw = 0; aw < SW; ++aw) {
        const int it = ot * ST - p.pt + at, ih = oh * SH - p.ph + ah, iw = ow * SW - p.pw + aw;
        if ((unsigned)it >= (unsigned)p.Ti || (unsigned)ih >= (unsigned)p.Hi || (unsigned)iw >= (unsigned)p.Wi) continue;
        float g[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) g[e] = 0.f;
        // candidates in ascending window order (the order the simple gather kernel uses): previous window first
#pragma unroll
        for (int dt = NT - 1; dt >= 0; --dt)
#pragma unroll
          for (int dh = NH - 1; dh >= 0; --dh)
#pragma unroll
            for (int dw = NW - 1; dw >= 0; --dw) {
              const int tt = at + dt * ST, th = ah + dh * SH, tw = aw + dw * SW;
              if (tt >= KT || th >= KH || tw >= KW) continue;                    // compile-time after unrolling
              const int w = (dt * NH + dh) * NW + dw, tap = (tt * KH + th) * KW + tw;
              if (!ok[w]) continue;
#pragma unroll
              for (int e = 0; e < EPL; ++e) g[e] += id[w][e] == tap ? go[w][e] : 0.f;
            }
        const size_t ipos = (((size_t)(b * p.Ti + it) * p.Hi + ih) * p.Wi + iw);
        if (p.mask) {
          if (!HOIST) PV<T>::ld(p.mask + (ipos * p.mask_ld + p.mask_coff + cg * EPL) * sizeof(T), mk[0]);
#pragma unroll
          for (int e = 0; e < EPL; ++e) g[e] = mk[HOIST ? (at * SH + ah) * SW + aw : 0][e] > 0.f ? g[e] : 0.f;
        }
        PV<T>::st(p.gin + (ipos * p.gin_ld + p.gin_coff + cg * EPL) * sizeof(T), g);
      }
}

template <typename T, int KT, int KH, int KW, int ST, int SH, int SW>
static int launch_strided_bwd(const PoolKP& kp, const flk_pool_args* a, hipStream_t s) {
  constexpr int EPL = PV<T>::EPL;
  FLK_REQUIRE(a->Ho < 65536 && (long)a->B * a->To < 65536, "flk_maxpool3d_bwd: grid too large");
  const dim3 grid((unsigned)((a->Wo * (a->C / EPL) + 255) / 256), (unsigned)a->Ho, (unsigned)(a->B * a->To));
  FLK_LAUNCH_KERNEL((maxpool_strided_bwd<T, KT, KH, KW, ST, SH, SW>), grid, dim3(256), 0, s, kp);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// the owner form needs a pad-before below the stride and every input cell inside some window's stride box [o*s - p, o*s - p + s)
static bool strided_owner_ok(const flk_pool_args* a, int kt, int kh, int kw, int st, int sh, int sw) {
  return a->kt == kt && a->kh == kh && a->kw == kw && a->st == st && a->sh == sh && a->sw == sw && a->pt >= 0 && a->pt < st && a->ph >= 0 &&
         a->ph < sh && a->pw >= 0 && a->pw < sw && a->Ti <= a->To * st - a->pt && a->Hi <= a->Ho * sh - a->ph && a->Wi <= a->Wo * sw - a->pw;
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled variants for stride-1 SAME pooling with an odd window (the Inception branch-3 pool, i3d.py:212):
// every cell is touched by kt*kh*kw windows, so the simple kernels above read each byte 27x through L1/L2.
// Here a workgroup stages the halo box of ONE 64-byte channel slab once (same 4-plane LDS image as
// conv_igemm.hip) and all window taps are LDS reads.
struct PoolTP {
  PoolKP k;
  int Tt, Ht, Wt, nTt, nTh, nTw, Th, Hh, Wh, P, plane_b, rows, ntiles, nslab;
  int dbg;             // timing experiments only (FLK_PF_DBG, W-run forward): 1 = no halo loads, 2 = no column maxima, 4 = no stores
};

__device__ static inline int pplane_off(int c, int plane_b) { return c * plane_b + (c >> 1) * 32; }

// 1-D grid -> (tile, channel slab).  Workgroups are dealt to the 8 XCDs round-robin, so the slabs of ONE tile are put
// on consecutive slots of the SAME XCD: together they touch whole cache lines of every position while those lines
// are still in that XCD's L2 (a slab is only 64 B of a position's row).
// Tile i sits on XCD i % 8 (dealing the tiles to the XCDs in contiguous chunks, as the convolutions do, measured 6.81 against 6.77-6.79 ms
// per step: DESIGN_LOG.md).
__device__ static inline bool tile_slab_of_block(const PoolTP& p, int nslab, int& tile, int& slab) {
  const int bid = blockIdx.x, xcd = bid & 7, r = bid >> 3;
  slab = r % nslab;
  tile = (r / nslab) * 8 + xcd;                              // (the host sizes the grid as 8 * ceil(ntiles / 8) * nslab)
  return tile < p.ntiles;
}
__device__ static inline bool tile_slab_of_block(const PoolTP& p, int& tile, int& slab) { return tile_slab_of_block(p, p.nslab, tile, slab); }

// x / d for 0 <= x < 2^20 and a small run-time divisor d, inv = 1.0f / d: (x + 0.5) / d is never within 0.5 / d of an integer, so the float
// quotient truncates exactly -- 3 VALU operations instead of the ~40 of a 32-bit integer division (which made the index arithmetic of the
// staging / write-out loops below as expensive as the pooling itself)
__device__ static inline int qdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
__device__ static inline void split3(int x, int d1, float inv1, int d2, float inv2, int& a, int& b, int& c) {   // x = (a * d1) + b * d2 + c, d1 = (rows of b) * d2
  a = qdiv(x, inv1); const int rem = x - a * d1; b = qdiv(rem, inv2); c = rem - b * d2;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void maxpool_s1_tiled_fwd(const PoolTP p) {
  constexpr int EPL = PV<T>::EPL, SLABC = 4 * EPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PoolKP& k = p.k;
  const int tid = threadIdx.x, ch = tid & 3;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int c0 = cslab * SLABC + ch * EPL;
  const bool chvalid = c0 < k.C;
  const int ot0 = tt * p.Tt, oh0 = th * p.Ht, ow0 = tw * p.Wt;
  const int it0 = ot0 - k.pt, ih0 = oh0 - k.ph, iw0 = ow0 - k.pw;
  const int HW = p.Hh * p.Wh;
  // stage the halo (-inf outside the tensor: padded cells never win)
  for (int hp = tid >> 2; hp < p.P; hp += 64) {
    int a, bq, c; split3(hp, HW, 1.0f / (float)HW, p.Wh, 1.0f / (float)p.Wh, a, bq, c);
    const int it = it0 + a, ih = ih0 + bq, iw = iw0 + c;
    float v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = -INFINITY;
    if (chvalid && (unsigned)it < (unsigned)k.Ti && (unsigned)ih < (unsigned)k.Hi && (unsigned)iw < (unsigned)k.Wi)
      PV<T>::ld(k.in + ((((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw) * k.in_ld + k.in_coff + c0) * sizeof(T), v);
    PV<T>::st(smem + pplane_off(ch, p.plane_b) + hp * 16, v);
  }
  __syncthreads();
  if (!chvalid) return;
  const int hw = p.Ht * p.Wt;
  for (int r = tid >> 2; r < p.rows; r += 64) {
    int rt, rh, rw; split3(r, hw, 1.0f / (float)hw, p.Wt, 1.0f / (float)p.Wt, rt, rh, rw);
    const int ot = ot0 + rt, oh = oh0 + rh, ow = ow0 + rw;
    if (ot >= k.To || oh >= k.Ho || ow >= k.Wo) continue;
    const char* base = smem + pplane_off(ch, p.plane_b) + ((rt * p.Hh + rh) * p.Wh + rw) * 16;
    float best[EPL];
    int bi[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    int tap = 0;
    for (int dt = 0; dt < k.kt; ++dt)
      for (int dh = 0; dh < k.kh; ++dh)
        for (int dw = 0; dw < k.kw; ++dw, ++tap) {
          float v[EPL];
          PV<T>::ld(base + ((dt * p.Hh + dh) * p.Wh + dw) * 16, v);
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            if (v[e] > best[e]) { best[e] = v[e]; bi[e] = tap; }
        }
    if (k.relu_input) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) bi[e] = best[e] > 0.f ? bi[e] : 255;
    }
    const size_t opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
    PV<T>::st(k.out + (opos * k.out_ld + k.out_coff + c0) * sizeof(T), best);
    PV<T>::stidx(k.idx + opos * k.C + c0, bi);
  }
}

// bf16 fast path of the tiled forward: the halo holds SORTABLE 16-bit keys (key = bits ^ (sign ? 0xffff : 0x8000): unsigned
// order == float order); one element-tap costs 2 VALU ops: pack (key << 16 | 255 - tap) and v_max_u32 -- the larger low
// byte wins among equal keys, i.e. the FIRST maximum in scan order.  Taps are compile-time so the LDS reads batch.
__device__ static inline uint32_t bf16x2_to_keys(uint32_t w) {
  const uint32_t sign = w & 0x80008000u;                 // per half: 0x8000 if negative
  const uint32_t flip = (sign >> 15) * 0x7fffu;          // 0x7fff for negative halves
  return w ^ (flip | 0x80008000u);                       // negative: ^0xffff, non-negative: ^0x8000
}
__device__ static inline uint32_t keys_to_bf16x2(uint32_t k) {
  const uint32_t neg = (~k) & 0x80008000u;               // original sign set <=> key top bit clear
  const uint32_t flip = (neg >> 15) * 0x7fffu;
  return k ^ (flip | 0x80008000u);
}

template <int KT, int KH, int KW>
__global__ __launch_bounds__(256, 2) void maxpool_s1_tiled_fwd_bf16(const PoolTP p) {
  constexpr int EPL = 8, SLABC = 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PoolKP& k = p.k;
  const int tid = threadIdx.x, ch = tid & 3;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int c0 = cslab * SLABC + ch * EPL;
  const bool chvalid = c0 < k.C;
  const int ot0 = tt * p.Tt, oh0 = th * p.Ht, ow0 = tw * p.Wt;
  const int it0 = ot0 - k.pt, ih0 = oh0 - k.ph, iw0 = ow0 - k.pw;
  const int HW = p.Hh * p.Wh;
  for (int hp = tid >> 2; hp < p.P; hp += 64) {
    int a, bq, c; split3(hp, HW, 1.0f / (float)HW, p.Wh, 1.0f / (float)p.Wh, a, bq, c);
    const int it = it0 + a, ih = ih0 + bq, iw = iw0 + c;
    uint4 v = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);     // -inf
    if (chvalid && (unsigned)it < (unsigned)k.Ti && (unsigned)ih < (unsigned)k.Hi && (unsigned)iw < (unsigned)k.Wi)
      v = *(const uint4*)(k.in + ((((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw) * k.in_ld + k.in_coff + c0) * 2);
    v.x = bf16x2_to_keys(v.x); v.y = bf16x2_to_keys(v.y); v.z = bf16x2_to_keys(v.z); v.w = bf16x2_to_keys(v.w);
    *(uint4*)(smem + pplane_off(ch, p.plane_b) + hp * 16) = v;
  }
  __syncthreads();
  if (!chvalid) return;
  const int hw = p.Ht * p.Wt;
  for (int r = tid >> 2; r < p.rows; r += 64) {
    int rt, rh, rw; split3(r, hw, 1.0f / (float)hw, p.Wt, 1.0f / (float)p.Wt, rt, rh, rw);
    const int ot = ot0 + rt, oh = oh0 + rh, ow = ow0 + rw;
    if (ot >= k.To || oh >= k.Ho || ow >= k.Wo) continue;
    const char* base = smem + pplane_off(ch, p.plane_b) + ((rt * p.Hh + rh) * p.Wh + rw) * 16;
    uint32_t best[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dt = 0; dt < KT; ++dt)
#pragma unroll
      for (int dh = 0; dh < KH; ++dh)
#pragma unroll
        for (int dw = 0; dw < KW; ++dw) {
          const uint32_t tag = 255u - (uint32_t)((dt * KH + dh) * KW + dw);
          const uint4 v = *(const uint4*)(base + ((dt * p.Hh + dh) * p.Wh + dw) * 16);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            best[2 * i] = max(best[2 * i], (w[i] << 16) | tag);
            best[2 * i + 1] = max(best[2 * i + 1], (w[i] & 0xffff0000u) | tag);
          }
        }
    uint4 o;
    o.x = keys_to_bf16x2((best[0] >> 16) | (best[1] & 0xffff0000u));
    o.y = keys_to_bf16x2((best[2] >> 16) | (best[3] & 0xffff0000u));
    o.z = keys_to_bf16x2((best[4] >> 16) | (best[5] & 0xffff0000u));
    o.w = keys_to_bf16x2((best[6] >> 16) | (best[7] & 0xffff0000u));
    if (k.relu_input) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((best[e] >> 16) <= 0x8000u) best[e] &= ~255u;      // tag 0 -> idx 255 ("no cell")
    }
    uint2 id;
    id.x = (255u - (best[0] & 255u)) | ((255u - (best[1] & 255u)) << 8) | ((255u - (best[2] & 255u)) << 16) | ((255u - (best[3] & 255u)) << 24);
    id.y = (255u - (best[4] & 255u)) | ((255u - (best[5] & 255u)) << 8) | ((255u - (best[6] & 255u)) << 16) | ((255u - (best[7] & 255u)) << 24);
    const size_t opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
    *(uint4*)(k.out + (opos * k.out_ld + k.out_coff + c0) * 2) = o;
    *(uint2*)(k.idx + opos * k.C + c0) = id;
  }
}

// W-run form of the bf16 3x3x3 forward: a thread owns one (t, h) of the tile, 8 channels and the whole run of WT outputs along w.  The
// 9-tap maximum over (dt, dh) of a halo column is computed ONCE and shared by the three outputs whose windows contain it -- (WT+2) * 9
// element-taps per WT outputs instead of WT * 27 (2.1x fewer VALU operations at WT = 7; the pool is VALU-bound).  Tie rule unchanged:
// the column stage tags a candidate with 255 - 3*(3 dt + dh), the combine stage subtracts dw, so the low byte is 255 - tap and the
// larger one (first maximum in scan order) wins among equal keys.
// plane of channel chunk c in the W-run kernel's halo image: 64 bytes of skew per plane.  gfx950 serves a ds_read_b128 / ds_write_b128 16
// lanes at a time (16 x 16 bytes = the 64 banks) and in one pass iff the lanes' 16-byte units differ mod 16; the 16 lanes are 4 (t, h)
// pairs x 4 channel chunks, the pairs' slots one halo row (WT + 2 = 9: odd) apart, so units 4 c + 9 j are all different.  The 32 bytes
// per plane PAIR of pplane_off put chunks 0 / 1 and 2 / 3 on the same banks: rocprofv3 counted SQ_LDS_BANK_CONFLICT = 65 % of this
// kernel's SQ_LDS_IDX_ACTIVE (four passes per read), as many LDS cycles per column as its VALU cycles.
__device__ static inline int wplane_off(int c, int plane_b) { return c * (plane_b + 64); }
template <int WT>
__global__ __launch_bounds__(256, 2) void maxpool_s1_wrun_fwd_bf16(const PoolTP p) {
  constexpr int EPL = 8, SLABC = 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PoolKP& k = p.k;
  const int tid = threadIdx.x, ch = tid & 3;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int c0 = cslab * SLABC + ch * EPL;
  const bool chvalid = c0 < k.C;
  const int ot0 = tt * p.Tt, oh0 = th * p.Ht, ow0 = tw * WT;
  const int it0 = ot0 - 1, ih0 = oh0 - 1, iw0 = ow0 - 1;
  constexpr int WH = WT + 2;                                   // = p.Wh
  const int HW = p.Hh * WH;
  const float inv_HW = 1.0f / (float)HW;                       // (hp + 0.5) / HW is never within 0.5 / HW of an integer: the float quotient truncates exactly
  // staging, SU halo positions per thread and pass (the whole halo in one pass: P <= 1008): the loads are requested together (as a
  // plain loop hipcc emitted load -> wait -> convert -> ds_write per position: 14 exposed memory round trips per workgroup)
  constexpr int SU = 16;
  for (int hp0 = tid >> 2; hp0 < p.P; hp0 += 64 * SU) {
    uint4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int hp = hp0 + 64 * u;
      const int a = (int)(((float)hp + 0.5f) * inv_HW), rem = hp - a * HW, bq = rem / WH, c = rem - bq * WH;
      const int it = it0 + a, ih = ih0 + bq, iw = iw0 + c;
      v[u] = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);     // -inf
      if (hp < p.P && chvalid && (unsigned)it < (unsigned)k.Ti && (unsigned)ih < (unsigned)k.Hi && (unsigned)iw < (unsigned)k.Wi && !PF_DBG(1))
        v[u] = *(const uint4*)(k.in + ((((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw) * k.in_ld + k.in_coff + c0) * 2);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int hp = hp0 + 64 * u;
      if (hp >= p.P) break;
      uint4 w = v[u];
      w.x = bf16x2_to_keys(w.x); w.y = bf16x2_to_keys(w.y); w.z = bf16x2_to_keys(w.z); w.w = bf16x2_to_keys(w.w);
      *(uint4*)(smem + wplane_off(ch, p.plane_b) + hp * 16) = w;
    }
  }
  __syncthreads();
  if (!chvalid) return;
  const int npairs = p.Tt * p.Ht;
  for (int pr = tid >> 2; pr < npairs; pr += 64) {
    const int rt = qdiv(pr, 1.0f / (float)p.Ht), rh = pr - rt * p.Ht;
    const int ot = ot0 + rt, oh = oh0 + rh;
    if (ot >= k.To || oh >= k.Ho) continue;
    const char* base = smem + wplane_off(ch, p.plane_b) + ((rt * p.Hh + rh) * WH) * 16;
    uint32_t cm[3][8];
    auto colmax = [&](int c, uint32_t (&m)[8]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = 0;
      uint4 tapv[9];                                  // the 9 reads of the column are requested together
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) tapv[dt * 3 + dh] = *(const uint4*)(base + ((dt * p.Hh + dh) * WH + c) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
          // tag = 27 - tap of (dt, dh, dw = 0): small enough for an inline constant, so the high-half key is ONE v_and_or_b32
          // (with 255 - tap the mask and the tag were two literals: v_and + v_or)
          const uint32_t tag = 27u - 3u * (uint32_t)(dt * 3 + dh);
          const uint4 v = tapv[dt * 3 + dh];
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            m[2 * i] = max(m[2 * i], (w[i] << 16) | tag);
            m[2 * i + 1] = max(m[2 * i + 1], (w[i] & 0xffff0000u) | tag);
          }
        }
    };
    if PF_DBG(2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cm[0][e] = cm[1][e] = cm[2][e] = (unsigned)(tid + e) << 8;
    }
    if (!PF_DBG(2)) { colmax(0, cm[0]); colmax(1, cm[1]); }
#pragma unroll
    for (int rw = 0; rw < WT; ++rw) {
      if (!PF_DBG(2)) colmax(rw + 2, cm[(rw + 2) % 3]);
      const int ow = ow0 + rw;
      if (ow >= k.Wo) continue;
      const uint32_t (&l)[8] = cm[rw % 3];
      const uint32_t (&m)[8] = cm[(rw + 1) % 3];
      const uint32_t (&r)[8] = cm[(rw + 2) % 3];
      uint32_t best[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = max(max(l[e], m[e] - 1u), r[e] - 2u);
      uint4 o;
      o.x = keys_to_bf16x2((best[0] >> 16) | (best[1] & 0xffff0000u));
      o.y = keys_to_bf16x2((best[2] >> 16) | (best[3] & 0xffff0000u));
      o.z = keys_to_bf16x2((best[4] >> 16) | (best[5] & 0xffff0000u));
      o.w = keys_to_bf16x2((best[6] >> 16) | (best[7] & 0xffff0000u));
      if (k.relu_input) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((best[e] >> 16) <= 0x8000u) best[e] &= ~255u;      // tag 0 -> idx 255 ("no cell")
      }
      uint32_t ix[8];                                                // tag 27 - tap -> tap; tag 0 -> 255 ("no cell")
#pragma unroll
      for (int e = 0; e < 8; ++e) { const uint32_t tg = best[e] & 255u; ix[e] = tg ? 27u - tg : 255u; }
      uint2 id;
      id.x = ix[0] | (ix[1] << 8) | (ix[2] << 16) | (ix[3] << 24);
      id.y = ix[4] | (ix[5] << 8) | (ix[6] << 16) | (ix[7] << 24);
      const size_t opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
      if PF_DBG(4) { if (o.x == 0x12345u && id.x == 0x777u) *(uint2*)(k.idx + opos * k.C + c0) = id; continue; }
      *(uint4*)(k.out + (opos * k.out_ld + k.out_coff + c0) * 2) = o;
      *(uint2*)(k.idx + opos * k.C + c0) = id;
    }
  }
}

// tile of the W-run kernel: WT outputs along w, (Tt, Ht) with Tt*Ht <= 64 (one (t,h) pair per thread quad) and a halo <= 1008 positions
template <int WT>
static int launch_wrun_fwd(const PoolKP& kp, const flk_pool_args* a, hipStream_t s) {
  PoolTP tp{};
  tp.k = kp;
  int bestT = 1, bestH = 1; double bestScore = -1.0;
  for (int Tt = 1; Tt <= a->To && Tt <= 8; ++Tt)
    for (int Ht = 1; Ht <= a->Ho && Tt * Ht <= 64; ++Ht) {
      const long halo = (long)(Tt + 2) * (Ht + 2) * (WT + 2);
      if (halo > FLK_MAX_HALO) continue;
      const long tiles = (long)((a->To + Tt - 1) / Tt) * ((a->Ho + Ht - 1) / Ht);
      // useful outputs per staged halo position, discounted by idle thread quads (pairs < 64) and ragged edge tiles
      const double eff = (double)a->To * a->Ho / ((double)tiles * Tt * Ht);
      const double score = eff * (double)(Tt * Ht * WT) / (double)halo * (Tt * Ht >= 48 ? 1.0 : (double)(Tt * Ht) / 48.0);
      if (score > bestScore) { bestScore = score; bestT = Tt; bestH = Ht; }
    }
  tp.Tt = bestT; tp.Ht = bestH; tp.Wt = WT; tp.rows = bestT * bestH * WT;
  tp.nTt = (a->To + tp.Tt - 1) / tp.Tt; tp.nTh = (a->Ho + tp.Ht - 1) / tp.Ht; tp.nTw = (a->Wo + WT - 1) / WT;
  tp.Th = tp.Tt + 2; tp.Hh = tp.Ht + 2; tp.Wh = WT + 2;
  tp.P = tp.Th * tp.Hh * tp.Wh;
  tp.plane_b = (tp.P * 16 + 255) / 256 * 256;
  const size_t lds = 4 * (size_t)tp.plane_b + 256;               // (wplane_off: 64 bytes of skew per plane)
  tp.ntiles = a->B * tp.nTt * tp.nTh * tp.nTw; tp.nslab = (a->C + 31) / 32;
#ifdef FLK_ABLATE
  tp.dbg = flk_ablate_env("FLK_PF_DBG");
#endif
  const dim3 grid((unsigned)((tp.ntiles + 7) / 8 * 8 * tp.nslab));
  static bool attr_set[FLK_MAX_DEVICES] = {};
  if (int rc = flk_raise_lds_limit((const void*)maxpool_s1_wrun_fwd_bf16<WT>, 96 * 1024, attr_set)) return rc;
  FLK_LAUNCH_KERNEL((maxpool_s1_wrun_fwd_bf16<WT>), grid, dim3(256), lds, s, tp);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void maxpool_s1_tiled_bwd(const PoolTP p) {
  constexpr int EPL = PV<T>::EPL, SLABC = 4 * EPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PoolKP& k = p.k;
  char* const sidx = smem + 4 * p.plane_b + 64;            // [chunk][halo position][EPL bytes]
  const int tid = threadIdx.x, ch = tid & 3;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int c0 = cslab * SLABC + ch * EPL;
  const bool chvalid = c0 < k.C;
  // tile of INPUT cells [i0, i0+tile); windows o with o - pad <= i <= o - pad + k-1  ->  o in [i - (k-1-pad), i + pad]
  const int i_t0 = tt * p.Tt, i_h0 = th * p.Ht, i_w0 = tw * p.Wt;
  const int o_t0 = i_t0 - (k.kt - 1 - k.pt), o_h0 = i_h0 - (k.kh - 1 - k.ph), o_w0 = i_w0 - (k.kw - 1 - k.pw);
  const int HW = p.Hh * p.Wh;
  for (int hp = tid >> 2; hp < p.P; hp += 64) {
    int a, bq, c; split3(hp, HW, 1.0f / (float)HW, p.Wh, 1.0f / (float)p.Wh, a, bq, c);
    const int ot = o_t0 + a, oh = o_h0 + bq, ow = o_w0 + c;
    uint4 g = make_uint4(0, 0, 0, 0);
    int id[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) id[e] = 255;                // never matches a tap
    if (chvalid && (unsigned)ot < (unsigned)k.To && (unsigned)oh < (unsigned)k.Ho && (unsigned)ow < (unsigned)k.Wo) {
      const size_t opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
      g = *(const uint4*)(k.gout + (opos * k.gout_ld + k.gout_coff + c0) * sizeof(T));
      PV<T>::ldidx(k.idx + opos * k.C + c0, id);
    }
    *(uint4*)(smem + pplane_off(ch, p.plane_b) + hp * 16) = g;
    PV<T>::stidx((uint8_t*)sidx + ((size_t)ch * p.P + hp) * EPL, id);
  }
  __syncthreads();
  if (!chvalid) return;
  const int hw = p.Ht * p.Wt;
  for (int r = tid >> 2; r < p.rows; r += 64) {
    int rt, rh, rw; split3(r, hw, 1.0f / (float)hw, p.Wt, 1.0f / (float)p.Wt, rt, rh, rw);
    const int it = i_t0 + rt, ih = i_h0 + rh, iw = i_w0 + rw;
    if (it >= k.Ti || ih >= k.Hi || iw >= k.Wi) continue;
    float g[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) g[e] = 0.f;
    // halo offset e (per dim) holds window o = o0 + local + e; this cell is its tap d = i - (o - pad) = k-1-e
    for (int et = 0; et < k.kt; ++et)
      for (int eh = 0; eh < k.kh; ++eh)
        for (int ew = 0; ew < k.kw; ++ew) {
          const int hp = ((rt + et) * p.Hh + rh + eh) * p.Wh + rw + ew;
          const int tap = ((k.kt - 1 - et) * k.kh + (k.kh - 1 - eh)) * k.kw + (k.kw - 1 - ew);
          int id[EPL];
          PV<T>::ldidx((const uint8_t*)sidx + ((size_t)ch * p.P + hp) * EPL, id);
          bool any = false;
#pragma unroll
          for (int e = 0; e < EPL; ++e) any |= id[e] == tap;
          if (!any) continue;
          float go[EPL];
          PV<T>::ld(smem + pplane_off(ch, p.plane_b) + hp * 16, go);
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            if (id[e] == tap) g[e] += go[e];
        }
    const size_t ipos = (((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw);
    if (k.mask) {
      float mk[EPL];
      PV<T>::ld(k.mask + (ipos * k.mask_ld + k.mask_coff + c0) * sizeof(T), mk);
#pragma unroll
      for (int e = 0; e < EPL; ++e) g[e] = mk[e] > 0.f ? g[e] : 0.f;
    }
    PV<T>::st(k.gin + (ipos * k.gin_ld + k.gin_coff + c0) * sizeof(T), g);
  }
}

// Scatter form of the backward (any stride): every OUTPUT window adds its gradient to the cell its argmax points at --
// one LDS atomic per element instead of kt*kh*kw index compares per input cell (the gather form above is VALU-bound).
// A workgroup owns a tile of INPUT cells (accumulators in LDS) and walks the windows that can reach it, reading
// idx / gout straight from global memory.
template <typename T>
__global__ __launch_bounds__(256, 4) void maxpool_scatter_bwd(const PoolTP p, unsigned m_khkw, unsigned m_kw) {
  constexpr int EPL = PV<T>::EPL, SLABC = 4 * EPL, UNR = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // accumulators [e][chunk][cell], plane stride RS = 16 (mod 64) floats: the 64 lanes of one ds_add_f32 (fixed e; 4
  // chunks x 16 neighbouring cells) fall on 64 different banks instead of the 4-8 a [cell][e] layout would give
  float* const acc = (float*)smem;
  const int RS = p.plane_b;
  const PoolKP& k = p.k;
  const int tid = threadIdx.x, ch = tid & 3;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int c0 = cslab * SLABC + ch * EPL;
  const bool chvalid = c0 < k.C;
  const int i_t0 = tt * p.Tt, i_h0 = th * p.Ht, i_w0 = tw * p.Wt;
  // windows that touch the tile: o in [ceil((i0 + pad - k + 1) / s), floor((i0 + tile - 1 + pad) / s)], clipped to the grid
  const int o_t0 = max(0, (i_t0 + k.pt - k.kt + k.st) / k.st), o_t1 = min(k.To - 1, (i_t0 + p.Tt - 1 + k.pt) / k.st);
  const int o_h0 = max(0, (i_h0 + k.ph - k.kh + k.sh) / k.sh), o_h1 = min(k.Ho - 1, (i_h0 + p.Ht - 1 + k.ph) / k.sh);
  const int o_w0 = max(0, (i_w0 + k.pw - k.kw + k.sw) / k.sw), o_w1 = min(k.Wo - 1, (i_w0 + p.Wt - 1 + k.pw) / k.sw);
  const int nt = max(0, o_t1 - o_t0 + 1), nh = max(0, o_h1 - o_h0 + 1), nw = max(0, o_w1 - o_w0 + 1);
  const int nhw = nh * nw, P = nt * nhw;
  const float inv_hw = 1.0f / (float)max(nhw, 1), inv_w = 1.0f / (float)max(nw, 1);
  for (int i = tid; i < 4 * RS * EPL; i += 256) acc[i] = 0.f;
  // bf16 mode accumulates in 32-bit FIXED POINT with a per-workgroup power-of-two scale: ds_add_f32 (and
  // ds_cmpst_rtn_b32) run ~10x slower than the integer LDS atomics on gfx950 (measured: 0.28 ms vs 0.12 ms for the
  // Mixed_3c pool), and integer sums are associative, so the result does not depend on the order the waves arrive in.
  // Scale = 2^24 / 2^floor(log2 max|gout|) over the windows this workgroup reads: |v * scale| < 2^25, <= 27 addends
  // < 2^30; a bf16 gradient (8 significant bits) within 2^-17 of the largest one is represented exactly and the
  // absolute error of a cell is < 27 * 2^-25 * max|gout|, far below the bf16 rounding of the stored result.
  // fp32 mode (the parity mode) keeps exact float atomics.
  constexpr bool FIXED = sizeof(T) == 2;
  float scale = 1.f, inv_scale = 1.f;
  if (FIXED) {
    unsigned* const smax = (unsigned*)(acc + 4 * RS * EPL);
    if (tid == 0) *smax = 0u;
    __syncthreads();
    unsigned mx = 0u;
    if (chvalid)
      for (int hp = tid >> 2; hp < P; hp += 64) {
        const int a = (int)(((float)hp + 0.5f) * inv_hw), rem = hp - a * nhw;
        const int bq = (int)(((float)rem + 0.5f) * inv_w), c = rem - bq * nw;
        const size_t opos = (((size_t)(b * k.To + o_t0 + a) * k.Ho + o_h0 + bq) * k.Wo + o_w0 + c);
        const uint4 g = *(const uint4*)(k.gout + (opos * k.gout_ld + k.gout_coff + c0) * sizeof(T));
        // largest |bf16| of the 8 packed values, as an fp32 bit pattern (magnitudes order like unsigned integers)
        const unsigned w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) mx = max(mx, max((w[i] << 16) & 0x7fffffffu, w[i] & 0x7fff0000u));
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
    if ((tid & 63) == 0) atomicMax(smax, mx);
    __syncthreads();
    int ex = (int)(*smax >> 23);                               // biased exponent of the largest magnitude
    ex = min(max(ex, 26), 254);                                // (inf/nan gradients are not representable; clamp)
    scale = __uint_as_float((unsigned)(127 + 24 + 127 - ex) << 23);
    inv_scale = __uint_as_float((unsigned)(127 - 24 - 127 + ex) << 23);
  }
  __syncthreads();
  const int khkw = k.kh * k.kw;
  float* const myacc = acc + ch * RS;
  if (chvalid)
    for (int base = tid >> 2; base < P; base += 64 * UNR) {
      int id[UNR][EPL];
      float go[UNR][EPL];
      int lt0[UNR], lh0[UNR], lw0[UNR];
      // all loads of the UNR positions are issued before the first atomic: one memory round trip per UNR positions
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int hp = min(base + u * 64, P - 1);
        const int a = (int)(((float)hp + 0.5f) * inv_hw), rem = hp - a * nhw;
        const int bq = (int)(((float)rem + 0.5f) * inv_w), c = rem - bq * nw;
        const int ot = o_t0 + a, oh = o_h0 + bq, ow = o_w0 + c;
        const size_t opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
        PV<T>::ldidx(k.idx + opos * k.C + c0, id[u]);
        PV<T>::ld(k.gout + (opos * k.gout_ld + k.gout_coff + c0) * sizeof(T), go[u]);
        // window o covers cells o*s - pad + d; local cell = that - tile origin
        lt0[u] = ot * k.st - k.pt - i_t0; lh0[u] = oh * k.sh - k.ph - i_h0; lw0[u] = ow * k.sw - k.pw - i_w0;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (base + u * 64 >= P) break;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int tap = id[u][e];                                    // 255 ("no cell") decodes out of range below
          const int dt = (int)(((unsigned)tap * m_khkw) >> 20), r2 = tap - dt * khkw;
          const int dh = (int)(((unsigned)r2 * m_kw) >> 20), dw = r2 - dh * k.kw;
          const int lt = lt0[u] + dt, lh = lh0[u] + dh, lw = lw0[u] + dw;
          if ((unsigned)lt < (unsigned)p.Tt && (unsigned)lh < (unsigned)p.Ht && (unsigned)lw < (unsigned)p.Wt && dt < k.kt) {
            float* const cell = &myacc[e * 4 * RS + (lt * p.Ht + lh) * p.Wt + lw];
            if (FIXED) atomicAdd((unsigned*)cell, (unsigned)__float2int_rn(go[u][e] * scale));
            else atomicAdd(cell, go[u][e]);
          }
        }
      }
    }
  __syncthreads();
  if (!chvalid) return;
  const int hw = p.Ht * p.Wt;
  for (int r = tid >> 2; r < p.rows; r += 64) {
    int rt, rh, rw; split3(r, hw, 1.0f / (float)hw, p.Wt, 1.0f / (float)p.Wt, rt, rh, rw);
    const int it = i_t0 + rt, ih = i_h0 + rh, iw = i_w0 + rw;
    if (it >= k.Ti || ih >= k.Hi || iw >= k.Wi) continue;
    float g[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) g[e] = FIXED ? (float)(int)__float_as_uint(myacc[e * 4 * RS + r]) * inv_scale : myacc[e * 4 * RS + r];
    const size_t ipos = (((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw);
    if (k.mask) {
      float mk[EPL];
      PV<T>::ld(k.mask + (ipos * k.mask_ld + k.mask_coff + c0) * sizeof(T), mk);
#pragma unroll
      for (int e = 0; e < EPL; ++e) g[e] = mk[e] > 0.f ? g[e] : 0.f;
    }
    PV<T>::st(k.gin + (ipos * k.gin_ld + k.gin_coff + c0) * sizeof(T), g);
  }
}

// ------------------------------------------------------------------------------------------------
// Branch_3 backward of an Inception block in ONE kernel (i3d.py:211-216 backward; bf16): the data-gradient of the 1x1x1 unit
// (a GEMM with K = 32..128) computed on MFMA inside the pool's scatter backward, instead of  1x1x1 data-gradient -> HBM ->
// scatter  (two dependent launches, 2 x cur_c bytes per position through HBM, and the critical chain of every block's
// backward phase: the two 3x3x3 data-gradients it runs beside are single launches).
//   gin[cell, c] = sum_{windows w whose argmax for channel c is cell} ( sum_k g[w, k] * Wt[k, c] )
// Workgroup = (tile of INPUT cells, slab of 32 channels), as maxpool_scatter_bwd.  A wave takes 16 window positions per pass:
// their g rows are the MFMA B fragments (lane (q, m) loads g[pos m][32 ks + 8 q ..], 16 bytes, straight from global: K-contiguous),
// the slab's weights the A fragments (packed by flk_pool_gemm_weights_create, kept in registers for the whole workgroup), so a lane
// ends up with the 8 values  (position m) x (channels 16 f + 4 q + j)  in fp32 -- never rounded to bf16, never stored -- and adds
// each to the cell its saved argmax names, in 32-bit fixed point with integer LDS atomics exactly like maxpool_scatter_bwd
// (order-independent: bitwise reproducible).  The per-workgroup scale needs max |value| first: the product pass runs twice
// (max, then scatter); its MFMAs are ~1 % of the kernel.
struct PoolGemmP {
  PoolTP t;
  const char* g; int g_ld, g_coff, KS;       // KS = K / 32 k-steps
  const char* wpack;                          // [slab][ks][f = 0,1][64 lanes][16 B]
  int dbg;                                    // timing experiments only (FLK_PG_DBG): 1 = no atomics, 2 = no tile write, 4 = no g loads, 8 = no index loads, 16 = no zeroing
};

template <int KS>
__global__ __launch_bounds__(256, 4) void maxpool_scatter_gemm_bwd(const PoolGemmP pg, unsigned m_khkw, unsigned m_kw) {
  typedef __bf16 frag8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* const acc = (unsigned*)smem;     // [32 channels][RS cells] fixed-point accumulators
  const PoolTP& p = pg.t;
  const PoolKP& k = p.k;
  const int RS = p.plane_b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, m = lane & 15;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int i_t0 = tt * p.Tt, i_h0 = th * p.Ht, i_w0 = tw * p.Wt;
  const int o_t0 = max(0, (i_t0 + k.pt - k.kt + k.st) / k.st), o_t1 = min(k.To - 1, (i_t0 + p.Tt - 1 + k.pt) / k.st);
  const int o_h0 = max(0, (i_h0 + k.ph - k.kh + k.sh) / k.sh), o_h1 = min(k.Ho - 1, (i_h0 + p.Ht - 1 + k.ph) / k.sh);
  const int o_w0 = max(0, (i_w0 + k.pw - k.kw + k.sw) / k.sw), o_w1 = min(k.Wo - 1, (i_w0 + p.Wt - 1 + k.pw) / k.sw);
  const int nt = max(0, o_t1 - o_t0 + 1), nh = max(0, o_h1 - o_h0 + 1), nw = max(0, o_w1 - o_w0 + 1);
  const int nhw = nh * nw, P = nt * nhw;
  const float inv_hw = 1.0f / (float)max(nhw, 1), inv_w = 1.0f / (float)max(nw, 1);
  for (int i = tid; i < 32 * RS; i += 256) acc[i] = 0u;
  unsigned* const smax = acc + 32 * RS;
  if (tid == 0) *smax = 0u;

  // A fragments of this slab: channels [32 cslab, +32), all K
  frag8 af[KS][2];
  {
    const char* wp = pg.wpack + ((size_t)cslab * KS * 2 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int f = 0; f < 2; ++f) af[ks][f] = *(const frag8*)(wp + (size_t)(ks * 2 + f) * 1024);
  }
  // window position hp -> linear output position and its origin relative to the tile
  auto decode = [&](int hp, size_t& opos, int& lt0, int& lh0, int& lw0) {
    const int a = (int)(((float)hp + 0.5f) * inv_hw), rem = hp - a * nhw;
    const int bq = (int)(((float)rem + 0.5f) * inv_w), c = rem - bq * nw;
    const int ot = o_t0 + a, oh = o_h0 + bq, ow = o_w0 + c;
    opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
    lt0 = ot * k.st - k.pt - i_t0; lh0 = oh * k.sh - k.ph - i_h0; lw0 = ow * k.sw - k.pw - i_w0;
  };
  auto product = [&](size_t opos, f32x4 (&v)[2]) {
    const char* gp = pg.g + (opos * pg.g_ld + pg.g_coff + q * 8) * 2;
    frag8 bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bf[ks] = *(const frag8*)(gp + ks * 64);
    v[0] = f32x4{0.f, 0.f, 0.f, 0.f}; v[1] = v[0];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      v[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][0], bf[ks], v[0], 0, 0, 0);
      v[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][1], bf[ks], v[1], 0, 0, 0);
    }
  };
  __syncthreads();
  // ---- pass 1: the largest |value| this workgroup will add (as an fp32 bit pattern: magnitudes order like unsigned integers) ----
  unsigned mx = 0u;
  for (int base = wave * 16; base < P; base += 64) {
    const int hp = min(base + m, P - 1);
    size_t opos; int lt0, lh0, lw0;
    decode(hp, opos, lt0, lh0, lw0);
    f32x4 v[2];
    product(opos, v);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = max(mx, __float_as_uint(v[f][j]) & 0x7fffffffu);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
  if (lane == 0) atomicMax(smax, mx);
  __syncthreads();
  int ex = (int)(*smax >> 23);
  ex = min(max(ex, 26), 254);
  const float scale = __uint_as_float((unsigned)(127 + 24 + 127 - ex) << 23);
  const float inv_scale = __uint_as_float((unsigned)(127 - 24 - 127 + ex) << 23);
  // ---- pass 2: the same products, scattered ----
  const int khkw = k.kh * k.kw;
  for (int base = wave * 16; base < P; base += 64) {
    const int hp = min(base + m, P - 1);
    const bool live = base + m < P;
    size_t opos; int lt0, lh0, lw0;
    decode(hp, opos, lt0, lh0, lw0);
    unsigned ib[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) ib[f] = *(const unsigned*)(k.idx + opos * k.C + cslab * 32 + f * 16 + q * 4);
    f32x4 v[2];
    product(opos, v);
    if (!live) continue;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tap = (int)((ib[f] >> (8 * j)) & 255u);              // 255 ("no cell") decodes out of range below
        const int dt = (int)(((unsigned)tap * m_khkw) >> 20), r2 = tap - dt * khkw;
        const int dh = (int)(((unsigned)r2 * m_kw) >> 20), dw = r2 - dh * k.kw;
        const int lt = lt0 + dt, lh = lh0 + dh, lw = lw0 + dw;
        if ((unsigned)lt < (unsigned)p.Tt && (unsigned)lh < (unsigned)p.Ht && (unsigned)lw < (unsigned)p.Wt && dt < k.kt)
          atomicAdd(&acc[(f * 16 + q * 4 + j) * RS + (lt * p.Ht + lh) * p.Wt + lw], (unsigned)__float2int_rn(v[f][j] * scale));
      }
  }
  __syncthreads();
  // ---- write the tile: thread = (cell, 16-byte channel chunk) ----
  const int ch = tid & 3, c0 = cslab * 32 + ch * 8;
  if (c0 >= k.C) return;
  const int hw = p.Ht * p.Wt;
  for (int r = tid >> 2; r < p.rows; r += 64) {
    int rt, rh, rw; split3(r, hw, 1.0f / (float)hw, p.Wt, 1.0f / (float)p.Wt, rt, rh, rw);
    const int it = i_t0 + rt, ih = i_h0 + rh, iw = i_w0 + rw;
    if (it >= k.Ti || ih >= k.Hi || iw >= k.Wi) continue;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = (float)(int)acc[(ch * 8 + e) * RS + r] * inv_scale;
    const size_t ipos = (((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw);
    PV<bf16_t>::st(k.gin + (ipos * k.gin_ld + k.gin_coff + c0) * 2, g);
  }
}

// The same kernel for tiles reached by at most 64 * NIT windows (the (4,7,7) tiles of the I3D blocks: 486): a wave's NIT passes are
// unrolled, the g rows and index bytes of NB passes are requested together (the loop form above waits for one load round trip per pass and
// per product pass -- 2 x 8 dependent round trips per workgroup: latency-bound, 1 TB/s), and the products and index bytes stay in registers
// between the max pass and the scatter pass: each is loaded and computed ONCE.  That alone changed little: the kernel is VALU-bound on the
// scatter's per-element decode (tap -> (dt,dh,dw), three bound checks, address: ~27 instructions x 8 elements x 8 passes per lane, one wave
// per SIMD and workgroup).  3x3x3 windows only: which taps of a window land inside the tile is a 27-bit mask computed once per window
// (8 elements share it), and a tap's cell offset comes from a 256-entry table in LDS: ~12 instructions per element.
// Measured (Mixed_3c, 8 x 32 x 28 x 28 x 256 channels, K = 64; tools/pool_gemm_time.py, FLK_PG_DBG ablations): loop form 192 us, this form
// 161 us; without atomics 137, without the tile write 138, without g loads 145, without index loads 138, without all of them and the
// zeroing 80 us -- what is left is the decode of 486 windows x 32 channels per 196-cell tile (2.48x the cells: the windows that reach a
// tile overlap its neighbours'), ~45 us of VALU time at ~12 instructions per element, plus the per-workgroup chain.  Tried and dropped:
// accumulator plane stride 4 / 20 (mod 64) instead of 16 (no change: the atomics are not bank-bound); the lanes of one atomic on windows 3
// apart, which cannot share an argmax cell (5 % slower: the loads lose locality, same-address serialisation is not the limit either); a
// persistent workgroup per tile that walks over its channel slabs with the decode, masks and g fragments kept and the next slab's
// weights / index bytes prefetched (169 us: 246 VGPRs, two workgroups per CU).
template <int KS, int NIT, int NB>
__global__ __launch_bounds__(256, KS >= 3 ? 2 : 3) void maxpool_scatter_gemm_bwd_reg(const PoolGemmP pg, unsigned m_khkw, unsigned m_kw) {
  typedef __bf16 frag8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* const acc = (unsigned*)smem;     // [32 channels][RS cells] fixed-point accumulators
  const PoolTP& p = pg.t;
  const PoolKP& k = p.k;
  const int RS = p.plane_b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, m = lane & 15;
  int bid, cslab;
  if (!tile_slab_of_block(p, bid, cslab)) return;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int i_t0 = tt * p.Tt, i_h0 = th * p.Ht, i_w0 = tw * p.Wt;
  const int o_t0 = max(0, (i_t0 + k.pt - k.kt + k.st) / k.st), o_t1 = min(k.To - 1, (i_t0 + p.Tt - 1 + k.pt) / k.st);
  const int o_h0 = max(0, (i_h0 + k.ph - k.kh + k.sh) / k.sh), o_h1 = min(k.Ho - 1, (i_h0 + p.Ht - 1 + k.ph) / k.sh);
  const int o_w0 = max(0, (i_w0 + k.pw - k.kw + k.sw) / k.sw), o_w1 = min(k.Wo - 1, (i_w0 + p.Wt - 1 + k.pw) / k.sw);
  const int nt = max(0, o_t1 - o_t0 + 1), nh = max(0, o_h1 - o_h0 + 1), nw = max(0, o_w1 - o_w0 + 1);
  const int nhw = nh * nw, P = nt * nhw;                       // <= 64 * NIT (the host checks the tile)
  const float inv_hw = 1.0f / (float)max(nhw, 1), inv_w = 1.0f / (float)max(nw, 1);
  if (!PG_DBG(16)) for (int i = tid; i < 32 * RS; i += 256) acc[i] = 0u;
  unsigned* const smax = acc + 32 * RS;
  if (tid == 0) *smax = 0u;
  unsigned* const lut = smax + 64;                             // byte offset of tap's cell relative to the window origin's cell
  {
    const int dt = tid / 9, r2 = tid - dt * 9, dh = r2 / 3, dw = r2 - dh * 3;
    lut[tid] = tid < 27 ? (unsigned)(((dt * p.Ht + dh) * p.Wt + dw) * 4) : 0u;
  }
  f32x4 v[NIT][2];
  unsigned ib[NIT][2];
  int org[NIT];                                                // byte offset of the window origin's cell in an accumulator plane (may be negative)
  unsigned okm[NIT];                                           // bit tap = the tap's cell lies inside the tile
  unsigned mx = 0u;
  if (P > 0) {
    frag8 af[KS][2];
    {
      const char* wp = pg.wpack + ((size_t)cslab * KS * 2 * 64 + lane) * 16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) af[ks][f] = *(const frag8*)(wp + (size_t)(ks * 2 + f) * 1024);
    }
#pragma unroll
    for (int i0 = 0; i0 < NIT; i0 += NB) {
      frag8 bf[NB][KS];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + u;
        const int hp = wave * 16 + 64 * i + m;
        const bool live = hp < P;
        const int hc = min(hp, P - 1);
        const int a = (int)(((float)hc + 0.5f) * inv_hw), rem = hc - a * nhw;
        const int bq = (int)(((float)rem + 0.5f) * inv_w), c = rem - bq * nw;
        const int ot = o_t0 + a, oh = o_h0 + bq, ow = o_w0 + c;
        const size_t opos = (((size_t)(b * k.To + ot) * k.Ho + oh) * k.Wo + ow);
        const int lt0 = ot * k.st - k.pt - i_t0, lh0 = oh * k.sh - k.ph - i_h0, lw0 = ow * k.sw - k.pw - i_w0;
        org[i] = ((lt0 * p.Ht + lh0) * p.Wt + lw0) * 4;
        // taps d = 0..2 of a dimension with 0 <= l0 + d < n, as 3 bits; the 27-bit mask is their outer product (bit (dt*3 + dh)*3 + dw)
        auto bits3 = [](int l0, int n) { return (7u << min(max(-l0, 0), 3)) & 7u & ((1u << min(max(n - l0, 0), 3)) - 1u); };
        const unsigned bt3 = bits3(lt0, p.Tt), bh3 = bits3(lh0, p.Ht), bw3 = bits3(lw0, p.Wt);
        const unsigned eh = (bh3 & 1u) | ((bh3 & 2u) << 2) | ((bh3 & 4u) << 4);          // spread by 3
        const unsigned et = (bt3 & 1u) | ((bt3 & 2u) << 8) | ((bt3 & 4u) << 16);         // spread by 9
        okm[i] = live ? bw3 * eh * et : 0u;
#pragma unroll
        for (int f = 0; f < 2; ++f) ib[i][f] = PG_DBG(8) ? (unsigned)(m * 0x01010101u) & 0x0f0f0f0fu : *(const unsigned*)(k.idx + opos * k.C + cslab * 32 + f * 16 + q * 4);
        const char* gp = pg.g + (opos * pg.g_ld + pg.g_coff + q * 8) * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bf[u][ks] = PG_DBG(4) ? af[ks][0] : *(const frag8*)(gp + ks * 64);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + u;
        v[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i][1] = v[i][0];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          v[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][0], bf[u][ks], v[i][0], 0, 0, 0);
          v[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][1], bf[u][ks], v[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) mx = max(mx, __float_as_uint(v[i][f][j]) & 0x7fffffffu);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
  __syncthreads();                                             // the accumulators and *smax are zero
  if (lane == 0) atomicMax(smax, mx);
  __syncthreads();
  int ex = (int)(*smax >> 23);
  ex = min(max(ex, 26), 254);
  const float scale = __uint_as_float((unsigned)(127 + 24 + 127 - ex) << 23);
  const float inv_scale = __uint_as_float((unsigned)(127 - 24 - 127 + ex) << 23);
  if (P > 0) {
    // Branch-free: the 8 table reads of a window are requested together, and an element whose cell is outside the tile adds into a
    // dummy word of its own thread instead of being skipped.  (Written as  if (inside) atomicAdd(.. + lut[tap] ..)  hipcc emitted, per
    // element, exec-mask branch -> ds_read_b32 -> s_waitcnt lgkmcnt(0) -> ds_add_u32: 64 exposed LDS round trips per lane.)
    unsigned* const dummy = lut + 256 + tid;
    if (!PG_DBG(1)) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        char* const plane0 = (char*)acc + (size_t)(q * 4 * RS) * 4 + org[i];
        unsigned off[2][4];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) off[f][j] = lut[(ib[i][f] >> (8 * j)) & 255u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned tap = (ib[i][f] >> (8 * j)) & 255u;             // 255 ("no cell"): bit 31 of the mask, never set
            const bool inside = (okm[i] >> (tap & 31u)) & 1u;
            unsigned* const cell = (unsigned*)(plane0 + (size_t)((f * 16 + j) * RS) * 4 + off[f][j]);
            atomicAdd(inside ? cell : dummy, (unsigned)__float2int_rn(v[i][f][j] * scale));
          }
      }
    }
  }
  __syncthreads();
  // ---- write the tile: thread = (cell, 16-byte channel chunk) ----
  const int ch = tid & 3, c0 = cslab * 32 + ch * 8;
  if (c0 >= k.C || PG_DBG(2)) return;
  const int hw = p.Ht * p.Wt;
  for (int r = tid >> 2; r < p.rows; r += 64) {
    int rt, rh, rw; split3(r, hw, 1.0f / (float)hw, p.Wt, 1.0f / (float)p.Wt, rt, rh, rw);
    const int it = i_t0 + rt, ih = i_h0 + rh, iw = i_w0 + rw;
    if (it >= k.Ti || ih >= k.Hi || iw >= k.Wi) continue;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = (float)(int)acc[(ch * 8 + e) * RS + r] * inv_scale;
    const size_t ipos = (((size_t)(b * k.Ti + it) * k.Hi + ih) * k.Wi + iw);
    PV<bf16_t>::st(k.gin + (ipos * k.gin_ld + k.gin_coff + c0) * 2, g);
  }
}

// input-cell tile for the scatter backward: minimise bytes moved per useful cell (tile writes + the windows read,
// which overlap between neighbouring tiles) plus a fixed per-workgroup cost
// max_reach > 0: only tiles reached by at most that many windows (the register-resident fused Branch_3 backward holds 64 x 8 of them; the
// reference's 90-frame clips -- T/2 = 45 frames in Mixed_3* -- otherwise get 5-frame tiles reached by 567 windows and the slower loop form:
// Mixed_3c 0.249 ms against 0.19 for 1.41 x the positions of the 64-frame clips)
static flk_tile choose_scatter_tile(const flk_pool_args* a, long max_reach = 0) {
  flk_tile best{1, 1, 1};
  double best_cost = 1e300;
  for (int Tt = 1; Tt <= a->Ti && Tt <= 8; ++Tt)
    for (int Ht = 1; Ht <= a->Hi && Tt * Ht <= 256; ++Ht) {
      const int wmax = 256 / (Tt * Ht) < a->Wi ? 256 / (Tt * Ht) : a->Wi;
      for (int Wt = 1; Wt <= wmax; ++Wt) {
        const double tiles = (double)((a->Ti + Tt - 1) / Tt) * ((a->Hi + Ht - 1) / Ht) * ((a->Wi + Wt - 1) / Wt);
        const double outs = (double)((Tt + a->kt - 2) / a->st + 1) * ((Ht + a->kh - 2) / a->sh + 1) * ((Wt + a->kw - 2) / a->sw + 1);
        if (max_reach > 0 && outs > (double)max_reach) continue;
        const double passes = (double)(((long)outs + 255) / 256);           // 64 position threads x 4-deep unroll
        const double cost = tiles * (Tt * Ht * Wt * 16.0 + outs * 24.0 + passes * 2048.0 + 2048.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = flk_tile{Tt, Ht, Wt}; }
      }
    }
  return best;
}

template <typename T>
static int launch_scatter_bwd(const PoolKP& kp, const flk_pool_args* a, hipStream_t s) {
  constexpr int EPL = PV<T>::EPL;
  PoolTP tp{};
  tp.k = kp;
  const flk_tile t = choose_scatter_tile(a);
  tp.Tt = t.Tt; tp.Ht = t.Ht; tp.Wt = t.Wt; tp.rows = t.Tt * t.Ht * t.Wt;
  tp.nTt = (a->Ti + t.Tt - 1) / t.Tt; tp.nTh = (a->Hi + t.Ht - 1) / t.Ht; tp.nTw = (a->Wi + t.Wt - 1) / t.Wt;
  tp.ntiles = a->B * tp.nTt * tp.nTh * tp.nTw; tp.nslab = (a->C + 4 * EPL - 1) / (4 * EPL);
  const dim3 grid((unsigned)((tp.ntiles + 7) / 8 * 8 * tp.nslab));
  auto magic = [](int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); };
  tp.plane_b = (tp.rows + 47) / 64 * 64 + 16;                         // accumulator plane stride in floats, = 16 (mod 64)
  const size_t lds = (size_t)(4 * tp.plane_b * EPL + 256) * sizeof(float);    // <= 35 KiB (+ one dummy word per thread)
  FLK_LAUNCH_KERNEL(maxpool_scatter_bwd<T>, grid, dim3(256), lds, s, tp, magic(a->kh * a->kw), magic(a->kw));
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// stride-1 SAME pooling with an odd window and enough reuse to pay for the LDS staging
static bool use_tiled(const flk_pool_args* a) {
  return a->st == 1 && a->sh == 1 && a->sw == 1 && a->kt * a->kh * a->kw >= 8 && a->To == a->Ti && a->Ho == a->Hi &&
         a->Wo == a->Wi && a->pt < a->kt && a->ph < a->kh && a->pw < a->kw;
}

template <typename T>
static int launch_tiled(const PoolKP& kp, const flk_pool_args* a, bool bwd, hipStream_t s) {
  PoolTP tp{};
  tp.k = kp;
  const flk_tile t = flk_choose_tile(a->To, a->Ho, a->Wo, a->kt, a->kh, a->kw, 1, 1, 1, 256, 768);   // 2 workgroups per CU
  tp.Tt = t.Tt; tp.Ht = t.Ht; tp.Wt = t.Wt; tp.rows = t.Tt * t.Ht * t.Wt;
  tp.nTt = (a->To + t.Tt - 1) / t.Tt; tp.nTh = (a->Ho + t.Ht - 1) / t.Ht; tp.nTw = (a->Wo + t.Wt - 1) / t.Wt;
  tp.Th = t.Tt + a->kt - 1; tp.Hh = t.Ht + a->kh - 1; tp.Wh = t.Wt + a->kw - 1;
  tp.P = tp.Th * tp.Hh * tp.Wh;
  FLK_REQUIRE(tp.P <= FLK_MAX_HALO, "maxpool tiled: halo %d too large", tp.P);
  tp.plane_b = (tp.P * 16 + 255) / 256 * 256;
  constexpr int EPL = PV<T>::EPL;
  const size_t lds = 4 * (size_t)tp.plane_b + 64 + (bwd ? (size_t)4 * tp.P * EPL + 16 : 0);
  tp.ntiles = a->B * tp.nTt * tp.nTh * tp.nTw; tp.nslab = (a->C + 4 * EPL - 1) / (4 * EPL);
  const dim3 grid((unsigned)((tp.ntiles + 7) / 8 * 8 * tp.nslab));
  static bool attr_fwd[FLK_MAX_DEVICES] = {}, attr_bwd[FLK_MAX_DEVICES] = {};
  if (int rc = flk_raise_lds_limit((const void*)maxpool_s1_tiled_fwd<T>, 96 * 1024, attr_fwd)) return rc;
  if (int rc = flk_raise_lds_limit((const void*)maxpool_s1_tiled_bwd<T>, 96 * 1024, attr_bwd)) return rc;
  if (bwd) FLK_LAUNCH_KERNEL(maxpool_s1_tiled_bwd<T>, grid, dim3(256), lds, s, tp);
  else if (sizeof(T) == 2 && a->kt == 3 && a->kh == 3 && a->kw == 3 && a->pt == 1 && a->ph == 1 && a->pw == 1) {
    return a->Wo % 7 == 0 ? launch_wrun_fwd<7>(kp, a, s) : launch_wrun_fwd<8>(kp, a, s);
  } else if (sizeof(T) == 2 && a->kt == 3 && a->kh == 3 && a->kw == 3) {
    static bool attr2[FLK_MAX_DEVICES] = {};
    if (int rc = flk_raise_lds_limit((const void*)maxpool_s1_tiled_fwd_bf16<3, 3, 3>, 96 * 1024, attr2)) return rc;
    FLK_LAUNCH_KERNEL((maxpool_s1_tiled_fwd_bf16<3, 3, 3>), grid, dim3(256), lds, s, tp);
  } else FLK_LAUNCH_KERNEL(maxpool_s1_tiled_fwd<T>, grid, dim3(256), lds, s, tp);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
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
  kp.relu_input = a->relu_input;
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
  FLK_REQUIRE(dtype == FLK_BF16 || dtype == FLK_F32, "flk_maxpool3d_fwd: bad dtype");
  if (use_tiled(a)) return dtype == FLK_BF16 ? launch_tiled<bf16_t>(kp, a, false, (hipStream_t)stream) : launch_tiled<float>(kp, a, false, (hipStream_t)stream);
  const int epl = dtype == FLK_BF16 ? 8 : 4;
  FLK_REQUIRE(a->Ho < 65536 && (long)a->B * a->To < 65536, "flk_maxpool3d_fwd: grid too large");
  const dim3 grid((unsigned)((a->Wo * (a->C / epl) + 255) / 256), (unsigned)a->Ho, (unsigned)(a->B * a->To));
  const int kcode = a->kt * 100 + a->kh * 10 + a->kw;
  if (dtype == FLK_BF16) {
    if (kcode == 133) FLK_LAUNCH_KERNEL((maxpool_fwd_kernel_k<bf16_t, 1, 3, 3>), grid, dim3(256), 0, (hipStream_t)stream, kp);
    else if (kcode == 333) FLK_LAUNCH_KERNEL((maxpool_fwd_kernel_k<bf16_t, 3, 3, 3>), grid, dim3(256), 0, (hipStream_t)stream, kp);
    else if (kcode == 222) FLK_LAUNCH_KERNEL((maxpool_fwd_kernel_k<bf16_t, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, kp);
    else FLK_LAUNCH_KERNEL(maxpool_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, kp);
  } else {
    if (kcode == 133) FLK_LAUNCH_KERNEL((maxpool_fwd_kernel_k<float, 1, 3, 3>), grid, dim3(256), 0, (hipStream_t)stream, kp);
    else if (kcode == 222) FLK_LAUNCH_KERNEL((maxpool_fwd_kernel_k<float, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, kp);
    else FLK_LAUNCH_KERNEL(maxpool_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, kp);
  }
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
  FLK_REQUIRE(dtype == FLK_BF16 || dtype == FLK_F32, "flk_maxpool3d_bwd: bad dtype");
  {
    hipStream_t s = (hipStream_t)stream;
#define FLK_OWNER(KT, KH, KW, ST, SH, SW)                                                                   \
    if (strided_owner_ok(a, KT, KH, KW, ST, SH, SW))                                                          \
      return dtype == FLK_BF16 ? launch_strided_bwd<bf16_t, KT, KH, KW, ST, SH, SW>(kp, a, s)                 \
                               : launch_strided_bwd<float, KT, KH, KW, ST, SH, SW>(kp, a, s)
    FLK_OWNER(1, 3, 3, 1, 2, 2); FLK_OWNER(3, 3, 3, 2, 2, 2); FLK_OWNER(2, 2, 2, 2, 2, 2);
#undef FLK_OWNER
  }
  // fp32 is the parity mode: it takes the gather form (fixed summation order) so that two fp32 runs are bitwise equal; the scatter
  // form's float LDS atomics are order-dependent in the last ulp, which Adam turns into 1e-4 relative on tiny components.  bf16 takes the
  // scatter form (32-bit fixed-point integer LDS atomics: order-independent, bitwise reproducible as well)
  if (dtype == FLK_BF16) return launch_scatter_bwd<bf16_t>(kp, a, (hipStream_t)stream);
  if (use_tiled(a)) return launch_tiled<float>(kp, a, true, (hipStream_t)stream);
  const int epl = dtype == FLK_BF16 ? 8 : 4;
  FLK_REQUIRE(a->Hi < 65536 && (long)a->B * a->Ti < 65536, "flk_maxpool3d_bwd: grid too large");
  const dim3 grid((unsigned)((a->Wi * (a->C / epl) + 255) / 256), (unsigned)a->Hi, (unsigned)(a->B * a->Ti));
  FLK_LAUNCH_KERNEL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, kp);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}


// ---- fused Branch_3 backward: weights + entry point -----------------------------------------------------------------------------
// wt_kc: fp32 [K][C] = the 1x1x1 unit's weight transposed, batch-norm scale folded in (Wt[k][c] = w[c][k] * scale[k]); K % 32 == 0,
// C % 8 == 0.  Packed as MFMA A fragments per (32-channel slab, 32-wide k-step, half): lane (q, m) holds Wt[32 ks + 8 q + j][32 s + 16 f + m].
extern "C" int flk_pool_gemm_weights_create(const float* wt_kc, int K, int C, void** out_dev) {
  FLK_REQUIRE(wt_kc && out_dev && K > 0 && K % 32 == 0 && K <= 128 && C > 0 && C % 8 == 0, "flk_pool_gemm_weights_create: K must be 32, 64, 96 or 128 and C a multiple of 8 "
              "(got K %d, C %d)", K, C);
  const int nslab = (C + 31) / 32, KS = K / 32;
  std::vector<uint16_t> h((size_t)nslab * KS * 2 * 64 * 8, 0);
  auto bf = [](float f) -> uint16_t {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  };
  for (int s = 0; s < nslab; ++s)
    for (int ks = 0; ks < KS; ++ks)
      for (int f = 0; f < 2; ++f)
        for (int lane = 0; lane < 64; ++lane) {
          const int q = lane >> 4, m = lane & 15, c = s * 32 + f * 16 + m;
          for (int j = 0; j < 8; ++j) {
            const int kk = ks * 32 + q * 8 + j;
            h[((((size_t)s * KS + ks) * 2 + f) * 64 + lane) * 8 + j] = c < C ? bf(wt_kc[(size_t)kk * C + c]) : 0;
          }
        }
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, h.size() * 2);
  if (e != hipSuccess) { flk_set_error("hipMalloc(%zu): %s", h.size() * 2, hipGetErrorString(e)); return FLK_ENOMEM; }
  e = hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(d); flk_set_error("hipMemcpy: %s", hipGetErrorString(e)); return FLK_EHIP; }
  *out_dev = d;
  return FLK_OK;
}
extern "C" int flk_pool_gemm_weights_destroy(void* dev) {
  if (dev) (void)hipFree(dev);
  return FLK_OK;
}

extern "C" int flk_maxpool3d_bwd_gemm(const flk_pool_args* a, const void* g, int g_ld, int g_coff, int K, const void* wpack,
                                      void* gin, int gin_ld, int gin_coff, int dtype, void* stream) {
  int rc = check_pool(a);
  if (rc) return rc;
  FLK_REQUIRE(g && wpack && gin, "flk_maxpool3d_bwd_gemm: null argument");
  FLK_REQUIRE(dtype == FLK_BF16, "flk_maxpool3d_bwd_gemm: bf16 only (fp32, the parity mode, keeps the two-launch gather path)");
  FLK_REQUIRE(K > 0 && K % 32 == 0 && K <= 128, "flk_maxpool3d_bwd_gemm: K must be 32, 64, 96 or 128 (got %d)", K);
  FLK_REQUIRE(g_ld % 8 == 0 && g_coff % 8 == 0 && g_coff + K <= g_ld && gin_ld % 8 == 0 && gin_coff % 8 == 0 && gin_coff + a->C <= gin_ld,
              "flk_maxpool3d_bwd_gemm: bad ld / coff");
  FLK_REQUIRE((size_t)a->B * a->To * a->Ho * a->Wo * (size_t)(g_ld > a->C ? g_ld : a->C) < (1ull << 31), "flk_maxpool3d_bwd_gemm: tensor too large");
  PoolGemmP pg{};
  fill(pg.t.k, a);
  pg.t.k.gin = (char*)gin; pg.t.k.gin_ld = gin_ld; pg.t.k.gin_coff = gin_coff;
  pg.g = (const char*)g; pg.g_ld = g_ld; pg.g_coff = g_coff; pg.KS = K / 32; pg.wpack = (const char*)wpack;
#ifdef FLK_ABLATE
  pg.dbg = flk_ablate_env("FLK_PG_DBG");
#endif
  const char* const reg_env = getenv("FLK_POOL_GEMM_REG");     // (read per call: the tests compare the two forms)
  const bool reg_form = !(reg_env && atoi(reg_env) == 0);
  const bool reg_able = reg_form && a->kt == 3 && a->kh == 3 && a->kw == 3;
  const flk_tile t = choose_scatter_tile(a, reg_able ? 512 : 0);
  PoolTP& tp = pg.t;
  tp.Tt = t.Tt; tp.Ht = t.Ht; tp.Wt = t.Wt; tp.rows = t.Tt * t.Ht * t.Wt;
  tp.nTt = (a->Ti + t.Tt - 1) / t.Tt; tp.nTh = (a->Hi + t.Ht - 1) / t.Ht; tp.nTw = (a->Wi + t.Wt - 1) / t.Wt;
  tp.ntiles = a->B * tp.nTt * tp.nTh * tp.nTw; tp.nslab = (a->C + 31) / 32;
  const dim3 grid((unsigned)((tp.ntiles + 7) / 8 * 8 * tp.nslab));
  auto magic = [](int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); };
  { // accumulator plane stride in words: 4 (mod 64) -- the four lane groups q of an atomic (planes 4 q + j) start 16 banks apart
    // (16 (mod 64), the stride of maxpool_scatter_bwd, puts all four on the same banks)
    constexpr int res = 4;
    tp.plane_b = (tp.rows - res + 63) / 64 * 64 + res;
  }
  const size_t lds = (size_t)(32 * tp.plane_b + 64) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const unsigned m1 = magic(a->kh * a->kw), m2 = magic(a->kw);
  // windows that can reach a tile: at most 64 * 8 -> the register-resident form
  const long reach = (long)((t.Tt + a->kt - 2) / a->st + 1) * ((t.Ht + a->kh - 2) / a->sh + 1) * ((t.Wt + a->kw - 2) / a->sw + 1);
  if (reg_able && reach <= 512) {
    const size_t lds_reg = lds + (size_t)(64 + 256 + 256) * sizeof(unsigned);      // + the tap table + one dummy word per thread
    switch (pg.KS) {
      case 1: FLK_LAUNCH_KERNEL((maxpool_scatter_gemm_bwd_reg<1, 8, 4>), grid, dim3(256), lds_reg, s, pg, m1, m2); break;
      case 2: FLK_LAUNCH_KERNEL((maxpool_scatter_gemm_bwd_reg<2, 8, 4>), grid, dim3(256), lds_reg, s, pg, m1, m2); break;
      case 3: FLK_LAUNCH_KERNEL((maxpool_scatter_gemm_bwd_reg<3, 8, 2>), grid, dim3(256), lds_reg, s, pg, m1, m2); break;
      default: FLK_LAUNCH_KERNEL((maxpool_scatter_gemm_bwd_reg<4, 8, 2>), grid, dim3(256), lds_reg, s, pg, m1, m2); break;
    }
    FLK_CHECK_HIP(hipGetLastError());
    return FLK_OK;
  }
  switch (pg.KS) {
    case 1: FLK_LAUNCH_KERNEL(maxpool_scatter_gemm_bwd<1>, grid, dim3(256), lds, s, pg, m1, m2); break;
    case 2: FLK_LAUNCH_KERNEL(maxpool_scatter_gemm_bwd<2>, grid, dim3(256), lds, s, pg, m1, m2); break;
    case 3: FLK_LAUNCH_KERNEL(maxpool_scatter_gemm_bwd<3>, grid, dim3(256), lds, s, pg, m1, m2); break;
    default: FLK_LAUNCH_KERNEL(maxpool_scatter_gemm_bwd<4>, grid, dim3(256), lds, s, pg, m1, m2); break;
  }
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

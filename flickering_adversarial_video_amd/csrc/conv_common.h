// Shared by the convolution kernels (conv_igemm.hip, conv_pc.hip): launch parameters, element-type traits, the LDS halo image's address
// arithmetic, the epilogue (scale / bias / add / ReLU / mask / 16-byte stores) and the LDS-DMA issue helper.  gfx950 only.
#pragma once
#include "flk_internal.h"

struct ConvKP {
  const char* in; const char* in2; const char* w; char* out; char* out2;
  const float* scale; const float* bias; const char* add; const char* mask;
  const float* pos_bias; long pos_bias_bstride;
  int in_ld, in_coff, cin;
  int B, Ti, Hi, Wi;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int To, Ho, Wo;
  int out_ld, out_coff, cout;
  int OT, OH, OW, ost, osh, osw, oot, ooh, oow;
  int add_ld, add_coff, mask_ld, mask_coff, relu;
  int Tt, Ht, Wt, nTt, nTh, nTw, rows;
  int Th, Hh, Wh, P, plane_b;
  int FP;            // frame pitch of the halo image in 16-byte slots (>= Hh * Wh; slot of halo cell (a, b, c) = a * FP + b * Wh + c)
  int tfast;         // tile rows enumerated w, then T, then h (1) instead of w, h, T (0): see pick_halo_layout
  int nslab, ntaps, cout_frags;
  int in2_ld, in2_coff, cin1, nslab1, out2_ld, out2_coff, cout1;
  unsigned m_HW, m_Wh, m_hw, m_Wt;   // ceil(2^20 / d): exact x / d for x * d < 2^20 (x < 1024 here)
  int ntile_n;
  int xcd_chunk;     // > 0: position tiles are dealt to the XCDs in contiguous chunks of this many (see the kernel's index decode)
  // deterministic split-K (blockIdx.y = slice of the input-channel slabs): raw fp32 partial sums [ksplit][positions][part_ld],
  // summed in slice order and finished by conv_splitk_finish_kernel
  float* part; int ksplit, part_ld; unsigned npos;
  int wn;            // grouped launches (conv_igemm_group_kernel): waves along N of THIS member (1 / 2 / 4), a run-time value there
};

template <typename T> struct Prec;
template <> struct Prec<bf16_t> {
  static constexpr int EPL = 8;
  typedef bf16x8 frag;
  __device__ static inline void mma(const frag& a, const frag& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  __device__ static inline void to_f32(const uint4& u, float* f) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static inline uint4 from_f32(const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
    return __builtin_bit_cast(uint4, v);
  }
};
template <> struct Prec<float> {
  static constexpr int EPL = 4;
  typedef f32x4 frag;
  __device__ static inline void mma(const frag& a, const frag& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
  }
  __device__ static inline void to_f32(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ static inline uint4 from_f32(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};

__device__ static inline int plane_off(int c, int plane_b) { return c * plane_b + (c >> 1) * 32; }
__device__ static inline int fdiv(int x, unsigned magic) { return (int)(((unsigned)x * magic) >> 20); }

// tile row r -> cell (rt, rh, rw) of the Tt x Ht x Wt tile.  m_hw / m_Wt are the magic numbers of the enumeration in use
// (tfast: rows run along w, then T, then h -- the 16 rows of an MFMA fragment then differ by the halo's frame pitch, which
// pick_halo_layout pads to a conflict-free residue; otherwise along w, then h, then T)
__device__ static inline void row_cell(const ConvKP& p, int r, int& rt, int& rh, int& rw) {
  const int inner = (p.tfast ? p.Tt : p.Ht) * p.Wt;
  const int o = fdiv(r, p.m_hw), rem = r - o * inner;
  const int i = fdiv(rem, p.m_Wt);
  rw = rem - i * p.Wt;
  rt = p.tfast ? i : o;
  rh = p.tfast ? o : i;
}

// position-class bias row of an output position (flk_conv_args.pos_bias), or nullptr
// (pos_bias_bstride != 0: one table per clip b -- per-clip perturbations)
__device__ static inline const float* pos_bias_row(const ConvKP& p, int b, int ot, int oh, int ow) {
  if (!p.pos_bias) return nullptr;
  const int hc = oh == 0 ? 0 : oh == p.Ho - 1 ? 3 : oh == p.Ho - 2 ? 2 : 1;
  const int wc = ow == 0 ? 0 : ow == p.Wo - 1 ? 3 : ow == p.Wo - 2 ? 2 : 1;
  return p.pos_bias + (size_t)b * p.pos_bias_bstride + (size_t)((ot * 4 + hc) * 4 + wc) * p.cout;
}

// acc * scale, rounded, + bias, rounded -- NEVER one fused multiply-add, in every epilogue variant (finish_store, finish_store_row,
// finish_store_row_pre): left to fp-contract, whether hipcc fuses the two steps depends on the shape of the surrounding code, and the same
// layer run through two kernels (128- against 64-channel tiles, a batch-1 against a batch-8 plan) would differ in the last bit.  (Two
// roundings are also what the oracle's separate multiply and add do.)
__device__ static inline float epi_scale_bias(float v, float sc, float bi, bool has_scale, bool has_bias) {
#pragma clang fp contract(off)
  float t = v;
  if (has_scale) t = t * sc;
  if (has_bias) t = t + bi;
  return t;
}

// the epilogue of EPL consecutive output channels [c0, c0 + EPL) of physical output position opos: v = acc*scale + bias (+ position
// bias) (+ add); relu; mask; 16-byte store into the first or second output segment
template <typename T>
__device__ static inline void finish_store(const ConvKP& p, size_t opos, const float* pb, int c0, float* v) {
  typedef Prec<T> PR;
  constexpr int EPL = PR::EPL;
  const bool hs = p.scale != nullptr, hb = p.bias != nullptr;
#pragma unroll
  for (int e = 0; e < EPL; e += 4) {
    const float4 sc = hs ? *(const float4*)(p.scale + c0 + e) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bi = hb ? *(const float4*)(p.bias + c0 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[e] = epi_scale_bias(v[e], sc.x, bi.x, hs, hb); v[e + 1] = epi_scale_bias(v[e + 1], sc.y, bi.y, hs, hb);
    v[e + 2] = epi_scale_bias(v[e + 2], sc.z, bi.z, hs, hb); v[e + 3] = epi_scale_bias(v[e + 3], sc.w, bi.w, hs, hb);
  }
  if (pb) {
#pragma unroll
    for (int e = 0; e < EPL; e += 4) {
      const float4 t4 = *(const float4*)(pb + c0 + e);
      v[e] += t4.x; v[e + 1] += t4.y; v[e + 2] += t4.z; v[e + 3] += t4.w;
    }
  }
  if (p.add) {
    float a[EPL];
    PR::to_f32(*(const uint4*)(p.add + (opos * p.add_ld + p.add_coff + c0) * sizeof(T)), a);
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] += a[e];
  }
  if (p.relu) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (p.mask) {
    float a[EPL];
    PR::to_f32(*(const uint4*)(p.mask + (opos * p.mask_ld + p.mask_coff + c0) * sizeof(T)), a);
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
  }
  if (c0 < p.cout1) *(uint4*)(p.out + (opos * p.out_ld + p.out_coff + c0) * sizeof(T)) = PR::from_f32(v);
  else *(uint4*)(p.out2 + (opos * p.out2_ld + p.out2_coff + (c0 - p.cout1)) * sizeof(T)) = PR::from_f32(v);
}

// finish_store for the NG store groups (channels c0 + g * 4 * EPL) of ONE output position at once: every add / mask operand of the
// position is requested before the first store.  Called group by group (finish_store), the loads of group g + 1 sit behind the
// store of group g -- the compiler must assume they alias -- and each pays a full memory round trip.
template <typename T, int NG>
__device__ static inline void finish_store_row(const ConvKP& p, size_t opos, const float* pb, int c0, float (&v)[NG][Prec<T>::EPL]) {
  typedef Prec<T> PR;
  constexpr int EPL = PR::EPL;
  uint4 av[NG], mv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL;
    const bool in = c < p.cout;
    if (p.add) av[g] = *(const uint4*)(p.add + (opos * p.add_ld + p.add_coff + (in ? c : 0)) * sizeof(T));
    if (p.mask) mv[g] = *(const uint4*)(p.mask + (opos * p.mask_ld + p.mask_coff + (in ? c : 0)) * sizeof(T));
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL;
    if (c >= p.cout) continue;
    float* w = v[g];
    const bool hs = p.scale != nullptr, hb = p.bias != nullptr;
#pragma unroll
    for (int e = 0; e < EPL; e += 4) {
      const float4 sc = hs ? *(const float4*)(p.scale + c + e) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 bi = hb ? *(const float4*)(p.bias + c + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      w[e] = epi_scale_bias(w[e], sc.x, bi.x, hs, hb); w[e + 1] = epi_scale_bias(w[e + 1], sc.y, bi.y, hs, hb);
      w[e + 2] = epi_scale_bias(w[e + 2], sc.z, bi.z, hs, hb); w[e + 3] = epi_scale_bias(w[e + 3], sc.w, bi.w, hs, hb);
    }
    if (pb) {
#pragma unroll
      for (int e = 0; e < EPL; e += 4) {
        const float4 t4 = *(const float4*)(pb + c + e);
        w[e] += t4.x; w[e + 1] += t4.y; w[e + 2] += t4.z; w[e + 3] += t4.w;
      }
    }
    if (p.add) {
      float a[EPL];
      PR::to_f32(av[g], a);
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] += a[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] = fmaxf(w[e], 0.f);
    }
    if (p.mask) {
      float a[EPL];
      PR::to_f32(mv[g], a);
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] = a[e] > 0.f ? w[e] : 0.f;
    }
    if (c < p.cout1) *(uint4*)(p.out + (opos * p.out_ld + p.out_coff + c) * sizeof(T)) = PR::from_f32(w);
    else *(uint4*)(p.out2 + (opos * p.out2_ld + p.out2_coff + (c - p.cout1)) * sizeof(T)) = PR::from_f32(w);
  }
}

// ---- the epilogue in two halves, so that a caller can request the add / mask operands of SEVERAL output positions before it finishes the
// first: called position by position (finish_store_row*), the operand loads of position i + 1 sit behind the stores of position i -- the
// compiler must assume they alias -- and a wave pays one memory round trip per position with nothing else to do (four per tile in the
// halo and LDS-DMA kernels, four times the store groups in conv_igemm_body's first form: a data-gradient's ReLU mask comes from HBM).
// epi_fetch: the operands of one position's NG store groups (a group past cout reads the row's first channels: valid memory, never used).
template <typename T, int NG>
__device__ static inline void epi_fetch(const ConvKP& p, size_t opos, int c0, uint4 (&av)[NG], uint4 (&mv)[NG]) {
  constexpr int EPL = Prec<T>::EPL;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL;
    const bool in = c < p.cout;
    av[g] = make_uint4(0u, 0u, 0u, 0u); mv[g] = av[g];
    if (p.add) av[g] = *(const uint4*)(p.add + (opos * p.add_ld + p.add_coff + (in ? c : 0)) * sizeof(T));
    if (p.mask) mv[g] = *(const uint4*)(p.mask + (opos * p.mask_ld + p.mask_coff + (in ? c : 0)) * sizeof(T));
  }
}
// finish_store for the NG store groups of one position with scale / bias in registers (sc / bi: EPL / 4 float4 per group) and the add / mask
// operands fetched: acc * scale, + bias (two roundings: epi_scale_bias), + position-class bias, + add, ReLU, mask, store -- finish_store's order
// (SB = false: the caller has applied scale and bias to the accumulators already)
template <typename T, int NG, bool SB = true>
__device__ static inline void finish_store_row_ops(const ConvKP& p, size_t opos, const float* pb, int c0, float (&v)[NG][Prec<T>::EPL],
                                                   const float4 (&sc)[NG][Prec<T>::EPL / 4], const float4 (&bi)[NG][Prec<T>::EPL / 4],
                                                   const uint4 (&av)[NG], const uint4 (&mv)[NG]) {
  typedef Prec<T> PR;
  constexpr int EPL = PR::EPL;
  const bool hs = p.scale != nullptr, hb = p.bias != nullptr;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL;
    if (c >= p.cout) continue;
    float* w = v[g];
    if constexpr (SB) {
#pragma unroll
      for (int h = 0; h < EPL / 4; ++h) {
        const int e = 4 * h;
        w[e] = epi_scale_bias(w[e], sc[g][h].x, bi[g][h].x, hs, hb); w[e + 1] = epi_scale_bias(w[e + 1], sc[g][h].y, bi[g][h].y, hs, hb);
        w[e + 2] = epi_scale_bias(w[e + 2], sc[g][h].z, bi[g][h].z, hs, hb); w[e + 3] = epi_scale_bias(w[e + 3], sc[g][h].w, bi[g][h].w, hs, hb);
      }
    }
    if (pb) {
#pragma unroll
      for (int e = 0; e < EPL; e += 4) {
        const float4 t4 = *(const float4*)(pb + c + e);
        w[e] += t4.x; w[e + 1] += t4.y; w[e + 2] += t4.z; w[e + 3] += t4.w;
      }
    }
    if (p.add) {
      float a[EPL];
      PR::to_f32(av[g], a);
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] += a[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] = fmaxf(w[e], 0.f);
    }
    if (p.mask) {
      float a[EPL];
      PR::to_f32(mv[g], a);
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] = a[e] > 0.f ? w[e] : 0.f;
    }
    if (c < p.cout1) *(uint4*)(p.out + (opos * p.out_ld + p.out_coff + c) * sizeof(T)) = PR::from_f32(w);
    else *(uint4*)(p.out2 + (opos * p.out2_ld + p.out2_coff + (c - p.cout1)) * sizeof(T)) = PR::from_f32(w);
  }
}
// scale / bias of a lane's NG store groups (a group past cout: the first channels, never used; absent: 1 / 0)
template <typename T, int NG>
__device__ static inline void epi_scale_bias_regs(const ConvKP& p, int c0, float4 (&sc)[NG][Prec<T>::EPL / 4], float4 (&bi)[NG][Prec<T>::EPL / 4]) {
  constexpr int EPL = Prec<T>::EPL;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL, cc = c < p.cout ? c : 0;
#pragma unroll
    for (int h = 0; h < EPL / 4; ++h) {
      sc[g][h] = p.scale ? *(const float4*)(p.scale + cc + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);
      bi[g][h] = p.bias ? *(const float4*)(p.bias + cc + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// finish_store_row with the batch-norm scale / bias of the lane's NG store groups already in registers (loaded once per lane: inside the
// row loop hipcc re-requests them behind every store -- it must assume they alias the output -- and waits for each)
template <typename T, int NG>
__device__ static inline void finish_store_row_pre(const ConvKP& p, size_t opos, int c0, float (&v)[NG][Prec<T>::EPL],
                                                   const float4 (&sc)[NG][2], const float4 (&bi)[NG][2]) {
  typedef Prec<T> PR;
  constexpr int EPL = PR::EPL;
  static_assert(EPL == 8, "bf16 only");
  uint4 av[NG], mv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL;
    const bool in = c < p.cout;
    if (p.add) av[g] = *(const uint4*)(p.add + (opos * p.add_ld + p.add_coff + (in ? c : 0)) * sizeof(T));
    if (p.mask) mv[g] = *(const uint4*)(p.mask + (opos * p.mask_ld + p.mask_coff + (in ? c : 0)) * sizeof(T));
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = c0 + g * 4 * EPL;
    if (c >= p.cout) continue;
    float* w = v[g];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = 4 * h;
      const bool hs = p.scale != nullptr, hb = p.bias != nullptr;
      w[e] = epi_scale_bias(w[e], sc[g][h].x, bi[g][h].x, hs, hb); w[e + 1] = epi_scale_bias(w[e + 1], sc[g][h].y, bi[g][h].y, hs, hb);
      w[e + 2] = epi_scale_bias(w[e + 2], sc[g][h].z, bi[g][h].z, hs, hb); w[e + 3] = epi_scale_bias(w[e + 3], sc[g][h].w, bi[g][h].w, hs, hb);
    }
    if (p.add) {
      float a[EPL];
      PR::to_f32(av[g], a);
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] += a[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] = fmaxf(w[e], 0.f);
    }
    if (p.mask) {
      float a[EPL];
      PR::to_f32(mv[g], a);
#pragma unroll
      for (int e = 0; e < EPL; ++e) w[e] = a[e] > 0.f ? w[e] : 0.f;
    }
    if (c < p.cout1) *(uint4*)(p.out + (opos * p.out_ld + p.out_coff + c) * sizeof(T)) = PR::from_f32(w);
    else *(uint4*)(p.out2 + (opos * p.out2_ld + p.out2_coff + (c - p.cout1)) * sizeof(T)) = PR::from_f32(w);
  }
}


// ---- LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves 64 lanes x 16 bytes to 1 KiB of contiguous LDS at m0 + 16 * lane ----
__device__ static inline unsigned lds_addr32(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
// 16 bytes of zeros in device memory (one copy per translation unit): the source of every halo cell that holds no data
static __device__ __attribute__((aligned(16))) unsigned flk_zero16[4];
// (per-lane 64-bit source address)
__device__ static inline void glds16_v64(const char* src, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}
__device__ static inline void glds16(unsigned voff, const char* sbase, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}


// LDS passes beyond one per position-fragment read, summed over the fragments of the tile, for a row enumeration (tfast: w, then T, then h;
// else w, h, T) and a frame pitch FP of the halo image (pick_halo_layout in conv_igemm.hip has the story: a ds_read_b128 serves the 16 lanes
// of one K chunk together, in one pass iff their 16-byte slots differ mod 16)
static inline int conv_halo_extra_passes(const ConvKP& kp, int tfast, int FP) {
  int total = 0;
  for (int r0 = 0; r0 < kp.rows; r0 += 16) {
    int cnt[16] = {}, mx = 0;
    for (int r = r0; r < r0 + 16 && r < kp.rows; ++r) {
      const int inner = (tfast ? kp.Tt : kp.Ht) * kp.Wt, o = r / inner, rem = r % inner, i = rem / kp.Wt, rw = rem % kp.Wt;
      const int rt = tfast ? i : o, rh = tfast ? o : i;
      const int c = ++cnt[(rt * kp.st * FP + rh * kp.sh * kp.Wh + rw * kp.sw) & 15];
      mx = c > mx ? c : mx;
    }
    total += mx - 1;
  }
  return total;
}

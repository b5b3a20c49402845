// Forward of I3D's stem Conv3d_1a_7x7 (7x7x7 / 2, 3 -> 64, SAME; i3d.py:168-170) straight from the resident uint8 clip, with the
// flickering perturbation applied on the way in (kinetics_i3d_utils.py:100-142) -- one kernel instead of
// flk_perturb_apply_s2d (77 MB read, 205 MB written per batch of 8) + the folded 4x4x4 convolution over that tensor.
//
// Why a kernel of its own.  The space-to-depth form (conv_igemm.hip mode 4) runs the 7x7x7x3 = 1029 multiply-accumulates of an
// output value as 49 K steps x 32 = 1568: every 16-byte chunk carries 2 pixels x 3 channels + 2 zeros, and a tap pair that straddles
// the 7-tap window is half empty.  Here K is packed along the pixel ROW instead: the 7 (kw) x 3 (c) = 21 values one (kt, kh) tap row
// contributes to an output position are 21 CONSECUTIVE values of the clip row [.., w, c] -- 3 chunks of 8 with 3 zero-weight slots,
// 49 x 3 = 147 chunks = 37 K steps (1184 executed multiply-accumulates: 87 % useful instead of 66 %).
//
// Alignment is what makes this awkward: output column ow starts its 21 values at byte 12 ow of the bf16 row image -- 8-byte aligned
// for even ow only.  Odd columns therefore start their chunks 2 values EARLIER (8-byte aligned again) and use a second copy of the
// weights shifted by 2 K slots (values 0, 1 and 23 of their 24-slot row have zero weights); the 4 waves of a workgroup split by column
// parity (waves 0, 1: even ow, waves 2, 3: odd ow), so every MFMA B fragment is parity-pure, and an operand chunk is two ds_read_b64.
//
// Workgroup = 256 threads, tile = 4 (t) x 8 (h) x 8 (w) output positions x 64 channels:
//   * halo image in LDS: 13 frames x 21 rows x 128 bytes (64 bf16 = input bytes [6 ow0 - 6, +64) of the row), plane pitch 2720;
//     staged ONCE per tile from the uint8 clip: 8 source bytes per task -> x = u8 * x_scale + x_bias -> the centred perturbed value
//     clamp(x, lo - p, hi - p) (= clip(x + p, lo, hi) - p, flk_apply_args.center = 1; p = adv_flag * clip(delta)/std of the frame and
//     channel; the perturbation itself enters in fp32 through the position-class bias of the epilogue, flk_stem_delta_bias) -> bf16;
//     zero outside the clip (SAME padding: 2 before, 3 after);
//   * lane (q = K chunk, m) of a wave: m = 4 rt + wi -> position (rt, rh = 4 (wave & 1) + fragment, ow = 2 wi + parity): the four
//     B fragments of a wave differ by a constant LDS offset (immediates), the K step's (kt, kh, j) offset is a per-lane-group constant;
//   * bank-conflict-free operand reads: a ds_read_b64 is served in two groups of 32 lanes (q = 0,1 | q = 2,3), 64 banks of 4 bytes.
//     The 4 wi (24 bytes apart) x 4 rt (two planes = 16 banks apart: plane pitch = 32 mod 128) of one q cover 32 banks and a second
//     chunk 32 bytes further (or one plane further) covers exactly the other 32 -- so the 148 chunk slots are PAIRED: (kt, kh, j = 0)
//     with (kt, kh, 2); (kt, kh, 1) with (kt + 1, kh, 1); three left-over pairs conflict two-way (scratch/stem_banks.py);
//   * weights: A fragments of both parities through a double-buffered 2 x 8 KiB LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no
//     register staging, no ds_write in the loop), issued one step ahead right behind the step's barrier, one barrier per K step; the 37
//     steps are fully unrolled (every LDS offset a compile-time constant);
//   * epilogue as conv_igemm.hip: lane = position, 8 consecutive channels per lane group: acc * bn_scale + bias + position-class bias
//     (the perturbation's contribution, exact in fp32), ReLU, 16-byte bf16 stores.
// LDS 52 160 bytes: three workgroups per CU.
#include <stdlib.h>
#include <new>
#include "flk_internal.h"

namespace {

constexpr int SF_TT = 4, SF_WT = 8;
constexpr int SF_TH = 2 * SF_TT + 5;                           // 13 frames
constexpr int SF_PB = 128;                                     // bytes per halo row
constexpr int SF_RING = 2 * 8192;
constexpr int SF_TAB = SF_TH * 32;
constexpr int SF_STEPS = 37;
constexpr int SF_NGRP = 9;                                     // 8-byte source groups per halo row
// NI = position fragments (16 positions each) per wave: the tile is 4 (t) x 2 NI (h) x 8 (w) positions.  NI = 8: a wave multiplies
// 128 positions x 64 channels per K step (32 MFMAs per 12 operand fragments and one 8 KiB ring refill per workgroup: the LDS pipe has
// slack); NI = 4: 64 positions (16 MFMAs per 8 fragments; three workgroups per CU)
template <int NI> struct SfGeo {
  static constexpr int HT = 2 * NI, HH = 2 * HT + 5;           // halo rows per frame: 21 | 37
  static constexpr int PP = HH * SF_PB + 32;                   // plane pitch (= 32 mod 128: see the bank note above)
  static constexpr int HALO = SF_TH * PP;
  static constexpr int TASKS = SF_TH * HH * SF_NGRP;
  static constexpr int LDS = HALO + SF_RING + SF_TAB;
  static constexpr int NTH = 112 / HT;
};

struct SfChunk { int kt, kh, j, valid; };
// K chunk of lane group q in step s (see the pairing above); shared by the host-side weight packing and the kernel
__host__ __device__ constexpr SfChunk sf_chunk(int s, int q) {
  const int p = 2 * s + (q >> 1), second = q & 1;
  if (p < 49) return SfChunk{p / 7, p % 7, second ? 2 : 0, 1};
  if (p < 70) return SfChunk{2 * ((p - 49) / 7) + second, (p - 49) % 7, 1, 1};
  if (p < 73) return SfChunk{6, 2 * (p - 70) + second, 1, 1};
  return SfChunk{6, 6, 1, second ? 0 : 1};                      // the 148th slot: zero weights, its partner's address
}
template <int NI> __host__ __device__ constexpr int sf_stepoff(int s, int q) {
  const SfChunk c = sf_chunk(s, q);
  return c.kt * SfGeo<NI>::PP + c.kh * SF_PB + 16 * c.j;
}

struct StemFwdKP {
  flk_apply_args a;
  const char* w;                 // packed A fragments: [37 steps][2 parities][4 fragments][64 lanes][16 B]
  const float* scale; const float* bias; const float* pos_bias; long pos_bias_bstride;
  char* out; int out_ld;
  int To, nTt, ntiles, chunk;
  int stagger;     // timing experiments only: workgroups 256..767 start (id / 256) * stagger * 8128 cycles late
  int ablate;      // timing experiments only (-DSF_ABLATE builds): bit 1 no staging, 2 no B reads, 3 no A reads, 4 no MFMA
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
__device__ inline int wrapT(int t, int T) { t %= T; return t < 0 ? t + T : t; }
__device__ inline float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// one LDS-DMA wave instruction: lane l moves 16 bytes from sbase + voff (per lane) to LDS byte lds_base + 16 l (lds_base wave-uniform)
__device__ inline void glds16(unsigned voff, const char* sbase, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

template <int OFF>
__device__ inline u32x2 lds_read64(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

__device__ inline unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

#ifdef SF_ABLATE
#define SF_ON(bit) (!(p.ablate & (1 << (bit))))
#else
#define SF_ON(bit) true
#endif

template <int NI>
__global__ __launch_bounds__(256, NI == 4 ? 3 : 2) void stem_fwd_u8_kernel(const StemFwdKP p) {
  typedef SfGeo<NI> G;
  constexpr int SF_HH = G::HH, SF_PP = G::PP, SF_HALO = G::HALO, SF_TASKS = G::TASKS, SF_HT = G::HT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo = smem;
  char* const ring = smem + SF_HALO;
  float* const tab = (float*)(smem + SF_HALO + SF_RING);
  const flk_apply_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;

  // tiles dealt to the XCDs in contiguous chunks (neighbouring tiles share halo rows: one L2), as conv_igemm.hip
  int bid;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    bid = xcd * p.chunk + slot;
    if (slot >= p.chunk || bid >= p.ntiles) return;
  }
  const int tw = bid % 14; bid /= 14;
  const int th = bid % G::NTH; bid /= G::NTH;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int ot0 = tt * SF_TT, oh0 = th * SF_HT, ow0 = tw * SF_WT;

#ifdef SF_ABLATE
  if (p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 768)
    for (int k = 0; k < (int)(blockIdx.x >> 8) * p.stagger; ++k) __builtin_amdgcn_s_sleep(127);
#endif
  // weights of step 0 (in flight while the halo is staged)
  // the 8 KiB of a step go global -> LDS by DMA: wave w moves pieces w and w + 4 (1 KiB each) of the step -- no register staging, no
  // ds_write in the K loop (stamps: the ring-write segment of a step 214 -> 73 ticks, the kernel -5.5 %)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ring_lds = lds_addr(ring);
  const unsigned wvoff0 = (unsigned)(wv * 1024 + lane * 16), wvoff1 = wvoff0 + 4096u;
  glds16(wvoff0, p.w, ring_lds + (unsigned)(wv * 1024));
  glds16(wvoff1, p.w, ring_lds + (unsigned)(wv * 1024) + 4096u);

  // ---- per-frame table: clamp bounds lo - p[c], hi - p[c], source frame, validity ----
  if (tid < SF_TH) {
    const int t = 2 * ot0 - 2 + tid;
    const bool valid = t >= 0 && t < a.T;
    float pv[3] = {0.f, 0.f, 0.f};
    int tx = 0;
    if (valid) {
      tx = wrapT(t - a.shift_x, a.T);                            // x'[t] = x[(t - shift_x) mod T]
      if (a.adv_flag != 0.f) {
        const int ts = wrapT(t - a.shift_p, a.T);                // p'[t] = p[(t - shift_p) mod T]
        const float dc = a.dclip_dev ? a.dclip_dev[b] : a.dclip;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float d = a.delta[(a.delta_per_clip ? b * a.T : 0) * 3 + ts * 3 + c];
          if (dc > 0.f) d = clipf(d, -dc, dc);
          pv[c] = a.adv_flag * (d * a.inv_std[c]);
        }
      }
    }
    float* e = tab + tid * 8;
    e[0] = a.lo - pv[0]; e[1] = a.lo - pv[1]; e[2] = a.lo - pv[2];
    e[3] = a.hi - pv[0]; e[4] = a.hi - pv[1]; e[5] = a.hi - pv[2];
    e[6] = __int_as_float(tx); e[7] = __int_as_float(valid ? 1 : 0);
  }
  __syncthreads();

  // ---- halo staging: task = (halo row R = plane * 21 + row, 8-byte source group grp) ----
  if (SF_ON(1)) {
    const uint8_t* const xb = (const uint8_t*)a.x + (size_t)b * a.T * 224 * 672;
    const int gb0 = 48 * tw - 8;                                  // source byte of group 0 (element -2 of the row image)
    constexpr int NIT = (SF_TASKS + 255) / 256;
#pragma unroll 1
    for (int n0 = 0; n0 < NIT; n0 += 5) {
      uint2 raw[5];
      int meta[5];                                               // (plane << 16) | (grp << 8) | ok ; -1: no task
      int dsto[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int id = tid + 256 * (n0 + k);
        meta[k] = -1; dsto[k] = 0; raw[k] = make_uint2(0u, 0u);
        if (id < SF_TASKS) {
          const int R = id / SF_NGRP, grp = id - R * SF_NGRP;
          const int pl = R / SF_HH, row = R - pl * SF_HH;
          const int h = 2 * oh0 - 2 + row, gb = gb0 + 8 * grp;
          const float* e = tab + pl * 8;
          const bool ok = __float_as_int(e[7]) != 0 && (unsigned)h < 224u && (unsigned)gb < 672u;
          if (ok) raw[k] = *(const uint2*)(xb + ((size_t)__float_as_int(e[6]) * 224 + h) * 672 + gb);
          meta[k] = (pl << 16) | (grp << 8) | (ok ? 1 : 0);
          dsto[k] = pl * SF_PP + row * SF_PB + 16 * grp - 4;
        }
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if (meta[k] < 0) continue;
        const int pl = meta[k] >> 16, grp = (meta[k] >> 8) & 255;
        const bool ok = meta[k] & 1;
        const float* e = tab + pl * 8;
        // byte i of the group has channel (1 + 2 grp + i) mod 3 (the row image starts at byte 48 tw - 8 = 1 mod 3 of a pixel row)
        const int c0 = grp % 3 == 0 ? 1 : grp % 3 == 1 ? 0 : 2;
        const float l0 = e[0], l1 = e[1], l2 = e[2], h0 = e[3], h1 = e[4], h2 = e[5];
        float L[3], H[3];
        L[0] = c0 == 0 ? l0 : c0 == 1 ? l1 : l2; H[0] = c0 == 0 ? h0 : c0 == 1 ? h1 : h2;
        L[1] = c0 == 0 ? l1 : c0 == 1 ? l2 : l0; H[1] = c0 == 0 ? h1 : c0 == 1 ? h2 : h0;
        L[2] = c0 == 0 ? l2 : c0 == 1 ? l0 : l1; H[2] = c0 == 0 ? h2 : c0 == 1 ? h0 : h1;
        const unsigned w2[2] = {raw[k].x, raw[k].y};
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = (float)((w2[i >> 2] >> (8 * (i & 3))) & 255u) * a.x_scale + a.x_bias;
          v[i] = __builtin_amdgcn_fmed3f(x, L[i % 3], H[i % 3]);
        }
        unsigned d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = ok ? pack_bf16(v[2 * i], v[2 * i + 1]) : 0u;
        char* dst = halo + dsto[k];
        if (grp > 0) *(unsigned*)dst = d[0];
        if (grp < 8) {
          *(uint2*)(dst + 4) = make_uint2(d[1], d[2]);
          *(unsigned*)(dst + 12) = d[3];
        }
      }
    }
  }

  // ---- compute plan ----
  const int par = wave >> 1, hsel = wave & 1;
  const int wi = m & 3, rt = m >> 2;
  const unsigned pos0 = lds_addr(halo) + (unsigned)(2 * rt * SF_PP + 2 * NI * hsel * SF_PB + 24 * wi + 8 * par);
  f32x4 acc[4][NI];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 af[4];
  u32x2 bl[NI], bh[NI];
#pragma unroll
  for (int s = 0; s < SF_STEPS; ++s) {
    char* const wcur = ring + (s & 1) * 8192;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step s have landed (issued one step ago)
    __syncthreads();      // everybody's have (step 0: also the halo image); slot (s + 1) & 1 was last read in step s - 1
    if (s + 1 < SF_STEPS) {
      const char* const wb = p.w + (size_t)(s + 1) * 8192;
      const unsigned dst = ring_lds + (unsigned)(((s + 1) & 1) * 8192 + wv * 1024);
      glds16(wvoff0, wb, dst);
      glds16(wvoff1, wb, dst + 4096u);
    }
    const int so = q == 0 ? sf_stepoff<NI>(s, 0) : q == 1 ? sf_stepoff<NI>(s, 1) : q == 2 ? sf_stepoff<NI>(s, 2) : sf_stepoff<NI>(s, 3);
    const unsigned cur = pos0 + (unsigned)so;
    if (s == 0 || SF_ON(3)) {
#pragma unroll
      for (int f = 0; f < 4; ++f) af[f] = *(const bf16x8*)(wcur + par * 4096 + (f * 64 + lane) * 16);
    }
    // B fragments: chunk of fragment i at cur + 2 i rows; two 8-byte reads each (16-byte alignment holds for every other column only)
    if (s == 0 || SF_ON(2)) {
      bl[0] = lds_read64<0>(cur); bh[0] = lds_read64<8>(cur);
      bl[1] = lds_read64<2 * SF_PB>(cur); bh[1] = lds_read64<2 * SF_PB + 8>(cur);
      bl[2] = lds_read64<4 * SF_PB>(cur); bh[2] = lds_read64<4 * SF_PB + 8>(cur);
      bl[3] = lds_read64<6 * SF_PB>(cur); bh[3] = lds_read64<6 * SF_PB + 8>(cur);
      if constexpr (NI == 8) {
        bl[4] = lds_read64<8 * SF_PB>(cur); bh[4] = lds_read64<8 * SF_PB + 8>(cur);
        bl[5] = lds_read64<10 * SF_PB>(cur); bh[5] = lds_read64<10 * SF_PB + 8>(cur);
        bl[6] = lds_read64<12 * SF_PB>(cur); bh[6] = lds_read64<12 * SF_PB + 8>(cur);
        bl[7] = lds_read64<14 * SF_PB>(cur); bh[7] = lds_read64<14 * SF_PB + 8>(cur);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0) ; release %0 %1 %2 %3 %4 %5 %6 %7"
                 : "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3]) :: "memory");
    if constexpr (NI == 8)
      asm volatile("; release %0 %1 %2 %3 %4 %5 %6 %7"
                   : "+v"(bl[4]), "+v"(bh[4]), "+v"(bl[5]), "+v"(bh[5]), "+v"(bl[6]), "+v"(bh[6]), "+v"(bl[7]), "+v"(bh[7]) :: "memory");
    if (SF_ON(4)) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bf16x8 bf = __builtin_bit_cast(bf16x8, (u32x4){bl[i].x, bl[i].y, bh[i].x, bh[i].y});
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f], bf, acc[f][i], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane = position m of fragment i; lane group q owns 8 channels of each 32-channel store group ----
  const int ot = ot0 + rt, ow = ow0 + 2 * wi + par;
  if (ot >= p.To) return;
  const int wc = ow == 0 ? 0 : ow == 111 ? 3 : ow == 110 ? 2 : 1;
  // the batch-norm scale / bias of the lane's 16 channels once, and per output row the position-class bias of both store groups before the
  // first store (written per store group, hipcc re-requested scale and bias behind every store -- it must assume they alias -- and waited
  // for each: 32 exposed round trips per wave)
  float4 sc[2][2], bi[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sc[g][h] = *(const float4*)(p.scale + g * 32 + q * 8 + 4 * h);
      bi[g][h] = *(const float4*)(p.bias + g * 32 + q * 8 + 4 * h);
    }
  // (the position-class bias of row i + 1 is requested BEFORE row i is stored: vmcnt retires in order, so waiting for loads issued behind
  // stores would wait for the stores as well)
  auto load_pb = [&](int i, float4 (&t)[2][2]) {
    const int oh = oh0 + NI * hsel + i;
    const int hc = oh == 0 ? 0 : oh == 111 ? 3 : oh == 110 ? 2 : 1;
    const float* pb = p.pos_bias ? p.pos_bias + (size_t)b * p.pos_bias_bstride + (size_t)((ot * 4 + hc) * 4 + wc) * 64 : nullptr;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h) t[g][h] = pb ? *(const float4*)(pb + g * 32 + q * 8 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4 tq[2][2][2];
  load_pb(0, tq[0]);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int oh = oh0 + NI * hsel + i;
    const size_t opos = ((size_t)(b * p.To + ot) * 112 + oh) * 112 + ow;
    if (i + 1 < NI) load_pb(i + 1, tq[(i + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const float4 (&t4)[2][2] = tq[i & 1];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int c0 = g * 32 + q * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[(g * 8 + e) >> 2][i][(g * 8 + e) & 3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * h;
        v[e] = __fadd_rn(__fmaf_rn(v[e], sc[g][h].x, bi[g][h].x), t4[g][h].x); v[e + 1] = __fadd_rn(__fmaf_rn(v[e + 1], sc[g][h].y, bi[g][h].y), t4[g][h].y);
        v[e + 2] = __fadd_rn(__fmaf_rn(v[e + 2], sc[g][h].z, bi[g][h].z), t4[g][h].z); v[e + 3] = __fadd_rn(__fmaf_rn(v[e + 3], sc[g][h].w, bi[g][h].w), t4[g][h].w);
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf(v[e], 0.f);
      if (SF_ON(5) || v[0] == 12345.f) *(bf16x8*)(p.out + (opos * p.out_ld + c0) * sizeof(bf16_t)) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

// canonical stem weights [7][7][7][3][cout = 64] (DHWIO) -> the 74 "taps" (37 K steps x 2 column parities) x 32 K slots of the
// kernel above, in conv_igemm's A-fragment order (flk_conv_weights_create_impl: nf = 4, one slab)
extern "C" int flk_stem_fwd_u8_weights_create(const float* w7, flk_conv_weights** out) {
  FLK_REQUIRE(w7 && out, "flk_stem_fwd_u8_weights_create: null argument");
  std::vector<float> v((size_t)2 * SF_STEPS * 32 * 64, 0.f);
  for (int s = 0; s < SF_STEPS; ++s)
    for (int par = 0; par < 2; ++par)
      for (int q = 0; q < 4; ++q) {
        const SfChunk c = sf_chunk(s, q);
        if (!c.valid) continue;
        for (int j = 0; j < 8; ++j) {
          const int wnd = 8 * c.j + j - 2 * par;            // odd columns start their chunks 2 values early
          if (wnd < 0 || wnd >= 21) continue;
          const int kw = wnd / 3, ch = wnd % 3;
          const float* src = w7 + ((((size_t)c.kt * 7 + c.kh) * 7 + kw) * 3 + ch) * 64;
          float* dst = &v[(((size_t)(2 * s + par)) * 32 + q * 8 + j) * 64];
          for (int co = 0; co < 64; ++co) dst[co] = src[co];
        }
      }
  return flk_conv_weights_create_impl(v.data(), 2 * SF_STEPS, 1, 1, 32, 64, nullptr, 0, FLK_BF16, 4, 0, out);
}

extern "C" int flk_stem_fwd_u8(const flk_apply_args* a, const flk_conv_weights* w, const float* bn_scale, const float* bn_bias,
                               const float* pos_bias, int64_t pos_bias_bstride, void* out, int out_ld, void* stream) {
  FLK_REQUIRE(a && w && w->dev && bn_scale && bn_bias && out, "flk_stem_fwd_u8: null argument");
  FLK_REQUIRE(a->x && a->delta && a->x_is_u8 && a->center == 1 && !a->delta_dense, "flk_stem_fwd_u8: needs a uint8 clip and a flicker "
              "perturbation applied with center = 1");
  FLK_REQUIRE(a->H == 224 && a->W == 224 && a->T >= 2 && a->T % 2 == 0 && a->B > 0, "flk_stem_fwd_u8: clip must be [B, even T, 224, 224, 3] (got %d,%d,%d,%d)",
              a->B, a->T, a->H, a->W);
  FLK_REQUIRE(a->lo <= a->hi, "flk_stem_fwd_u8: lo > hi");
  FLK_REQUIRE(w->dtype == FLK_BF16 && w->ntaps == 2 * SF_STEPS && w->nf == 4 && w->cout == 64 && w->cin == 32 && w->nslab == 1,
              "flk_stem_fwd_u8: weights are not from flk_stem_fwd_u8_weights_create");
  FLK_REQUIRE(out_ld >= 64 && out_ld % 8 == 0, "flk_stem_fwd_u8: out_ld");
  FLK_REQUIRE(((size_t)a->x & 7) == 0, "flk_stem_fwd_u8: clip must be 8-byte aligned");
  FLK_REQUIRE((size_t)a->B * (a->T / 2) * 112 * 112 * out_ld < (1ull << 31), "flk_stem_fwd_u8: tensor too large");
  StemFwdKP kp{};
  kp.a = *a;
  kp.w = (const char*)w->dev;
  kp.scale = bn_scale; kp.bias = bn_bias; kp.pos_bias = pos_bias; kp.pos_bias_bstride = (long)pos_bias_bstride;
  kp.out = (char*)out; kp.out_ld = out_ld;
  kp.To = a->T / 2; kp.nTt = (kp.To + SF_TT - 1) / SF_TT;
#ifdef SF_ABLATE
  kp.ablate = flk_ablate_env("FLK_SF_ABLATE");
  kp.stagger = flk_ablate_env("FLK_SF_STAGGER");
#endif
  hipStream_t st = (hipStream_t)stream;
  // (NI = 8 -- 128 positions per wave, 32 MFMAs per 12 fragment reads, two workgroups per CU -- measured 0.273 vs 0.258 ms: only in
  //  -DFLK_STEM_NI8 timing builds)
#ifdef FLK_STEM_NI8
  constexpr int NI = 8;
#else
  constexpr int NI = 4;
#endif
  kp.ntiles = a->B * kp.nTt * SfGeo<NI>::NTH * 14;
  kp.chunk = (kp.ntiles + 7) / 8;
  static bool attr_set[FLK_MAX_DEVICES] = {};
  if (int rc = flk_raise_lds_limit((const void*)stem_fwd_u8_kernel<NI>, SfGeo<NI>::LDS, attr_set)) return rc;
  FLK_LAUNCH_KERNEL(stem_fwd_u8_kernel<NI>, dim3((unsigned)(kp.chunk * 8)), dim3(256), SfGeo<NI>::LDS, st, kp);
  FLK_CHECK_HIP(hipGetLastError());
  flk_last_kernel_tag = "stem_fwd_u8_kernel";
  return FLK_OK;
}

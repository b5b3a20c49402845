// Fused delta-gradient of the I3D stem: d(loss)/d(delta[t,c]) straight from the stem's output gradient, on MFMA.
//
// Replaces, for the flickering attack (delta is [T,3]), the stem's data-gradient convolution + the masked reduction over
// (b,h,w) (Conv3DBackpropInputV2 of i3d.py:169 + the clip_by_value / reduce_sum gradients of kinetics_i3d_utils.py:100-142):
//
//   g[t,c] = sum_b sum_{h,w} m[b,t,h,w,c] * gx[b,t,h,w,c],   gx = transposed 7x7x7/2 convolution of G (64 channels),
//   m = 1[lo <= x' + a p' <= hi]   (the clip mask of the perturbed clip)
//
// The data-gradient gx has only 3 output channels per pixel: as an implicit GEMM its N dimension is 3 (24 of 32 after the
// space-to-depth fold, half of the folded taps structurally zero) -- 50 % of the MFMA work is padding and the narrow tile is
// LDS-read-bound (384 TFLOP/s algorithmic in round 1).  But gx is only ever needed SUMMED against the mask, and that sum
// regroups into the weight-gradient form of the same convolution with the MASK as its input:
//
//   g[t,c] = sum_{(ot,kt): 2ot+kt-2=t} sum_{kh,kw,co} Wa[kt,kh,kw,c,co] * C[ot,kt; co; c,kh,kw]
//   C[..]  = sum_{b,oh,ow} G[b,ot,oh,ow,co] * m[b,t,2oh+kh-2,2ow+kw-2,c]
//
// i.e. a GEMM with M = 64 channels x (3-4 output frames), N = 147 = 3*7*7 (padded to 160: 92 % useful), K = output positions:
// no structural zeros, 40 MFMAs per 9 operand reads.  The mask operand is generated on the fly from the resident uint8 / fp32
// clip (never materialised in HBM), the weights enter in fp32 in the epilogue (the bf16 data-gradient rounded them to bf16),
// and neither gx (205 MB at bs 8) nor the space-to-depth clip is read or written.
//
// Two launches:
//  1. stem_mask_kernel (88 MB written at bs 8 x 64 frames): the clip mask de-interleaved -- per (clip, frame, input row h) 6 byte sequences
//     (channel c, pixel parity), byte 9 + i = m[h, 2 i + parity, c] (0x40 = pass, 0x00 = clipped), zero padded = outside the frame.  Column
//     (c, kw) of the B operand at output position ow is byte 9 + ow + (kw>>1) - 1 of sequence (c, kw & 1): the stride-2 tap becomes a byte
//     shift of 0..3 inside 12 ALIGNED bytes.  (Round 2 stored all 7 shifts, 21 rows of 112 bytes per input row: every B fragment was 8
//     consecutive bytes, but the mask was 270 MB and its pre-pass cost the step 0.09 ms; this layout is 3x smaller, the GEMM pays one more
//     ds_read_b32 per fragment, +3 %.)
//  2. stem_delta_grad_kernel: workgroup = (clip b, frame pair t2, chunk of output rows), 8 waves.  Waves 0..6 only run MFMAs: wave (p, q) owns
//     C for output frame ot = t2+1-p and clip frame t = 2*t2+q (tap kt = 2p+q), 64 x 160 accumulators; the odd frame has three taps, so
//     wave 7 has no MFMA work: it is the PRODUCER.  It feeds the G tiles (three K steps ahead, four LDS buffers) and the mask rows (ring of
//     16 input rows per frame) by LDS-DMA (global_load_lds: no VGPR round trip), issued from inline asm so that hipcc inserts no wait
//     of its own, and retires them with ONE counted s_waitcnt vmcnt(N) per step (N = what this step issued: everything older has landed).
//     K step = 32 output positions = 4 runs of 8 consecutive ow:
//       A (G^T): the [32 positions][4 planes][64 ch] tile sits row-major in LDS (128-byte rows; the XOR swizzle of the 16-byte slots is
//                applied on the DMA's SOURCE address) and is read with ds_read_b64_tr_b16 (hardware transpose: lane = channel, 8 positions);
//       B (mask): 12 aligned bytes per lane (ds_read_b64 + ds_read_b32) of which it uses 8, from byte kw>>1 on; a byte is 0x00 / 0x40, so
//                four v_perm_b32 -- their selectors carry the byte shift -- turn them into 8 bf16 values 0.0 / 2.0 (bf16 2.0 = 0x4000: its low
//                byte is zero); the factor 2 is undone in the epilogue.
// Partials are written in the layout of attack.hip's grad_reduce_stage1, so its deterministic stage 2 (batch / chunk sum in a
// fixed order, roll, 1/std, delta-clip mask) is reused unchanged.
#include <stdlib.h>
#include "flk_internal.h"

// timing ablations (WRONG results) only in -DFLK_ABLATE builds; compile-time zero in the product library
#ifdef FLK_ABLATE
#define SG_DBG(bit) (p.dbg & (bit))
#else
#define SG_DBG(bit) (0)
#endif

namespace {

constexpr int SG_THREADS = 512;
constexpr int SG_CO = 64;              // stem output channels
constexpr int SG_NCOL = 147;           // 3 * 7 * 7 columns (c, kh, kw)
constexpr int SG_NPAD = 160;           // padded to 10 fragments of 16
constexpr int SG_NF = SG_NPAD / 16;
constexpr int SG_WO = 112;             // output width (the I3D stem on 224 x 224 frames)
constexpr int SG_MPAD = 9;             // zero bytes in front of a mask sequence: the 8 positions of a run + tap shift s = kw>>1 start at byte 8 + 8 m + s
constexpr int SG_MSEQ = 128;           // one mask sequence (channel c, w parity): 9 zeros | 112 mask bytes | 7 zeros
constexpr int SG_ROWSET = 6 * SG_MSEQ;    // one input row of one frame in HBM: 3 channels x 2 parities = 768 bytes = 48 x 16
constexpr int SG_RPITCH = SG_ROWSET + 48; // pitch of a ring slot in LDS: 204 dwords = 12 banks mod 64 (consecutive rows kh land on different banks)
constexpr int SG_RING = 16;            // input rows kept per frame (9 live + 4 in flight)
#ifndef SG_PD
#define SG_PD 2                        // B-operand prefetch distance in fragments (3, 4: the same time)
#endif
constexpr int SG_MIRROR = 6;           // ring slots 0..5 are kept a second time behind slot 15: the 7 rows kh of a window start at any slot and never wrap
constexpr int SG_RINGP = SG_RING + SG_MIRROR;   // physical slots per frame
constexpr int SG_GTILE = 4 * 32 * 128;    // bytes of one K step's G tile: 4 planes x 32 positions x 64 bf16
#ifndef SG_LA
#define SG_LA 3
#endif
#ifndef SG_SPB
#define SG_SPB 1
#endif
constexpr int SG_LA_ = SG_LA;          // barrier intervals of DMA look-ahead (2 or 3; alone the same time, inside the step 3 is 0.01 ms ahead)
constexpr int SG_SPB_ = SG_SPB;        // K steps per barrier interval (2 with SG_LA = 2: 0.485-0.488 vs 0.482-0.483 ms -- the barriers are not what the loop waits for)
constexpr int SG_GBUFS = SG_SPB_ * (SG_LA_ + 1);   // G tiles
constexpr int SG_OFF_MRING = SG_GBUFS * SG_GTILE;
constexpr int SG_OFF_RED = SG_OFF_MRING + 2 * SG_RINGP * SG_RPITCH;
constexpr int SG_LDS = SG_OFF_RED + 8 * 3 * 4;

struct StemGradKP {
  const char* G; int g_ld;             // bf16 [B][To][Ho][Wo][g_ld], channels [0,64)
  const float* Wf;                     // fp32 [7][160][64]: Wf[kt][c*49+kh*7+kw][co] = W[kt,kh,kw,c,co] * bn_scale[co]; then 1 KiB of zeros
  const char* mask;                    // bytes [B][T][H][21][112] from stem_mask_kernel
  float* partials;
  int B, T, H, To, Ho, Wo;
  int nchunk, rows_per_chunk;
  int dbg;                             // timing experiments only (FLK_SG_DBG): 1 = no mask stream, 2 = no MFMA phase, 4 = no G stream
};

__device__ inline int sg_wrap(int t, int T) { t %= T; return t < 0 ? t + T : t; }
__device__ inline float sg_clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
// perturbation added at frame t (flicker delta only) -- the same expression as attack.hip: pert_at
__device__ inline float sg_pert(const flk_apply_args& a, int b, int t, int c) {
  float d = a.delta[(a.delta_per_clip ? b * a.T : 0) * 3 + sg_wrap(t - a.shift_p, a.T) * 3 + c];
  const float dc = a.dclip_dev ? a.dclip_dev[b] : a.dclip;
  if (dc > 0.f) d = sg_clip(d, -dc, dc);
  return d * a.inv_std[c];
}

// ---- 1. the clip mask, de-interleaved ----
// Per (clip, frame, row) the mask is kept as 6 byte sequences (channel c, pixel parity): byte SG_MPAD + i of sequence (c, par) is the pass
// byte (0x40 = pass, 0x00 = clipped) of pixel w = 2 i + par, zero padded on both sides = beyond the frame.  Column (c, kw) of the GEMM's B
// operand at output position ow is byte SG_MPAD + ow + (kw>>1) - 1 of sequence (c, kw & 1): the stride-2 tap shift is a byte shift the
// GEMM applies itself (12 aligned bytes; the shift goes into the v_perm selectors that widen the bytes).
// One workgroup = SM_ROWS rows of one frame, one pass, one barrier; all index arithmetic is wave-uniform or by constants.  A thread owns 8
// pixels x 3 channels of a row (24 contiguous values), builds the four bytes each of the 6 sequences gets from them in registers and writes
// them as ONE aligned dword per sequence into an LDS image (sequences 3 bytes further right than in HBM: dword aligned); the image then
// leaves as 48 coalesced 16-byte pieces per row, shifted back by v_alignbyte_b32.  HBM: 1 read of the clip, 1.14x that written.
constexpr int SM_ROWS = 8;             // rows per workgroup
constexpr int SM_SEQ = 132;            // LDS image of a sequence: 12 zeros | 112 mask bytes | 8 zeros (33 dwords)
constexpr int SM_GROUPS = SG_WO / 4;   // 28 groups of 8 pixels per row
__global__ __launch_bounds__(256) void stem_mask_kernel(const flk_apply_args a, char* out) {
  __shared__ __attribute__((aligned(16))) unsigned S[SM_ROWS][6 * SM_SEQ / 4];
  const int tid = threadIdx.x;
  const int hblocks = (a.H + SM_ROWS - 1) / SM_ROWS;               // workgroup = (clip b, frame t, rows h0 .. h0 + SM_ROWS): uniform b, t
  const int hb = blockIdx.x % hblocks, bt = blockIdx.x / hblocks, t = bt % a.T, b = bt / a.T;
  const int h0 = hb * SM_ROWS, nr = min(SM_ROWS, a.H - h0);
  if (tid < SM_ROWS * 6 * 5) {                                     // the zero pads: dwords 0..2 and 31..32 of each sequence
    const int r = tid / 30, k = tid % 30, sq = k / 5, e = k % 5;
    S[r][sq * (SM_SEQ / 4) + (e < 3 ? e : 28 + e)] = 0u;
  }
  const int r = tid / SM_GROUPS, j = tid - r * SM_GROUPS;          // this thread's row and pixel group
  if (r < nr) {
    const size_t src = ((((size_t)b * a.T + sg_wrap(t - a.shift_x, a.T)) * a.H + h0 + r) * a.W) * 3 + 24 * j;   // x'[t] = x[(t - shift_x) mod T]
    float xv[24];
    if (a.x_is_u8) {
      unsigned xu[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) xu[k] = *(const unsigned*)((const unsigned char*)a.x + src + 4 * k);
#pragma unroll
      for (int e = 0; e < 24; ++e) xv[e] = (float)((xu[e >> 2] >> (8 * (e & 3))) & 255u) * a.x_scale + a.x_bias;
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float4 f = *(const float4*)((const float*)a.x + src + 4 * k);
        xv[4 * k] = f.x; xv[4 * k + 1] = f.y; xv[4 * k + 2] = f.z; xv[4 * k + 3] = f.w;
      }
    }
    float pv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pv[c] = a.adv_flag * sg_pert(a, b, t, c);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        unsigned d = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u = xv[(2 * k + par) * 3 + c] + pv[c];       // both clip gradients are inclusive at the bounds
          d |= (u >= a.lo && u <= a.hi) ? (0x40u << (8 * k)) : 0u;
        }
        S[r][(c * 2 + par) * (SM_SEQ / 4) + 3 + j] = d;
      }
  }
  __syncthreads();
  constexpr int NPIECE = SG_ROWSET / 16;                           // 48 pieces per row, 8 per sequence
  char* const dst = out + ((size_t)bt * a.H + h0) * SG_ROWSET;
  for (int it = tid; it < nr * NPIECE; it += 256) {
    const int rr = it / NPIECE, pc = it - rr * NPIECE;
    const unsigned* sp = &S[rr][(pc >> 3) * (SM_SEQ / 4) + 4 * (pc & 7)];
    const unsigned d0 = sp[0], d1 = sp[1], d2 = sp[2], d3 = sp[3], d4 = sp[4];
    uint4 o;                                                       // HBM byte g of a sequence = image byte g + 3
    o.x = __builtin_amdgcn_alignbyte(d1, d0, 3u);
    o.y = __builtin_amdgcn_alignbyte(d2, d1, 3u);
    o.z = __builtin_amdgcn_alignbyte(d3, d2, 3u);
    o.w = __builtin_amdgcn_alignbyte(d4, d3, 3u);
    *(uint4*)(dst + (size_t)it * 16) = o;
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
// one LDS-DMA wave instruction: lane l moves 16 bytes from gsrc (per lane) to LDS byte address lds_base + 16 l (lds_base wave-uniform)
__device__ inline void glds16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// ---- 2. the GEMM ----
__global__ __launch_bounds__(SG_THREADS, 2) void stem_delta_grad_kernel(const StemGradKP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const gbuf = smem;
  char* const mring = smem + SG_OFF_MRING;
  float* const red = (float*)(smem + SG_OFF_RED);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = wave & 3, q = wave >> 2;            // output-frame plane, clip frame of the pair
  const int g = lane >> 4, i16 = lane & 15;
  const int T2 = p.T / 2;
  int id = blockIdx.x;
  const int b = id % p.B; id /= p.B;                 // clips fastest: with 8 clips the planes of one clip stay on one XCD's L2
  const int t2 = id % T2;
  const int chunk = id / T2;
  const int oh_lo = chunk * p.rows_per_chunk;
  const int oh_hi = min(p.Ho, oh_lo + p.rows_per_chunk);
  constexpr int gpr = SG_WO / 8;                     // 8-position groups per output row (14; the host checks Wo = 112)
  const int ngroups = (oh_hi - oh_lo) * gpr;
  const int nsteps = (ngroups + 3) >> 2;
  const bool producer = wave == 7;                   // (p = 3, q = 1): the odd frame has three taps kt = 1, 3, 5

  // highest input row K step s touches: its last group's output row, tap kh = 6
  auto hmax_of = [&](int s) {
    int gi = 4 * s + 3;
    if (gi > ngroups - 1) gi = ngroups - 1;
    return 2 * (oh_lo + gi / gpr) + 4;
  };
  const unsigned lds0 = lds_addr(smem);
  const char* const zeros = (const char*)(p.Wf + (size_t)7 * SG_NPAD * SG_CO) + (lane << 4);   // 1 KiB of zeros behind the weights
  // G tile of K step s into buffer buf: 16 DMA instructions of 1 KiB (8 positions x 128 B); lane l lands in row l/8, slot l%8 and
  // therefore fetches chunk (l%8) ^ 2*gsw(row) -- the swizzle the transposing reads below expect.  (Measured: issuing a piece costs
  // its wave 60-100 cycles; spreading the pieces over the consumer waves -- all of them, or only wave 3 whose SIMD carries half the MFMA
  // load -- slowed the MFMA phases by more than the producer gained: 0.70-0.76 ms against 0.64 ms for the whole delta-gradient.)
  // (the plane bases and the lane's share of the offset are computed ONCE: written inside the loop, hipcc redid the 64-bit plane
  // product with ~20 scalar multiplies / adds and a branch per DMA piece.  It did not change the time: without the MFMA phase the loop
  // still takes 0.69 us per step for its ~18 pieces -- ~90 cycles per global_load_lds from ONE wave is the instruction's own rate)
  const char* gplane[4];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    const int ot = t2 + 1 - pp;
    gplane[pp] = ot >= 0 && ot < p.To ? p.G + (size_t)(b * p.To + ot) * p.Ho * p.Wo * p.g_ld * 2 : nullptr;
  }
  unsigned glane[4];                                   // byte offset of this lane inside the 8 positions of group gg (swizzled chunk)
#pragma unroll
  for (int gg = 0; gg < 4; ++gg) {
    const int j = gg * 8 + (lane >> 3);
    const int gsw = ((j >> 1) & 1) | (((j >> 3) & 1) << 1);
    const int ch = (lane & 7) ^ (2 * gsw);
    glane[gg] = (unsigned)(((lane >> 3) * p.g_ld + ch * 8) * 2);
  }
  auto g_issue = [&](int s, int buf) {
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int gi = 4 * s + gg;
      const int r = gi / gpr, col = gi - r * gpr;
      const int oh = oh_lo + r;
      const unsigned pos_off = (unsigned)((oh * p.Wo + col * 8) * p.g_ld * 2) + glane[gg];   // < 2^31: the host checks the tensor size
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const char* src = gplane[pp] && oh < oh_hi ? gplane[pp] + pos_off : zeros;
        glds16(src, lds0 + (unsigned)(buf * SG_GTILE + (pp * 32 + gg * 8) * 128));
      }
    }
  };
  // mask rows h_from..h_to of both frames into their ring slots: 1 DMA instruction (48 x 16 B) per row and frame, a second one into the
  // mirror slot for the slots 0..5; rows outside the frame come from the zero page.  Returns the number of DMA instructions.
  auto mask_issue = [&](int h_from, int h_to) {
    int n = 0;
    for (int h = h_from; h <= h_to; ++h) {
      const bool inside = h >= 0 && h < p.H;
      const int slot = (h + 2) & (SG_RING - 1);
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const char* row = p.mask + (((size_t)b * p.T + 2 * t2 + qq) * p.H + (inside ? h : 0)) * SG_ROWSET;
        const unsigned dst = lds0 + (unsigned)(SG_OFF_MRING + (qq * SG_RINGP + slot) * SG_RPITCH);
        const char* src = inside ? row + (lane << 4) : zeros;
        if (lane < SG_ROWSET / 16) glds16(src, dst);
        if (slot < SG_MIRROR && lane < SG_ROWSET / 16) glds16(src, dst + (unsigned)(SG_RING * SG_RPITCH));
      }
      n += slot < SG_MIRROR ? 4 : 2;
    }
    return n;
  };

  // ---- prologue: the mask rows and G tiles of the first SG_LA barrier intervals ----
  const int h_first = 2 * oh_lo - 2;
  int h_req = hmax_of(min(SG_SPB_ * SG_LA_ - 1, nsteps - 1));  // highest mask row requested
  if (producer) {
    mask_issue(h_first, h_req);
#pragma unroll
    for (int j = 0; j < SG_SPB_ * SG_LA_; ++j)
      if (j < nsteps) g_issue(j, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  float sum[3] = {0.f, 0.f, 0.f};
  if (producer) {
    // ================= producer wave: one barrier per interval of SG_SPB K steps, like the consumers =================
    // Interval d issues the mask rows and the G tiles of interval d + SG_LA, then waits until only the DMA of the last SG_LA - 1 intervals
    // is still in flight: everything issued earlier -- what interval d+1 reads -- has landed (vmcnt retires in issue order).
    int prev = 0;                                      // pieces issued in interval d - 1
    const int nint = (nsteps + SG_SPB_ - 1) / SG_SPB_;
    for (int d = 0; d < nint; ++d) {
      int mine = 0;
#pragma unroll
      for (int j = 0; j < SG_SPB_; ++j) {
        const int st = SG_SPB_ * (d + SG_LA_) + j;
        if (st < nsteps) {
          const int hm = hmax_of(st);
          if (hm > h_req && !SG_DBG(1)) mine += mask_issue(h_req + 1, hm);
          if (hm > h_req) h_req = hm;
          if (!SG_DBG(4)) { g_issue(st, st % SG_GBUFS); mine += 16; }      // that buffer was last read in interval d - 1
        }
      }
      const int allow = SG_LA_ >= 3 ? mine + prev : mine;
      prev = mine;
#define SG_WAIT(n) if (allow >= n) { asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory"); continue; }
      SG_WAIT(56) SG_WAIT(52) SG_WAIT(48) SG_WAIT(44) SG_WAIT(40) SG_WAIT(38) SG_WAIT(36) SG_WAIT(34) SG_WAIT(32) SG_WAIT(30)
      SG_WAIT(28) SG_WAIT(26) SG_WAIT(24) SG_WAIT(22) SG_WAIT(20) SG_WAIT(18) SG_WAIT(16) SG_WAIT(12) SG_WAIT(8) SG_WAIT(6)
      SG_WAIT(4) SG_WAIT(2)
#undef SG_WAIT
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    // ================= consumer waves: MFMA plan of this lane =================
    // A fragments (ds_read_b64_tr_b16): lane 4q'+p' of a 16-lane group supplies row r0+q', columns 4p'..4p'+3 of the 4 x 16 block
    int a_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + (i16 >> 2), pp = i16 & 3;
        const int gsw = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        a_off[i][h] = ((pl * 32 + row) * 8 + ((2 * (i ^ gsw)) + (pp >> 1))) * 16 + 8 * (pp & 1);
      }
    // B fragments: column n = 16 f + i16 -> (c, kh, kw); columns >= 147 read column 146 (their weights are zero)
    int b_off[SG_NF]; unsigned b_selA[SG_NF], b_selB[SG_NF];
#pragma unroll
    for (int f = 0; f < SG_NF; ++f) {
      int n = 16 * f + i16;
      if (n > SG_NCOL - 1) n = SG_NCOL - 1;
      const int c = n / 49, rem = n - c * 49, kh = rem / 7, kw = rem - kh * 7;
      // tap kw of position ow reads pixel 2*ow + kw - 2 = byte SG_MPAD + ow + (kw>>1) - 1 of sequence (c, kw&1); the row kh sits kh slots behind
      // the window's first slot (no wrap: mirror slots); the shift kw>>1 goes into the v_perm selectors
      b_off[f] = (q * SG_RINGP + kh) * SG_RPITCH + (c * 2 + (kw & 1)) * SG_MSEQ + SG_MPAD - 1;
      b_selA[f] = 0x010c000cu + (unsigned)(kw >> 1) * 0x01000100u;
      b_selB[f] = b_selA[f] + 0x02000200u;
    }
    int m_row = 0, m_col = g;                          // this lane group's 8-run: relative output row, 8-group column (gpr >= 4)
    f32x4 acc[4][SG_NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int f = 0; f < SG_NF; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsteps; ++s) {
      if (!SG_DBG(2)) {
        const char* const gb = gbuf + (s % SG_GBUFS) * SG_GTILE;
        bf16x8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(gb + a_off[i][0]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(gb + a_off[i][1]));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          af[i] = __builtin_bit_cast(bf16x8, both);
        }
        int oh = oh_lo + m_row;
        if (oh > oh_hi - 1) oh = oh_hi - 1;                     // padded groups: G is zero there, the address must stay on staged rows
        const int slot0 = (2 * oh) & (SG_RING - 1);             // ring slot of input row 2*oh - 2 (tap kh = 0)
        const char* const mb = mring + m_col * 8 + slot0 * SG_RPITCH;
        // B operands two fragments ahead of their MFMAs (three rotating register sets; left to itself hipcc requests the bytes of two
        // fragments, waits, multiplies, and only then requests the next two: one exposed LDS round trip per pair)
        constexpr int PD = SG_PD;                        // prefetch distance in fragments
        uint2 q01[PD + 1]; unsigned q2[PD + 1];
        auto loadb = [&](int f) { const char* const mp = mb + b_off[f]; q01[f % (PD + 1)] = *(const uint2*)mp; q2[f % (PD + 1)] = *(const unsigned*)(mp + 8); };
#pragma unroll
        for (int f = 0; f < PD; ++f) loadb(f);
#pragma unroll
        for (int f = 0; f < SG_NF; ++f) {
          if (f + PD < SG_NF) loadb(f + PD);
          __builtin_amdgcn_sched_barrier(0);
          const uint2 m01 = q01[f % (PD + 1)];
          const unsigned m2 = q2[f % (PD + 1)];
          // mask bytes s + 0..7 of the 12 (s = kw>>1 <= 3) -> bf16 (m << 8): 0x4000 = 2.0.  v_perm_b32 picks from 8 bytes; selector
          // 0x0c = constant zero (the bf16 low bytes), s+1 | s (bytes 0..4 of {m01.y, m01.x}), s+3 | s+2 (bytes 2..6)
          const unsigned selA = b_selA[f], selB = b_selB[f];
          uint4 bw;
          bw.x = __builtin_amdgcn_perm(m01.y, m01.x, selA);
          bw.y = __builtin_amdgcn_perm(m01.y, m01.x, selB);
          bw.z = __builtin_amdgcn_perm(m2, m01.y, selA);
          bw.w = __builtin_amdgcn_perm(m2, m01.y, selB);
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, bw);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][f], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      m_col += 4;
      if (m_col >= gpr) { m_col -= gpr; ++m_row; }
      if (SG_SPB_ == 1 || (s + 1) % SG_SPB_ == 0 || s == nsteps - 1) __syncthreads();
    }
    // contract the accumulators with the fp32 weights of this wave's tap kt = 2*plane + q
    const float* const wk = p.Wf + (size_t)(2 * pl + q) * SG_NPAD * SG_CO + 4 * g;
#pragma unroll
    for (int f = 0; f < SG_NF; ++f) {
      const int n = 16 * f + i16;
      float sf = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 w = *(const float4*)(wk + (size_t)n * SG_CO + 16 * i);
        sf += acc[i][f][0] * w.x + acc[i][f][1] * w.y + acc[i][f][2] * w.z + acc[i][f][3] * w.w;
      }
      // column n belongs to channel n / 49: fragments 0-2 -> 0, 3 mixed 0|1, 4-5 -> 1, 6 mixed 1|2, 7-9 -> 2 (pad columns: zero weights)
      const int c = n < 49 ? 0 : n < 98 ? 1 : 2;
      if (f <= 2) sum[0] += sf;
      else if (f == 4 || f == 5) sum[1] += sf;
      else if (f >= 7) sum[2] += sf;
      else { sum[0] += c == 0 ? sf : 0.f; sum[1] += c == 1 ? sf : 0.f; sum[2] += c == 2 ? sf : 0.f; }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = sum[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave * 3 + c] = v * 0.5f;                // the mask operand was 2.0
  }
  __syncthreads();
  if (tid < 6) {
    const int qq = tid / 3, c = tid - qq * 3;
    float v = 0.f;
    for (int w = 0; w < 4; ++w) v += red[(qq * 4 + w) * 3 + c];   // fixed order over the taps
    p.partials[(((size_t)b * T2 + t2) * p.nchunk + chunk) * 6 + qq * 3 + c] = v;
  }
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------------------
int flk_grad_reduce_stage2_launch(const flk_apply_args* a, int nchunk, const float* partials, float* gdelta, hipStream_t s);   // attack.hip

// chunks of output rows per (clip, output frame): one 512-thread workgroup fits a CU, so the launch runs in ceil(workgroups / 256)
// rounds of 1 / n of a full K loop each.  n minimises rounds / n, with 3 % per extra chunk for the mask rows chunks re-read: 256 (clip,
// frame) pairs (bs 8, T = 64) keep n = 1; the reference's T = 90 at bs 8 -- 360 pairs, two rounds for 1.41 rounds of work -- takes
// n = 2 (three rounds of half length: 0.74 -> 0.57 ms); 32 pairs (bs 1, T = 64) n = 8 as before.
static int sg_nchunk(int B, int T, int Ho) {
  const int pairs = B * (T / 2);
  int best = 1;
  double best_cost = 1e30;
  for (int n = 1; n <= 16 && n <= Ho; ++n) {
    const double cost = (double)((pairs * n + 255) / 256) / n * (1.0 + 0.03 * n);
    if (cost < best_cost - 1e-12) { best_cost = cost; best = n; }
  }
  return best;
}

static int64_t sg_partial_bytes(int B, int T, int H) {
  (void)H;
  return ((int64_t)B * (T / 2) * 16 * 6 * (int64_t)sizeof(float) + 255) / 256 * 256;      // (16 = the largest chunk count)
}

// scratch = [stage-1 partials | clip mask in B-operand order: B*T*H*768 bytes]
extern "C" int64_t flk_stem_delta_grad_scratch_bytes(int B, int T, int H) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  return sg_partial_bytes(B, T, H) + (int64_t)B * T * H * SG_ROWSET;
}

// w7: the stem's canonical weights [7][7][7][3][64] (kt,kh,kw,c,co); scale: the folded batch-norm scale per output channel.
extern "C" int flk_stem_delta_grad_weights_create(const float* w7, const float* scale, float** out_dev) {
  FLK_REQUIRE(w7 && scale && out_dev, "flk_stem_delta_grad_weights_create: null argument");
  std::vector<float> h((size_t)7 * SG_NPAD * SG_CO + 256, 0.f);   // + 1 KiB of zeros: the DMA source of padded / out-of-range G pieces
  for (int kt = 0; kt < 7; ++kt)
    for (int kh = 0; kh < 7; ++kh)
      for (int kw = 0; kw < 7; ++kw)
        for (int c = 0; c < 3; ++c)
          for (int co = 0; co < SG_CO; ++co)
            h[((size_t)kt * SG_NPAD + c * 49 + kh * 7 + kw) * SG_CO + co] = w7[((((size_t)kt * 7 + kh) * 7 + kw) * 3 + c) * SG_CO + co] * scale[co];
  float* d = nullptr;
  FLK_CHECK_HIP(hipMalloc((void**)&d, h.size() * sizeof(float)));
  FLK_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  *out_dev = d;
  return FLK_OK;
}

extern "C" int flk_stem_delta_grad_weights_destroy(float* dev) {
  if (dev) (void)hipFree(dev);
  return FLK_OK;
}

static int sg_check(const flk_apply_args* a) {
  FLK_REQUIRE(a && a->x && a->delta, "flk_stem_delta_grad: null argument");
  FLK_REQUIRE(!a->delta_dense, "flk_stem_delta_grad: flicker perturbation [T,3] only (the dense attack needs the per-pixel gradient)");
  FLK_REQUIRE(a->B > 0 && a->T >= 2 && a->T % 2 == 0 && a->H > 0 && a->H % 2 == 0 && a->W == 224,
              "flk_stem_delta_grad: T, H must be even and W = 224 (the I3D stem; got %d, %d, %d)", a->T, a->H, a->W);
  FLK_REQUIRE(a->lo <= a->hi, "flk_stem_delta_grad: lo > hi");
  return FLK_OK;
}

// step 1 alone: the clip mask into the scratch (it depends on the clip and on delta only, so a caller may run it on another stream
// while the rest of the backward pass is still busy -- flk_net_backward_delta does)
extern "C" int flk_stem_delta_grad_mask(const flk_apply_args* a, float* scratch, void* stream) {
  int rc = sg_check(a);
  if (rc) return rc;
  FLK_REQUIRE(scratch, "flk_stem_delta_grad_mask: null scratch");
  char* const mask = (char*)scratch + sg_partial_bytes(a->B, a->T, a->H);
  FLK_LAUNCH_KERNEL(stem_mask_kernel, dim3((unsigned)((long)a->B * a->T * ((a->H + SM_ROWS - 1) / SM_ROWS))), dim3(256), 0, (hipStream_t)stream, *a, mask);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// chunks per frame pair for launches of nb clips each (per-clip perturbations: the chunking of a batch-1 call)
static void sg_chunks(const flk_apply_args* a, int nb, int Ho, int& nchunk, int& rows_per_chunk) {
  nchunk = sg_nchunk(a->delta_per_clip ? 1 : nb, a->T, Ho);
#ifdef FLK_ABLATE
  if (const int e = flk_ablate_env("FLK_SG_NCHUNK")) nchunk = e < 16 ? e : 16;
#endif
  rows_per_chunk = (Ho + nchunk - 1) / nchunk;
  nchunk = (Ho + rows_per_chunk - 1) / rows_per_chunk;       // no empty chunks
}

// the GEMM for clips [b0, b0 + nb) of the batch (mask in the scratch already; stage-1 partials of those clips into their slots of the scratch).
// A batch may be covered by several such launches of EQUAL nb (on different streams), followed by ONE flk_stem_delta_grad_finish with the same nb.
int flk_stem_delta_grad_part(const flk_apply_args* a, int b0, int nb, const void* G, int g_ld, const float* wf_dev, float* scratch, hipStream_t s) {
  int rc = sg_check(a);
  if (rc) return rc;
  FLK_REQUIRE(G && wf_dev && scratch, "flk_stem_delta_grad: null argument");
  FLK_REQUIRE(g_ld >= SG_CO && g_ld % 8 == 0, "flk_stem_delta_grad: bad channel stride %d", g_ld);
  FLK_REQUIRE(b0 >= 0 && nb > 0 && b0 + nb <= a->B, "flk_stem_delta_grad: clips [%d, %d) of %d", b0, b0 + nb, a->B);
  StemGradKP kp{};
  kp.B = nb; kp.T = a->T; kp.H = a->H;
  kp.To = a->T / 2; kp.Ho = a->H / 2; kp.Wo = a->W / 2;
  FLK_REQUIRE((size_t)a->B * kp.To * kp.Ho * kp.Wo * g_ld < (1ull << 31), "flk_stem_delta_grad: tensor too large");
  sg_chunks(a, nb, kp.Ho, kp.nchunk, kp.rows_per_chunk);
  // every use of the clip index in the kernel is linear in it: a slice is the same kernel on shifted bases
  kp.G = (const char*)G + (size_t)b0 * kp.To * kp.Ho * kp.Wo * g_ld * 2; kp.g_ld = g_ld; kp.Wf = wf_dev;
  kp.partials = scratch + (size_t)b0 * (a->T / 2) * kp.nchunk * 6;
  kp.mask = (const char*)scratch + sg_partial_bytes(a->B, a->T, a->H) + (size_t)b0 * a->T * a->H * SG_ROWSET;
#ifdef FLK_ABLATE
  kp.dbg = flk_ablate_env("FLK_SG_DBG");
#endif
  static bool attr_set[FLK_MAX_DEVICES] = {};
  if ((rc = flk_raise_lds_limit((const void*)stem_delta_grad_kernel, SG_LDS, attr_set))) return rc;
  FLK_LAUNCH_KERNEL(stem_delta_grad_kernel, dim3((unsigned)(nb * kp.To * kp.nchunk)), dim3(SG_THREADS), SG_LDS, s, kp);
  FLK_CHECK_HIP(hipGetLastError());
  flk_last_kernel_tag = "stem_delta_grad_kernel";
  return FLK_OK;
}

// stage 2 over the whole batch, behind launches of nb clips each
int flk_stem_delta_grad_finish(const flk_apply_args* a, int nb, float* scratch, float* gdelta, hipStream_t s) {
  int rc = sg_check(a);
  if (rc) return rc;
  FLK_REQUIRE(scratch && gdelta && nb > 0 && a->B % nb == 0, "flk_stem_delta_grad_finish: bad argument");
  int nchunk, rows;
  sg_chunks(a, nb, a->H / 2, nchunk, rows);
  flk_apply_args a2 = *a;
  a2.fold_t = 2;                                       // stage 2 reads the partials as (frame pair, parity)
  return flk_grad_reduce_stage2_launch(&a2, nchunk, scratch, gdelta, s);
}

// G: bf16 [B][T/2][H/2][W/2][g_ld] = d(loss)/d(pre-ReLU stem output) (64 channels); gdelta: [T,3] fp32.  mask_done != 0: the scratch
// already holds the mask of these arguments (flk_stem_delta_grad_mask, ordered before this call by the caller).
extern "C" int flk_stem_delta_grad(const flk_apply_args* a, const void* G, int g_ld, const float* wf_dev, float* gdelta,
                                   float* scratch, int mask_done, void* stream) {
  int rc = sg_check(a);
  if (rc) return rc;
  FLK_REQUIRE(G && wf_dev && gdelta && scratch, "flk_stem_delta_grad: null argument");
  if (!mask_done && (rc = flk_stem_delta_grad_mask(a, scratch, stream))) return rc;
  if ((rc = flk_stem_delta_grad_part(a, 0, a->B, G, g_ld, wf_dev, scratch, (hipStream_t)stream))) return rc;
  return flk_stem_delta_grad_finish(a, a->B, scratch, gdelta, (hipStream_t)stream);
}

// ---- exact perturbation path of the stem's FORWARD in bf16 mode (flk_apply_args.center, flk_conv_args.pos_bias) -------------------
// class of an output index o in [0, n): which taps k of the 7-tap / stride-2 / pad-before-2 window stay inside the frame
//   0: o == 0 (k >= 2)   1: interior (all)   2: o == n-2 (k <= 5)   3: o == n-1 (k <= 3)
static inline bool sb_tap_valid(int cls, int k) { return cls == 0 ? k >= 2 : cls == 2 ? k <= 5 : cls == 3 ? k <= 3 : true; }

extern "C" int flk_stem_delta_bias_weights_create(const float* w7, const float* scale, float** out_dev) {
  FLK_REQUIRE(w7 && scale && out_dev, "flk_stem_delta_bias_weights_create: null argument");
  std::vector<float> h((size_t)7 * 16 * 3 * SG_CO, 0.f);           // [kt][hc][wc][c][co]
  for (int kt = 0; kt < 7; ++kt)
    for (int hc = 0; hc < 4; ++hc)
      for (int wc = 0; wc < 4; ++wc)
        for (int kh = 0; kh < 7; ++kh)
          for (int kw = 0; kw < 7; ++kw) {
            if (!sb_tap_valid(hc, kh) || !sb_tap_valid(wc, kw)) continue;
            for (int c = 0; c < 3; ++c)
              for (int co = 0; co < SG_CO; ++co)
                h[((((size_t)kt * 4 + hc) * 4 + wc) * 3 + c) * SG_CO + co] += w7[((((size_t)kt * 7 + kh) * 7 + kw) * 3 + c) * SG_CO + co] * scale[co];
          }
  float* d = nullptr;
  FLK_CHECK_HIP(hipMalloc((void**)&d, h.size() * sizeof(float)));
  FLK_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  *out_dev = d;
  return FLK_OK;
}

namespace {
// table[ot][hc][wc][co] = sum_{kt: frame 2*ot+kt-2 inside the clip} sum_c a*p'[t,c] * S[kt][hc][wc][c][co]
// (delta_per_clip: blockIdx.y = clip, one table per clip, (T/2)*16*64 floats apart)
__global__ __launch_bounds__(64) void stem_delta_bias_kernel(const flk_apply_args a, const float* S, float* tab) {
  const int co = threadIdx.x, cls = blockIdx.x & 15, ot = blockIdx.x >> 4, b = blockIdx.y;
  float acc = 0.f;
  for (int kt = 0; kt < 7; ++kt) {
    const int t = 2 * ot + kt - 2;
    if (t < 0 || t >= a.T) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) acc += a.adv_flag * sg_pert(a, b, t, c) * S[(((size_t)kt * 16 + cls) * 3 + c) * SG_CO + co];
  }
  tab[(((size_t)b * (a.T / 2) + ot) * 16 + cls) * SG_CO + co] = acc;
}
}  // namespace

extern "C" int flk_stem_delta_bias(const flk_apply_args* a, const float* sums_dev, float* table_out, void* stream) {
  FLK_REQUIRE(a && a->delta && sums_dev && table_out, "flk_stem_delta_bias: null argument");
  FLK_REQUIRE(!a->delta_dense && a->T >= 2 && a->T % 2 == 0, "flk_stem_delta_bias: flicker perturbation [T,3], even T");
  FLK_LAUNCH_KERNEL(stem_delta_bias_kernel, dim3((unsigned)(a->T / 2 * 16), (unsigned)(a->delta_per_clip ? a->B : 1)), dim3(64), 0, (hipStream_t)stream,
                     *a, sums_dev, table_out);
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

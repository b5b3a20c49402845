// 3-D convolution as implicit GEMM on the CDNA4 matrix cores, with LDS-staged T x H x W halo tiles.
//
// One kernel serves every convolution of I3D (Unit3D, i3d.py:51-71) and VideoResNet, forward and
// data-gradient: taps, strides, pads, channel slices and the output-position map are runtime
// parameters; only the element type and the output-channel tile width are template parameters.
//
// Mapping (per 256-thread workgroup = 4 waves):
//   * tile = up to 256 output positions (a Tt x Ht x Wt box) x BN = 16*NF output channels;
//   * K loop = input-channel slabs (64 bytes of channels per position) x taps;
//   * per slab the input HALO box ((Tt-1)*st+kt) x ... is staged ONCE into LDS and every tap reads
//     it at a constant byte offset (27x less L2 traffic than a per-tap im2col gather);
//   * LDS halo image = 4 planes [16-byte channel chunk][halo position]: the 16 positions x 4 chunks
//     one ds_read_b128 wave-instruction touches are bank-conflict-free for consecutive positions
//     (planes 0/1 and 2/3 aligned mod 256 B, the pairs 32 B apart so the staging writes are 2-way);
//   * weights arrive pre-packed in MFMA A-fragment order (flk_internal.h); how they reach the MFMAs depends on the
//     launch (template MODE, chosen in flk_conv3d):
//       0  wide wave tiles: per (slab, tap) through a double-buffered LDS tile shared by the 4 waves, register prefetch
//          one step ahead, one barrier per step;
//       1  narrow wave tiles / small grids ("direct A"): every wave streams its own fragments into a register queue
//          4-8 steps deep (inline-asm loads, hand-counted vmcnt), no per-step barrier;
//       2  mode 1 for 1x1x1 convolutions, activation slabs prefetched in depth as well;
//       3  1x1x1 convolutions with wide wave tiles: LDS ring as in 0, activations and weights in register queues with
//          ONE counted vmcnt per step;
//       4  the folded 7x7x7 stem: K steps assembled from the structurally non-zero 16-byte chunks only;
//   * MFMA orientation: A = weights (rows = output channels), B = activations (cols = positions),
//     so an accumulator lane owns ONE position and 4*NF consecutive channels -> 16-byte epilogue
//     accesses for scale/bias/add/mask and stores.
//   * bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32 (exact fp32, parity mode).
#include <stdlib.h>
#include <algorithm>
#include "flk_internal.h"
#include "conv_common.h"

constexpr int NPAIR = (FLK_MAX_HALO * 4 + 255) / 256;  // (position, chunk) pairs staged per thread (16)

// WN = waves along N: the 4 waves form a (4/WN) x WN grid; the workgroup tile is 64*(4/WN) rows x 16*NF channels and
// every wave owns 64 rows x 16*NF/WN channels.  WN > 1 trades weight re-streaming for more workgroups on the layers
// with few output positions (Mixed_4*/Mixed_5*).
// MODE 0: weights through the LDS ring (wide wave tiles).  MODE 1: every wave streams its own fragments into registers
// ("direct A").  MODE 2: direct A for a 1x1x1 convolution (activation slabs prefetched in depth as well).  MODE 3: LDS ring
// for a 1x1x1 convolution with wide wave tiles, activations and weights prefetched in depth with hand-counted waits.
// MODE 4: the folded 7x7x7 stem (4x4x4 taps over ONE 32-channel slab): K steps built from the non-zero 16-byte chunks only.
// The kernel body: block (bx, by) of the launch described by p.  WN = 0 is the GROUPED form (conv_igemm_group_kernel: several
// convolutions in one grid): NF then counts the channel fragments of ONE WAVE, the waves-along-N count is the member's run-time p.wn
// and the workgroup tile is 16 * NF * p.wn channels wide -- direct-A weights (MODE 1) only, where the tile width appears in two
// address computations and nowhere in the loop structure.
// timing experiments on the row-ahead ring loop (-DCONV_ABLATE=bits builds only; WRONG results): 1 no weight stream (neither the row-ahead
// loads nor their waits: the ring is written from registers that hold zeros), 2 no barrier per step, 4 no ring write, 8 no MFMAs, 16 no
// position-fragment reads, 32 no weight-fragment reads, 64 no halo staging of the large-halo forms (the image keeps whatever LDS held).  The product
// build compiles every CAB() to true.
// (The substitutes for skipped reads are ZERO fragments, and bit 1 drops the LOADS with the waits: a substitute taken from a weight-queue
//  register, and a ring write of a register whose load nothing had waited for, are what faulted in round 4 -- DESIGN.md.  Variants are built
//  with tools/build_variant.py, which puts them through tools/audit_asm_loads.py with the same flags and refuses to link a flagged one;
//  tests/test_asm_audit_cpu.py audits bits 16, 32, 48 and 63.)
#ifdef CONV_ABLATE
#define CAB(bit) (!((CONV_ABLATE) & (bit)))
#else
#define CAB(bit) true
#endif
template <typename T, int NF, int WN, int MODE>
__device__ __forceinline__ void conv_igemm_body(const ConvKP& p, const int bx, const int by) {
  constexpr bool K1 = MODE == 2 || MODE == 3;
  constexpr bool RTWN = WN == 0;
  static_assert(!RTWN || MODE == 1, "run-time wave layouts: direct-A weights only");
  typedef Prec<T> PR;
  typedef typename PR::frag frag;
  constexpr int EPL = PR::EPL;
  constexpr int SLABC = 4 * EPL;
  constexpr int WCH = NF >= 4 ? (NF + 3) / 4 : 1;   // 16-byte weight chunks per thread per (slab, tap)
  // NF = 6 (96-channel tiles): the step's 6 KiB are 384 chunks -- threads 0..127 move a second one, the others re-read their first
  // (a valid address) and do not store it
  const bool w2ok = NF % 4 == 0 || threadIdx.x + 256 < NF * 64;
  const int w2off = w2ok ? 4096 : 0;
  constexpr int NFW = RTWN ? NF : NF / (RTWN ? 1 : WN);   // channel fragments per wave
  const int WNr = RTWN ? p.wn : WN;                       // waves along N
  const int WM = 4 / WNr;                                 // waves along M
  const int NFT = NFW * WNr;                              // channel fragments of the workgroup tile (= NF unless grouped)
  static_assert(RTWN || (NF % (RTWN ? 1 : WN) == 0), "channel tile does not divide over the waves");
  static_assert(4 * NFW >= EPL, "wave tile too narrow for 16-byte epilogue groups");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // small halos (P <= 256: all 1x1x1 convolutions) keep TWO halo images so the K loop needs one barrier per slab
  const bool small_halo = p.P <= 256;
  const int halo_bytes = 4 * p.plane_b + 64;
  char* const halo = smem;
  char* const wbuf = smem + (small_halo ? 2 : 1) * halo_bytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  const int wm = wave % WM, wn = wave / WM;

  // 1-D grid.  Workgroups that share an activation tile (the N tiles of one position tile) get ids 8 apart: blocks b and
  // b + 8 land on the same XCD (round-robin dispatch), so the tile is fetched into that XCD's L2 once.  Speed only.
  int bid, ntile;
  if (p.xcd_chunk > 0) {
    // chunked form: XCD x (= id % 8 under round-robin dispatch) works through the position tiles [x * chunk, (x + 1) * chunk) in order,
    // all N tiles of a position tile back to back -- neighbouring tiles, whose halos overlap, meet in ONE L2 instead of eight
    const int id = bx, xcd = id & 7, slot = id >> 3;
    const int k = slot / p.ntile_n;
    ntile = slot - k * p.ntile_n;
    bid = xcd * p.xcd_chunk + k;
  } else {
    const int per = 8 * p.ntile_n, id = bx;
    const int grp = id / per, r = id - grp * per;
    ntile = r >> 3;
    bid = grp * 8 + (r & 7);
  }
  if (bid >= p.B * p.nTt * p.nTh * p.nTw) return;     // tail of the last group of 8 (whole workgroup, before any barrier)
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  const int b = bid / p.nTt;
  const int ot0 = tt * p.Tt, oh0 = th * p.Ht, ow0 = tw * p.Wt;
  const int it0 = ot0 * p.st - p.pt, ih0 = oh0 * p.sh - p.ph, iw0 = ow0 * p.sw - p.pw;

  // ---- staging plan: pair n of this thread = (halo position (tid>>2) + 64 n, chunk tid&3) ----
  // Large halos of the multi-tap forms (DMAH): by LDS-DMA instead -- wave w fills chunk plane w, one wave-instruction per block of 64
  // consecutive cells, every lane with its own source address (a block of zeros for padding and for the cells no tap reads): pair n of a
  // lane = (halo position lane + 64 n, chunk wave).  No staging registers, no selects, no ds_write_b128 (13 cycles of the SIMD's path to
  // the LDS each); measured cost of the register form on these launches: 11-17 % (-DCONV_ABLATE=64).  conv_pc.hip stages the same way.
  constexpr bool DMAH = MODE == 0 || MODE == 1 || MODE == 5 || MODE == 6;
  const bool dma_halo = DMAH && !small_halo;
  const int ch = dma_halo ? wave : (tid & 3);
  constexpr int NPK = K1 ? 4 : NPAIR;   // 1x1x1 tiles have P <= 256: 4 pairs per thread
#ifndef FLK_HALO_BATCH
#define FLK_HALO_BATCH 4
#endif
  constexpr int HB = K1 ? 4 : FLK_HALO_BATCH;   // halo pieces requested together when a slab is staged (NPK % HB == 0); 8 / 16: the same times
                                                // (Conv3d_2c 0.240 / 0.241 / 0.243 ms, the step 6.09-6.11 ms for all three)
  int goff[NPK];  // linear input position of the pair, -1 = zero fill (padding), -2 = beyond the halo
  {
    const int HW = p.Hh * p.Wh;
#pragma unroll
    for (int n = 0; n < NPK; ++n) {
      const int hp = dma_halo ? lane + 64 * n : (tid >> 2) + 64 * n;
      int g = -2;
      if (hp < p.P) {
        const int a = fdiv(hp, p.m_HW), rem = hp - a * p.FP;
        if (rem < HW) {                                   // (the slots between two frames, FP > Hh * Wh, are never read)
          const int bq = fdiv(rem, p.m_Wh), c = rem - bq * p.Wh;
          const int it = it0 + a, ih = ih0 + bq, iw = iw0 + c;
          g = -1;
          if ((unsigned)it < (unsigned)p.Ti && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi)
            g = ((b * p.Ti + it) * p.Hi + ih) * p.Wi + iw;
        }
      }
      goff[n] = g;
    }
  }
  char* const hdst = halo + plane_off(ch, p.plane_b) + (tid >> 2) * 16;

  // ---- compute plan: this wave owns tile rows [64*wave, 64*wave+64), 4 fragments of 16 positions ----
  const bool wave_active = wm * 64 < p.rows;
  int rowpos[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + i * 16 + m;
    int pos = 0;
    if (r < p.rows) {
      int rt, rh, rw;
      row_cell(p, r, rt, rh, rw);
      pos = rt * p.st * p.FP + rh * p.sh * p.Wh + rw * p.sw;
    }
    rowpos[i] = pos * 16 + plane_off(q, p.plane_b);
  }

  f32x4 acc[NFW][4];
#pragma unroll
  for (int f = 0; f < NFW; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // src = segment base + (coff + slab*SLABC + ch*EPL) elements; ld = that segment's channel stride
  auto ldhalo = [&](const char* src, int ld, int g, bool chvalid) -> uint4 {
    const bool ok = g >= 0 && chvalid;
    uint4 v = *(const uint4*)(src + (size_t)(ok ? g : 0) * ld * sizeof(T));   // position 0 is always readable
    if (!ok) v = make_uint4(0, 0, 0, 0);
    return v;
  };
  // slab s -> (segment base pointer incl. channel offset, ld, chunk validity)
  // split-K: this workgroup reduces the slabs [s_lo, s_lo + nslab) only (block y = slice; one slice = everything)
  const int s_lo = by * p.nslab / p.ksplit;
  const int nslab = (by + 1) * p.nslab / p.ksplit - s_lo;
  auto slab_src = [&](int s, const char*& src, int& ld) -> bool {
    s += s_lo;
    if (s < p.nslab1) {
      src = p.in + (size_t)(p.in_coff + s * SLABC + ch * EPL) * sizeof(T); ld = p.in_ld;
      return s * SLABC + ch * EPL < p.cin1;
    }
    const int s2 = s - p.nslab1;
    src = p.in2 + (size_t)(p.in2_coff + s2 * SLABC + ch * EPL) * sizeof(T); ld = p.in2_ld;
    return s2 * SLABC + ch * EPL < p.cin - p.cin1;
  };
  // slab s into the image at byte offset img of the halo area, by LDS-DMA (dma_halo); complete for this wave at return, for the others
  // behind the next barrier
  auto stage_dma = [&](int s) {
    const char* src; int ld;
    const bool chvalid = slab_src(s, src, ld);
    const unsigned lb = lds_addr32(halo) + (unsigned)plane_off(ch, p.plane_b);
#pragma unroll
    for (int n = 0; n < NPK; ++n) {
      if (n * 64 >= p.P || !CAB(64)) break;
      const char* const a = (goff[n] >= 0 && chvalid) ? src + (size_t)goff[n] * ld * sizeof(T) : (const char*)flk_zero16;
      glds16_v64(a, (unsigned)__builtin_amdgcn_readfirstlane((int)(lb + (unsigned)(n * 1024))));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const size_t wstep = (size_t)p.cout_frags * 1024;
  const int nsteps = nslab * p.ntaps;
  const char* const wbase = p.w + (size_t)s_lo * p.ntaps * wstep;
  if constexpr (MODE == 0) {
    // ---- wide wave tiles (>= 32 MFMAs per K step): weights through a double-buffered LDS tile shared by the 4 waves,
    // register prefetch one step ahead, nested tap loops.
    // weight stream: (slab, tap) step k lives at w + (k*cout_frags + ntile*NF) KiB.  Every thread moves WCH 16-byte
    // chunks per step, unconditionally (NF=2: threads t and t+128 move the same chunk) so the prefetch stays in VGPRs.
    const int wchunk = NF >= 4 ? tid : (tid & 127);
    const char* wsrc = wbase + (size_t)ntile * NF * 1024 + wchunk * 16;
    uint4 wreg0 = *(const uint4*)wsrc, wreg1 = make_uint4(0, 0, 0, 0);   // named scalars: an array here is demoted to scratch
    if (WCH == 2) wreg1 = *(const uint4*)(wsrc + w2off);
    wsrc += wstep;

    int it_w = 0;
    // Small halos (every 1x1x1 convolution, small tiles: P <= 256 -> 4 pairs per thread) are software-pipelined:
    // slab s+1 is fetched into named registers while slab s computes, so the K loop does not expose one global-load
    // latency per slab.  Larger halos are staged after the barrier (amortised over kt*kh*kw taps; the second resident
    // workgroup covers the stall).
    uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = pre0, pre2 = pre0, pre3 = pre0;
    auto prefetch = [&](int s) {
      const char* src; int ld;
      const bool chvalid = slab_src(s, src, ld);
      pre0 = ldhalo(src, ld, goff[0], chvalid); pre1 = ldhalo(src, ld, goff[1], chvalid);
      pre2 = ldhalo(src, ld, goff[2], chvalid); pre3 = ldhalo(src, ld, goff[3], chvalid);
    };
    if (small_halo) prefetch(0);
    for (int s = 0; s < nslab; ++s) {
      const int hsel = small_halo ? (s & 1) * halo_bytes : 0;
      if (small_halo) {
        // image (s & 1) was last read while computing slab s-2; every wave has passed the barrier of slab s-1 since
        char* const hd = hdst + hsel;
        if (goff[0] != -2) *(uint4*)(hd) = pre0;
        if (goff[1] != -2) *(uint4*)(hd + 1024) = pre1;
        if (goff[2] != -2) *(uint4*)(hd + 2048) = pre2;
        if (goff[3] != -2) *(uint4*)(hd + 3072) = pre3;
        if (s + 1 < nslab) prefetch(s + 1);
      } else {
        __syncthreads();  // every wave has finished reading the previous slab's halo
        stage_dma(s);
      }
      int tapoff_t = 0;
      for (int dt = 0; dt < p.kt; ++dt, tapoff_t += p.FP * 16) {
        int tapoff_h = tapoff_t;
        for (int dh = 0; dh < p.kh; ++dh, tapoff_h += p.Wh * 16) {
          int tapoff = tapoff_h;
          for (int dw = 0; dw < p.kw; ++dw, tapoff += 16) {
            char* const wcur = wbuf + (it_w & 1) * (NF * 1024);
            *(uint4*)(wcur + wchunk * 16) = wreg0;
            if (WCH == 2 && w2ok) *(uint4*)(wcur + wchunk * 16 + 4096) = wreg1;
            ++it_w;
            if (it_w < nsteps) {
              wreg0 = *(const uint4*)wsrc;
              if (WCH == 2) wreg1 = *(const uint4*)(wsrc + w2off);
              wsrc += wstep;
            }
            __syncthreads();
            if (wave_active) {
              frag bf[4], af[NFW];
  #pragma unroll
              for (int i = 0; i < 4; ++i) bf[i] = *(const frag*)(halo + hsel + rowpos[i] + tapoff);
  #pragma unroll
              for (int f = 0; f < NFW; ++f) af[f] = *(const frag*)(wcur + ((wn * NFW + f) * 64 + lane) * 16);
              __builtin_amdgcn_sched_barrier(0);   // keep every fragment read ahead of the MFMA chain (counted lgkmcnt waits)
  #pragma unroll
              for (int f = 0; f < NFW; ++f)
  #pragma unroll
                for (int i = 0; i < 4; ++i) PR::mma(af[f], bf[i], acc[f][i]);
            }
          }
        }
      }
    }

  } else if constexpr (MODE == 5 || MODE == 6) {
    // ---- mode 0 for kw = 3 with the weights fetched a ROW of three taps ahead.  Mode 0 requests the fragments of step k + 1 during step k:
    // one step is 16 NFW MFMAs = 256 cycles for a wave, an L2 round trip under load several times that, so every step of a workgroup ends
    // up as long as the round trip (timing-only build without MFMAs, halo loads and stores: Conv3d_2c still takes 0.117 of its 0.258 ms:
    // 850 cycles per step and workgroup).  Here the three taps of a (dt, dh) row live in three named registers; a register is refilled --
    // from inline asm, hipcc drains compiler-issued queues at loop merge points -- with the same tap of the NEXT row right after the ring
    // write that consumed it, so a fragment has three steps to arrive and ONE counted s_waitcnt vmcnt(2) per step releases exactly the
    // register the step writes to LDS (the two younger refills stay in flight; halo loads the compiler issues in between only make the
    // wait longer).  Ring, barriers, fragment reads and K order as in mode 0: bitwise the same outputs.
    static_assert(WN == 1, "mode 5 shares the weights through LDS");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NWAIT = 2 * WCH;                      // loads younger than the ones a step consumes: two steps' worth
    const int wchunk = NF >= 4 ? tid : (tid & 127);
    // step k of this channel tile: scalar base wtile + k * wstep, this thread's chunk a 32-bit offset (SADDR form: no 64-bit VGPR pointers)
    const char* const wtile = wbase + (size_t)ntile * NF * 1024;
    const unsigned wvoff = (unsigned)(wchunk * 16);
    const unsigned wvoff2 = wvoff + (unsigned)w2off;    // second chunk of a step (WCH = 2; NF = 6: threads without one re-read their first)
    const int last_step = nsteps - 1;
    u32x4 q0 = {}, q1 = {}, q2 = {}, r0 = {}, r1 = {}, r2 = {};   // (r*: the second chunk of a step for channel tiles of more than 64, WCH = 2;
                                                                  //  every one is defined by its first wload before any use: the zeros are dead code)
    auto wload = [&](u32x4& dst, u32x4& dst2, int k) {
      if (!CAB(1)) return;
      const char* const ws = wtile + (size_t)(k < last_step ? k : last_step) * wstep;      // past the end: re-read the last step (never used)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(wvoff), "s"(ws) : "memory");
      if constexpr (WCH == 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst2) : "v"(wvoff2), "s"(ws) : "memory");
    };
    auto wwait = [&](u32x4& q, u32x4& r) {
      if constexpr (WCH == 2) asm volatile("s_waitcnt vmcnt(%2) ; release %0 %1" : "+v"(q), "+v"(r) : "i"(NWAIT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%1) ; release %0" : "+v"(q) : "i"(NWAIT) : "memory");
    };
    // MODE 6: the ring write moves BEHIND the barrier.  Mode 5 (and 0) write the fragments of step k, wait for the write (lgkmcnt(0) of
    // __syncthreads), meet at the barrier and then read them back: two LDS round trips in a row on every step's critical path, both behind
    // the other two workgroups' reads in the LDS queue.  Here step k, once through its barrier, requests its fragments and THEN writes step
    // k + 1's into the other slot (free: every wave passed barrier k with its reads of step k - 1 consumed); the write completes under the
    // step's MFMAs and the lgkmcnt(0) of the next barrier finds it done.  The registers hold steps k + 1 .. k + 3.
    constexpr bool WA = MODE == 6;
    int it_w = 0;
    if constexpr (WA) {
      wload(q0, r0, 0);
      if (!CAB(1)) {}
      else if constexpr (WCH == 2) asm volatile("s_waitcnt vmcnt(0) ; release %0 %1" : "+v"(q0), "+v"(r0) :: "memory");
      else asm volatile("s_waitcnt vmcnt(0) ; release %0" : "+v"(q0) :: "memory");
      *(u32x4*)(wbuf + wchunk * 16) = q0;
      if (WCH == 2 && w2ok) *(u32x4*)(wbuf + wchunk * 16 + 4096) = r0;
      wload(q0, r0, 1); wload(q1, r1, 2); wload(q2, r2, 3);
    } else {
      wload(q0, r0, 0); wload(q1, r1, 1); wload(q2, r2, 2);
    }
    auto step = [&](u32x4& q, u32x4& r, int tapoff) {
      char* const wcur = wbuf + (it_w & 1) * (NF * 1024);
      if constexpr (!WA) {
        if (CAB(1)) wwait(q, r);
        if (CAB(4)) {
          *(u32x4*)(wcur + wchunk * 16) = q;
          if (WCH == 2 && w2ok) *(u32x4*)(wcur + wchunk * 16 + 4096) = r;
        }
        ++it_w;
        wload(q, r, it_w + 2);                          // the same tap of the next row (it_w is already k + 1; clamped past the end)
      }
      if (CAB(2)) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag bf[4], af[NFW];
      if (wave_active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = CAB(16) ? *(const frag*)(halo + rowpos[i] + tapoff) : frag{};
#pragma unroll
        for (int f = 0; f < NFW; ++f) af[f] = CAB(32) ? *(const frag*)(wcur + ((wn * NFW + f) * 64 + lane) * 16) : frag{};
      }
      if constexpr (WA) {
        char* const wnext = wbuf + ((it_w + 1) & 1) * (NF * 1024);
        if (CAB(1)) wwait(q, r);
        if (CAB(4)) {
          *(u32x4*)(wnext + wchunk * 16) = q;
          if (WCH == 2 && w2ok) *(u32x4*)(wnext + wchunk * 16 + 4096) = r;
        }
        ++it_w;
        wload(q, r, it_w + 3);                          // (q held step k + 1 = it_w now; its next turn is step k + 4)
      }
      if (wave_active && CAB(8)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NFW; ++f)
#pragma unroll
          for (int i = 0; i < 4; ++i) PR::mma(af[f], bf[i], acc[f][i]);
      }
    };
    // (large halos only -- P > 256, one image staged behind a barrier per slab: the two-image pipeline's four prefetch registers on top of
    //  the three weight registers would cost the third wave per SIMD)
    for (int s = 0; s < nslab; ++s) {
      __syncthreads();  // every wave has finished reading the previous slab's halo
      stage_dma(s);       // (its vmcnt(0) drains the row-ahead weight queue as well: in order, and the halo pieces are the youngest)
      int tapoff_t = 0;
      for (int dt = 0; dt < p.kt; ++dt, tapoff_t += p.FP * 16) {
        int tapoff = tapoff_t;
        for (int dh = 0; dh < p.kh; ++dh, tapoff += p.Wh * 16) {
          step(q0, r0, tapoff);
          step(q1, r1, tapoff + 16);
          step(q2, r2, tapoff + 32);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail refills: drained before the registers may be reused
  } else if constexpr (MODE == 4) {
    // ---- folded stem (flk_conv_weights_create_s2d_stem, api.cpp): chunk c of a position holds ONE (qt,qh) parity and
    // tap index 3 of an axis exists for parity 0 only, so for dt == 3 / dh == 3 half (or three quarters) of the chunks
    // of a tap are structurally zero.  A K step (32 K values = 4 chunks, one per lane group q) is assembled from
    // non-zero chunks only: class 0 one tap x chunks q; class 1 (dt = 3) taps (dw, dw+1) x chunks {0,1}; class 2
    // (dh = 3) taps (dw, dw+1) x chunks {0,2}; class 3 tap dw + q x chunk 0.  49 steps instead of 64; weights as in
    // mode 0 (LDS ring, one-step register prefetch), the halo is staged once.
    static_assert(WN == 1, "mode 4 shares the weights through LDS");
    const int wchunk = NF >= 4 ? tid : (tid & 127);
    const char* wsrc = p.w + (size_t)ntile * NF * 1024 + wchunk * 16;
    uint4 wreg0 = *(const uint4*)wsrc, wreg1 = make_uint4(0, 0, 0, 0);
    if (WCH == 2) wreg1 = *(const uint4*)(wsrc + w2off);
    wsrc += wstep;
    {
      const char* src; int ld;
      const bool chvalid = slab_src(0, src, ld);
#pragma unroll
      for (int n0 = 0; n0 < NPK; n0 += HB) {
        if (n0 * 64 >= p.P) break;
        uint4 v[HB];
#pragma unroll
        for (int n = 0; n < HB; ++n) v[n] = ldhalo(src, ld, goff[n0 + n], chvalid);
#pragma unroll
        for (int n = 0; n < HB; ++n)
          if (goff[n0 + n] != -2) *(uint4*)(hdst + (n0 + n) * 1024) = v[n];
      }
    }
    // per-lane displacement of (chunk, tap) against the class-0 address rowpos[i] = position * 16 + plane_off(q)
    const int base_q = plane_off(q, p.plane_b);
    const int lo1 = plane_off(q & 1, p.plane_b) + (q >> 1) * 16 - base_q;
    const int lo2 = plane_off((q & 1) * 2, p.plane_b) + (q >> 1) * 16 - base_q;
    const int lo3 = plane_off(0, p.plane_b) + q * 16 - base_q;
    int it_w = 0;
    for (int dt = 0; dt < 4; ++dt)
      for (int dh = 0; dh < 4; ++dh) {
        const int cls = (dt == 3) + 2 * (dh == 3);
        const int lo = cls == 0 ? 0 : cls == 1 ? lo1 : cls == 2 ? lo2 : lo3;
        const int wstride = cls == 0 ? 1 : cls == 3 ? 4 : 2;
        int tapoff = (dt * p.FP + dh * p.Wh) * 16 + lo;
        for (int dw = 0; dw < 4; dw += wstride, tapoff += wstride * 16) {
          char* const wcur = wbuf + (it_w & 1) * (NF * 1024);
          *(uint4*)(wcur + wchunk * 16) = wreg0;
          if (WCH == 2 && w2ok) *(uint4*)(wcur + wchunk * 16 + 4096) = wreg1;
          ++it_w;
          if (it_w < nsteps) {
            wreg0 = *(const uint4*)wsrc;
            if (WCH == 2) wreg1 = *(const uint4*)(wsrc + w2off);
            wsrc += wstep;
          }
          __syncthreads();
          if (wave_active) {
            frag bf[4], af[NFW];
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *(const frag*)(halo + rowpos[i] + tapoff);
#pragma unroll
            for (int f = 0; f < NFW; ++f) af[f] = *(const frag*)(wcur + ((wn * NFW + f) * 64 + lane) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NFW; ++f)
#pragma unroll
              for (int i = 0; i < 4; ++i) PR::mma(af[f], bf[i], acc[f][i]);
          }
        }
      }
  } else if constexpr (MODE == 3) {
    // ---- 1x1x1 convolution, wide wave tiles: a GEMM whose K loop is only cin/32 steps long, each step a new slab from
    // HBM.  Activation slabs run D steps ahead and weight chunks 2 steps ahead in register queues, issued from inline
    // asm in a FIXED order (every step: W(k+2) then A(k+D); the prologue replays that order from step -D), so one
    // hand-counted s_waitcnt vmcnt(N) per step retires exactly what the step consumes.
    static_assert(WN == 1, "mode 3 shares the weights through LDS");
#ifdef FLK_M3_D
    constexpr int D = FLK_M3_D;
#else
    constexpr int D = NF >= 8 ? 2 : 4;
#endif
    constexpr int NWAIT = D == 2 ? WCH + 4 : 8 + WCH;      // loads younger than the youngest load step k consumes
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int wchunk = NF >= 4 ? tid : (tid & 127);
    const char* const wfirst = wbase + (size_t)ntile * NF * 1024 + wchunk * 16;
    const char* const wlast = wfirst + (size_t)(nsteps - 1) * wstep;
    u32x4 pre[D][4], wq[2][2];
    auto aload = [&](u32x4 (&dst)[4], int s) {
      const char* src; int ld;
      const bool chvalid = slab_src(s < nslab ? s : nslab - 1, src, ld);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const char* ptr = src + (size_t)((goff[n] >= 0 && chvalid) ? goff[n] : 0) * ld * sizeof(T);   // position 0 is always readable
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[n]) : "v"(ptr) : "memory");
      }
    };
    auto wload = [&](u32x4 (&dst)[2], int k) {
      const char* ws = wfirst + (size_t)k * wstep;
      ws = ws < wlast ? ws : wlast;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[0]) : "v"(ws) : "memory");
      if constexpr (WCH == 2) {
        const char* ws1 = ws + w2off;                                   // (beyond the 13-bit immediate offset)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[1]) : "v"(ws1) : "memory");
      }
    };
#pragma unroll
    for (int k = -D; k < 0; ++k) {                          // the steady-state issue order, replayed from step -D
      if (k + 2 >= 0) wload(wq[(k + 2) & 1], k + 2);
      aload(pre[(k + D) % D], k + D);
    }
    constexpr int U = D < 2 ? 2 : D;                        // unroll so that both queues are indexed statically
    for (int k0 = 0; k0 < nsteps; k0 += U) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int kk = k0 + j;
        if (kk >= nsteps) break;
        u32x4 (&a4)[4] = pre[j % D];
        u32x4 (&w2)[2] = wq[j & 1];
        if constexpr (WCH == 2)
          asm volatile("s_waitcnt vmcnt(%6) ; release %0 %1 %2 %3 %4 %5" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]), "+v"(w2[0]), "+v"(w2[1]) : "i"(NWAIT) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%5) ; release %0 %1 %2 %3 %4" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]), "+v"(w2[0]) : "i"(NWAIT) : "memory");
        const int hsel = (kk & 1) * halo_bytes;
        {
          const char* src; int ld;
          const bool chvalid = slab_src(kk, src, ld);
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (goff[n] == -2) continue;
            const u32x4 v = (goff[n] >= 0 && chvalid) ? a4[n] : u32x4{0u, 0u, 0u, 0u};
            *(u32x4*)(hdst + hsel + n * 1024) = v;
          }
        }
        char* const wcur = wbuf + (kk & 1) * (NF * 1024);
        *(u32x4*)(wcur + wchunk * 16) = w2[0];
        if (WCH == 2 && w2ok) *(u32x4*)(wcur + wchunk * 16 + 4096) = w2[1];
        // unconditional (clamped past the end): every queue register has ONE definition per step on every path, so the
        // compiler has no reason to copy a register whose load is still in flight (tools/audit_asm_loads.py checks this)
        wload(w2, kk + 2);
        aload(a4, kk + D);
        __syncthreads();
        if (wave_active) {
          frag bf[4], af[NFW];
#pragma unroll
          for (int i = 0; i < 4; ++i) bf[i] = *(const frag*)(halo + hsel + rowpos[i]);
#pragma unroll
          for (int f = 0; f < NFW; ++f) af[f] = *(const frag*)(wcur + ((wn * NFW + f) * 64 + lane) * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int f = 0; f < NFW; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) PR::mma(af[f], bf[i], acc[f][i]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ---- narrow wave tiles (WN >= 2) and small grids of narrow channel tiles: few MFMAs per K step, so one step is
    // far shorter than a memory round trip or a workgroup barrier.  Every wave reads ITS OWN weight fragments straight
    // from global memory into registers (the 4/WN waves that share them meet in the L1): no LDS hop and NO per-step barrier -- waves only meet when
    // the halo image changes (once per slab).  The K loop is flattened over (slab, tap) and unrolled by D; the
    // fragment loads run D steps ahead in a register queue indexed with compile-time constants only.  They are
    // issued from inline asm with hand-counted s_waitcnt vmcnt(N): hipcc's own bookkeeping falls back to a
    // near-complete drain at the loop's merge points, which puts one full memory latency into every K step.  Loads
    // the compiler issues itself (halo prefetch) only make the hand-written counts wait longer, never shorter.
    //   * K1 (one tap per slab): the activation slabs are prefetched D slabs ahead as well.
#ifdef FLK_DA_D
    constexpr int D = FLK_DA_D;
#else
    constexpr int D = (NFW <= 2 && !K1) ? 8 : 4;   // K1 also keeps D activation slabs (16 VGPRs each) in flight
#endif
    // fragment f of this wave at step k = w + (k*cout_frags + ntile*NF + wn*NFW + f) KiB + lane*16
    const char* const wfirst = wbase + (size_t)ntile * NFT * 1024 + (wn * NFW * 64 + lane) * 16;
    const char* const wlast = wfirst + (size_t)(nsteps - 1) * wstep;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // a native vector: inline asm takes it in registers
    u32x4 wq[D][NFW];
    auto wload = [&](u32x4 (&dst)[NFW], const char* ws) {
      ws = ws < wlast ? ws : wlast;                     // past the end: re-read the last step (never used)
      static_assert(NFW <= 4, "offset field");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[0]) : "v"(ws) : "memory");
      if constexpr (NFW > 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(dst[1]) : "v"(ws) : "memory");
      if constexpr (NFW > 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(dst[2]) : "v"(ws) : "memory");
      if constexpr (NFW > 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(dst[3]) : "v"(ws) : "memory");
    };
    // wait until at most N of the loads issued so far are outstanding, and make the queue entry depend on that wait
    // wait until at most N of the loads issued so far are outstanding, and make the queue entry depend on that wait
    auto wwait = [&](u32x4 (&q)[NFW]) {
      constexpr int N = (D - 1) * NFW;
      if constexpr (NFW == 2) asm volatile("s_waitcnt vmcnt(%2) ; release %0 %1" : "+v"(q[0]), "+v"(q[1]) : "i"(N) : "memory");
      else if constexpr (NFW == 4)
        asm volatile("s_waitcnt vmcnt(%4) ; release %0 %1 %2 %3" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "i"(N) : "memory");
      else asm volatile("s_waitcnt vmcnt(%1) ; release %0" : "+v"(q[0]) : "i"(N) : "memory");
    };
    static_assert(NFW == 1 || NFW == 2 || NFW == 4, "queue wait is written for 1, 2 or 4 fragments per wave");
#pragma unroll
    for (int j = 0; j < D; ++j) wload(wq[j], wfirst + (size_t)j * wstep);
    const char* wnext = wfirst + (size_t)D * wstep;

    uint4 pre[D][4];
    auto prefetch = [&](uint4 (&dst)[4], int s) {
      const char* src; int ld;
      const bool chvalid = slab_src(s < nslab ? s : nslab - 1, src, ld);
#pragma unroll
      for (int n = 0; n < 4; ++n) dst[n] = ldhalo(src, ld, goff[n], chvalid);
    };
    auto put_halo = [&](const uint4 (&v)[4], char* hd) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (goff[n] != -2) *(uint4*)(hd + n * 1024) = v[n];
    };
    if (K1) {
#pragma unroll
      for (int j = 0; j < D; ++j) prefetch(pre[j], j);
    } else if (small_halo) prefetch(pre[0], 0);

    int s = 0, dt = 0, dh = 0, dw = 0, tapoff_t = 0, tapoff_h = 0, tapoff = 0, hsel = 0;
    bool newslab = true;
    for (int k0 = 0; k0 < nsteps; k0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int kk = k0 + j;
        if (kk >= nsteps) break;
        // halo image: two images alternate for small halos.  Image i is rewritten only after a barrier that every
        // wave reaches after its last read of it (the barrier of the slab in between).
        if (K1) {
          hsel = (kk & 1) * halo_bytes;
          put_halo(pre[j], hdst + hsel);
          prefetch(pre[j], kk + D);
          __syncthreads();
        } else if (newslab) {
          newslab = false;
          if (small_halo) {
            hsel = (s & 1) * halo_bytes;
            put_halo(pre[0], hdst + hsel);
            prefetch(pre[0], s + 1);
            __syncthreads();
          } else {
            __syncthreads();  // every wave has finished reading the previous slab's halo
            stage_dma(s);     // (its vmcnt(0) also empties the weight queue: the counted waits behind it find their pieces landed)
            __syncthreads();
          }
        }
        frag bf[4];
        if (wave_active) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bf[i] = *(const frag*)(halo + hsel + rowpos[i] + tapoff);
        }
        wwait(wq[j]);
        if (wave_active) {
#pragma unroll
          for (int f = 0; f < NFW; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) PR::mma(__builtin_bit_cast(frag, wq[j][f]), bf[i], acc[f][i]);
        }
        wload(wq[j], wnext);          // refill this queue slot for step kk + D (clamped past the end: never used)
        wnext += wstep;
        if (!K1) {   // next tap (scalar state): w fastest, then h, then t, then the next slab
          if (++dw < p.kw) tapoff += 16;
          else {
            dw = 0;
            if (++dh < p.kh) tapoff_h += p.Wh * 16;
            else {
              dh = 0;
              if (++dt < p.kt) tapoff_t += p.FP * 16;
              else { dt = 0; tapoff_t = 0; ++s; newslab = true; }
              tapoff_h = tapoff_t;
            }
            tapoff = tapoff_h;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the queue's tail before the wave may end
  }

  // ---- epilogue: lane = position (m of fragment i); lane group q owns EPL channels of every 4*EPL-channel store group ----
  if (!wave_active) return;
  // store group g of this wave = global group wn*NG + g: channels ntile*16NF + (wn*NG + g)*4*EPL + q*EPL + [0, EPL)
  constexpr int NG = 4 * NFW / EPL;   // 16-byte channel groups per lane
  const int cbase = ntile * 16 * NFT + wn * NG * 4 * EPL + q * EPL;
  // (Position by position, store group by store group.  Measured and not kept, round 5: scale / bias once per lane + the add / mask operands
  //  of two positions in flight, as in the LDS-DMA kernels below -- Mixed_4 data-gradients 0 ... -3 %, but the late VideoResNet layers at
  //  batch 1 (8-70 workgroups, pure latency chains) +18 %, r2plus1d_18 bs 1 2.16 -> 2.30 ms: hipcc schedules this form's loads earlier.)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + i * 16 + m;
    if (r >= p.rows) continue;
    int rt, rh, rw;
    row_cell(p, r, rt, rh, rw);
    const int ot = ot0 + rt, oh = oh0 + rh, ow = ow0 + rw;
    if (ot >= p.To || oh >= p.Ho || ow >= p.Wo) continue;
    const size_t opos = ((size_t)(b * p.OT + ot * p.ost + p.oot) * p.OH + oh * p.osh + p.ooh) * p.OW + ow * p.osw + p.oow;
    const float* pb = pos_bias_row(p, b, ot, oh, ow);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c0 = cbase + g * 4 * EPL;
      if (c0 >= p.cout) continue;
      float v[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[e] = acc[(g * EPL + e) >> 2][i][(g * EPL + e) & 3];
      if (p.ksplit > 1) {                                // raw partial sums of this slice; conv_splitk_finish_kernel does the rest
        float* dst = p.part + ((size_t)by * p.npos + opos) * p.part_ld + c0;
#pragma unroll
        for (int e = 0; e < EPL; e += 4) *(float4*)(dst + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
        continue;
      }
      finish_store<T>(p, opos, pb, c0, v);
    }
  }
}

template <typename T, int NF, int WN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvKP p) {
  conv_igemm_body<T, NF, WN, MODE>(p, (int)blockIdx.x, (int)blockIdx.y);
}

// Several convolutions in ONE grid (an Inception block's Branch_1 and Branch_2 3x3x3 units, i3d.py:201-209, forward or data-gradient):
// blocks [start[i], start[i + 1]) work on member i exactly as a launch of their own would (same tiles, same K order per output: bitwise the
// results of separate launches).  The members share the template instance -- NFW channel fragments per wave, direct-A weights -- and
// differ in everything ConvKP holds, the wave layout included (p.wn).  One launch instead of two on two streams: no fork for the small
// member, whose latency-bound workgroups (a 27-step K loop for Branch_2) fill the tail of the large one instead of running beside it.
constexpr int FLK_MAX_GROUP = 3;
struct ConvGroupKP {
  ConvKP m[FLK_MAX_GROUP];
  int start[FLK_MAX_GROUP + 1];      // first block of member i; members past the last hold INT_MAX
};
// MODE 1: direct-A members of NFW fragments per wave and p.wn waves along N each.  MODE 0: ring members, all with the SAME channel tile
// of NFW fragments (one wave row: the ring is sized by the template) -- the large launches whose weights go through LDS.  MODE 5: ring
// members that ALL qualify for the weights a row of taps ahead (3-tap rows, large halos).
template <typename T, int NFW, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_group_kernel(const ConvGroupKP g) {
  const int bx = (int)blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < FLK_MAX_GROUP; ++k) i += bx >= g.start[k];
  conv_igemm_body<T, NFW, MODE == 1 ? 0 : 1, MODE>(g.m[i], bx - g.start[i], 0);
}

// second launch of a split-K convolution: sum the slices in slice order (fixed: bitwise reproducible), then the epilogue.
// One thread = one output position x EPL channels.
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvKP p) {
  constexpr int EPL = Prec<T>::EPL;
  const int ngrp = (p.cout + EPL - 1) / EPL;
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= p.npos * (unsigned)ngrp) return;
  const unsigned opos = gid / (unsigned)ngrp;
  const int c0 = (int)(gid - opos * (unsigned)ngrp) * EPL;
  float v[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) v[e] = 0.f;
  // slices four at a time: their loads are requested together, the sums stay in slice order (a run-time-bounded loop is load -> wait -> add)
  for (int k0 = 0; k0 < p.ksplit; k0 += 4) {
    float4 t[4][EPL / 4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u < p.ksplit ? k0 + u : k0;             // (a slice past the end re-reads slice k0 and is not added)
      const float* src = p.part + ((size_t)k * p.npos + opos) * p.part_ld + c0;
#pragma unroll
      for (int e = 0; e < EPL; e += 4) t[u][e / 4] = *(const float4*)(src + e);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k0 + u >= p.ksplit) break;
#pragma unroll
      for (int e = 0; e < EPL; e += 4) { v[e] += t[u][e / 4].x; v[e + 1] += t[u][e / 4].y; v[e + 2] += t[u][e / 4].z; v[e + 3] += t[u][e / 4].w; }
    }
  }
  const float* pb = nullptr;
  if (p.pos_bias) {
    const int ow = (int)(opos % (unsigned)p.OW), oh = (int)(opos / (unsigned)p.OW % (unsigned)p.OH), ot = (int)(opos / (unsigned)(p.OW * p.OH) % (unsigned)p.OT);
    pb = pos_bias_row(p, (int)(opos / (unsigned)(p.OW * p.OH * p.OT)), ot, oh, ow);
  }
  finish_store<T>(p, opos, pb, c0, v);
}

// ------------------------------------------------------------------------------------------------
// 1x1x1 convolution (a plain GEMM over the positions) with BOTH operands streamed into LDS by the DMA path (global_load_lds_dwordx4):
// R ring slots of [256 positions x 64 bytes | NF KiB of weight fragments], R - 1 K steps in flight per workgroup WITHOUT holding them
// in registers.  Mode 3 keeps its activation slabs in a register queue and NF = 8 leaves room for two of them: ~37 KB in flight per
// CU, which a 2 us HBM miss turns into ~4.7 TB/s for the whole chip -- the measured rate of the fused Inception GEMMs.  Here two
// resident workgroups keep 2 x (R - 1) x (16 + NF) KiB on their way.
//   * tile = 256 CONSECUTIVE positions of the flattened [B,T,H,W] grid (stride 1, logical == physical grid) x 16 NF channels;
//   * a DMA wave-instruction moves 64 lanes x 16 bytes to 1 KiB of contiguous LDS: lane l of piece j = (position 16 j + (l >> 2),
//     LDS slot l & 3 of that position's 64 bytes).  The slots are swizzled, slot = chunk ^ g[(position >> 2) & 3], g = {0,3,2,1},
//     by choosing which chunk a lane FETCHES: with it the 16 lanes a ds_read_b128 serves per cycle ({0-3,12-15,20-27}, ...) fall on
//     16 different 16-byte bank groups;
//   * every wave issues the same number of pieces per step (4 activation + ceil(NF / 4) weight pieces), so ONE counted
//     s_waitcnt vmcnt per step retires exactly the slot the step consumes; one barrier per step publishes it and frees the slot the
//     next issue overwrites; the last step waits for everything (nothing newer is in flight);
//   * invalid channel chunks (cin % 32 != 0) fetch chunk 0 of the same position instead -- finite values against zero weights.
// Epilogue: finish_store, as everywhere.
template <int NF, int R>
__global__ __launch_bounds__(256, 2) void conv1x1_dma_kernel(const ConvKP p) {
  typedef Prec<bf16_t> PR;
  typedef typename PR::frag frag;
  constexpr int EPL = 8;
  constexpr int NFA = (NF + 3) / 4;              // weight pieces per wave per step
  constexpr int NPW = 4 + NFA;                   // DMA instructions per wave per step
  constexpr int SLOT = 16384 + NF * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  int ptile, ntile;
  {
    const int per = 8 * p.ntile_n, id = blockIdx.x;
    const int grp = id / per, r = id - grp * per;
    ntile = r >> 3;
    ptile = grp * 8 + (r & 7);
  }
  const unsigned npos = p.npos;
  if ((unsigned)ptile * 256u >= npos) return;
  const unsigned pos0 = (unsigned)ptile * 256u;
  const unsigned lds0 = lds_addr32(smem);

  // ---- DMA plan of this lane ----
  const int gsw[4] = {0, 3, 2, 1};
  const int cs = (lane & 3) ^ gsw[(lane >> 4) & 3];        // the chunk this lane fetches into slot lane & 3 of position lane >> 2
  unsigned voff1[4], voff2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned pos = pos0 + (unsigned)(16 * (wave + 4 * k) + (lane >> 2));
    pos = pos < npos ? pos : npos - 1;                      // rows past the end: a valid address, never stored
    voff1[k] = (pos * (unsigned)p.in_ld + (unsigned)(p.in_coff + cs * EPL)) * 2u;
    voff2[k] = (pos * (unsigned)p.in2_ld + (unsigned)(p.in2_coff + cs * EPL)) * 2u;
  }
  unsigned wvoff[NFA];
  int wpiece[NFA];
#pragma unroll
  for (int k = 0; k < NFA; ++k) {
    const int f = wave + 4 * k < NF ? wave + 4 * k : wave;   // (NF = 6: waves 2, 3 move their first piece twice -- equal counts per wave)
    wpiece[k] = f;
    wvoff[k] = (unsigned)((ntile * NF + f) * 1024 + lane * 16);
  }
  const size_t wstep = (size_t)p.cout_frags * 1024;
  const int nslab = p.nslab;
  auto issue = [&](int step, int slot) {
    const int s = step < nslab ? step : nslab - 1;
    const bool seg2 = s >= p.nslab1;
    const int sl = seg2 ? s - p.nslab1 : s;
    const char* const base = (seg2 ? p.in2 : p.in) + (size_t)sl * 64;
    const int cinseg = seg2 ? p.cin - p.cin1 : p.cin1;
    const unsigned adj = sl * 32 + cs * EPL < cinseg ? 0u : (unsigned)(-(cs * 16));
    const unsigned sb = lds0 + (unsigned)(slot * SLOT);
#pragma unroll
    for (int k = 0; k < 4; ++k) glds16((seg2 ? voff2[k] : voff1[k]) + adj, base, sb + (unsigned)((wave + 4 * k) * 1024));
    const char* const wb = p.w + (size_t)s * wstep;
#pragma unroll
    for (int k = 0; k < NFA; ++k) glds16(wvoff[k], wb, sb + 16384u + (unsigned)(wpiece[k] * 1024));
  };

  // ---- compute plan: wave w owns tile rows [64 w, 64 w + 64) ----
  int boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) boff[i] = (64 * wave + 16 * i + m) * 64 + ((q ^ gsw[(m >> 2) & 3]) * 16);
  f32x4 acc[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  static_assert(R == 3, "the tail logic below is written for one step of look-ahead beyond the one being waited for");
  int islot = 0;                                  // slot of the next issue
#pragma unroll
  for (int k = 0; k < R - 1; ++k) { issue(k, islot); islot = islot + 1 == R ? 0 : islot + 1; }      // (nslab >= 2: the host checks)
  int cslot = 0;                                  // slot of the step being consumed
  for (int k = 0; k < nslab; ++k) {
    // this wave's pieces of step k have landed: all but the NPW of step k + 1 -- in the last step there is nothing newer.  (Until the
    // third session of round 3 the tail re-read the last slab into dead slots to keep the counts equal, 2 of nslab + 2 fetches wasted:
    // 256 -> 128 channels at 200 704 positions 35.9 -> 35.0 us, 256 -> 384 74.4 -> 71.3 us, 64 -> 128 24.0 -> 22.3 us.)
    if (k + 1 < nslab) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                              // everybody's have; and everybody is done reading the slot issued into next
    if (k + R - 1 < nslab) issue(k + R - 1, islot);
    islot = islot + 1 == R ? 0 : islot + 1;
    const char* const sb = smem + cslot * SLOT;
    cslot = cslot + 1 == R ? 0 : cslot + 1;
    frag bf[4], af[NF];
#pragma unroll
    for (int i = 0; i < 4; ++i) bf[i] = *(const frag*)(sb + boff[i]);
#pragma unroll
    for (int f = 0; f < NF; ++f) af[f] = *(const frag*)(sb + 16384 + (f * 64 + lane) * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) PR::mma(af[f], bf[i], acc[f][i]);
  }

  // ---- epilogue ----
  constexpr int NG = 4 * NF / EPL;
  const int cbase = ntile * 16 * NF + q * EPL;
  constexpr bool PRE = NG <= 3;                    // (NG = 4: 64 more registers would spill)
  float4 sc[PRE ? NG : 1][2], bi[PRE ? NG : 1][2];
  if (PRE) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c = cbase + g * 4 * EPL, cc = c < p.cout ? c : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        sc[g][h] = p.scale ? *(const float4*)(p.scale + cc + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);
        bi[g][h] = p.bias ? *(const float4*)(p.bias + cc + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if constexpr (PRE) {
    // the add / mask operands of two positions in flight (conv_common.h, epi_fetch): two memory round trips per tile, not four
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += 2) {
      unsigned pos[2];
      uint4 av[2][NG], mv[2][NG];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pos[u] = pos0 + (unsigned)(64 * wave + 16 * (i0 + u) + m);
        epi_fetch<bf16_t, NG>(p, (size_t)(pos[u] < npos ? pos[u] : 0u), cbase, av[u], mv[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (pos[u] >= npos) continue;
        float v[NG][EPL];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int e = 0; e < EPL; ++e) v[g][e] = acc[(g * EPL + e) >> 2][i0 + u][(g * EPL + e) & 3];
        finish_store_row_ops<bf16_t, NG>(p, (size_t)pos[u], nullptr, cbase, v, sc, bi, av[u], mv[u]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned pos = pos0 + (unsigned)(64 * wave + 16 * i + m);
      if (pos >= npos) continue;
      float v[NG][EPL];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[g][e] = acc[(g * EPL + e) >> 2][i][(g * EPL + e) & 3];
      finish_store_row<bf16_t, NG>(p, (size_t)pos, nullptr, cbase, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The temporal half of a (2+1)D unit -- a (3,1,1) convolution, stride 1, pad 1 (torchvision Conv2Plus1D, reference call site
// model.py:421; forward and data-gradient) -- through an LDS-DMA ring like conv1x1_dma_kernel's.  The halo kernels give these layers 3 K
// steps per staged slab (2.0-2.2 TB/s of compulsory bytes at layer1).  Here a workgroup owns ALL T frames of G groups of 16 consecutive
// spatial positions (T * G = 16 fragments of 16 positions = 256 rows; T = 16 / 8 / 4 / 2), a ring slot holds one 32-channel slab of them
// -- 16 one-KiB DMA pieces, one per (group, frame): 16 positions x 64 bytes, contiguous in memory but for the channel stride -- plus the
// slab's three taps of weights; tap dt of output frame t reads piece t + dt - 1 of its group, the pieces for t = -1 and t = T are zeros
// written once per slot and never fetched: no halo bytes at all, 48 NF / 4 MFMAs per wave between two barriers, R - 1 slabs in flight.
template <int NF, int R>
__global__ __launch_bounds__(256, 1) void conv_t3_dma_kernel(const ConvKP p) {
  typedef Prec<bf16_t> PR;
  typedef typename PR::frag frag;
  constexpr int EPL = 8;
  constexpr int NWP = 3 * NF / 4;                // weight pieces per wave per slab (three taps)
  constexpr int NPW = 4 + NWP;                   // DMA instructions per wave per slab
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  const int T = p.Ti, G = 16 / T, HW = p.Hi * p.Wi, nch = HW >> 4;     // nch 16-position chunks per frame
  const int WOFF = (16 + 2 * G) * 1024, SLOT = WOFF + 3 * NF * 1024;
  int ptile, ntile;
  {
    const int per = 8 * p.ntile_n, id = blockIdx.x;
    const int grp = id / per, r = id - grp * per;
    ntile = r >> 3;
    ptile = grp * 8 + (r & 7);
  }
  const int tpc = (nch + G - 1) / G;             // tiles per clip
  if (ptile >= p.B * tpc) return;
  const int b = ptile / tpc, sp = ptile - b * tpc;
  const unsigned lds0 = lds_addr32(smem);

  // zero pieces (frames -1 and T of every group) of every slot: written once, never overwritten by a DMA
  for (int i = tid; i < R * 2 * G * 64; i += 256) {
    const int slot = i / (2 * G * 64), r = i - slot * (2 * G * 64), g = r / 128, e = r - g * 128;
    const int piece = g * (T + 2) + (e >> 6) * (T + 1);
    *(uint4*)(smem + slot * SLOT + piece * 1024 + (e & 63) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }

  // ---- this wave's four fragments f = 4 wave + i = (group g, frame t); a group beyond the frame's last chunk is clamped (never stored) ----
  const int gsw[4] = {0, 3, 2, 1};
  const int cs = (lane & 3) ^ gsw[(lane >> 4) & 3];        // the chunk this lane fetches into slot lane & 3 of position lane >> 2
  unsigned voff[4];
  int lpiece[4];
  bool fvalid[4];
  unsigned fpos[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = 4 * wave + i, g = f / T, t = f - g * T;
    int c16 = sp * G + g;
    fvalid[i] = c16 < nch;
    c16 = c16 < nch ? c16 : nch - 1;
    const unsigned base = (unsigned)((b * T + t) * HW + c16 * 16);
    voff[i] = ((base + (unsigned)(lane >> 2)) * (unsigned)p.in_ld + (unsigned)(p.in_coff + cs * EPL)) * 2u;
    fpos[i] = base + (unsigned)m;
    lpiece[i] = __builtin_amdgcn_readfirstlane(g * (T + 2) + t + 1);      // (wave-uniform: it goes into m0)
  }
  const size_t wstep = (size_t)p.cout_frags * 1024;
  const int nslab = p.nslab;
  unsigned wvoff[NWP];                            // byte offset of this lane's 16 bytes of weight piece j = wave + 4 k inside a slab's three taps
#pragma unroll
  for (int k = 0; k < NWP; ++k) {
    const int j = wave + 4 * k, dt = j / NF, f = j - dt * NF;
    wvoff[k] = (unsigned)dt * (unsigned)wstep + (unsigned)((ntile * NF + f) * 1024 + lane * 16);
  }
  auto issue = [&](int sl, int slot) {
    const char* const base = p.in + (size_t)sl * 64;
    const unsigned adj = sl * 32 + cs * EPL < p.cin ? 0u : (unsigned)(-(cs * 16));
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(slot * SLOT)));
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(voff[i] + adj, base, sb + (unsigned)(lpiece[i] * 1024));
    // weights of (slab sl, tap dt), fragment f of this channel tile: piece j = wave + 4 k of the 3 NF pieces (dt = j / NF, f = j % NF)
    const char* const wb = p.w + (size_t)sl * 3 * wstep;
#pragma unroll
    for (int k = 0; k < NWP; ++k) glds16(wvoff[k], wb, sb + (unsigned)(WOFF + (wave + 4 * k) * 1024));
  };

  f32x4 acc[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // B fragment of (fragment i, tap dt): piece lpiece[i] + dt - 1, row m, chunk q (swizzled like the DMA image)
  int boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) boff[i] = (lpiece[i] - 1) * 1024 + m * 64 + ((q ^ gsw[(m >> 2) & 3]) * 16);

  int islot = 0;
#pragma unroll
  for (int k = 0; k < R - 1; ++k) {
    if (k < nslab) issue(k, islot);
    islot = islot + 1 == R ? 0 : islot + 1;
  }
  int cslot = 0;
  for (int k = 0; k < nslab; ++k) {
    // this wave's pieces of slab k have landed once at most the pieces of the min(R - 2, nslab - 1 - k) younger slabs are outstanding
    const int younger = nslab - 1 - k < R - 2 ? nslab - 1 - k : R - 2;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPW) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                              // everybody's pieces (and, first time, the zero pieces) are visible; the slot issued into next is free
    if (k + R - 1 < nslab) issue(k + R - 1, islot);
    islot = islot + 1 == R ? 0 : islot + 1;
    const char* const sb = smem + cslot * SLOT;
    cslot = cslot + 1 == R ? 0 : cslot + 1;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      frag bf[4], af[NF];
#pragma unroll
      for (int i = 0; i < 4; ++i) bf[i] = *(const frag*)(sb + boff[i] + dt * 1024);
#pragma unroll
      for (int f = 0; f < NF; ++f) af[f] = *(const frag*)(sb + WOFF + ((dt * NF + f) * 64 + lane) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) PR::mma(af[f], bf[i], acc[f][i]);
    }
  }

  // ---- epilogue (conv1x1_dma_kernel's) ----
  constexpr int NG = 4 * NF / EPL;
  const int cbase = ntile * 16 * NF + q * EPL;
  constexpr bool PRE = NG <= 3;
  float4 sc[PRE ? NG : 1][2], bi[PRE ? NG : 1][2];
  if (PRE) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c = cbase + g * 4 * EPL, cc = c < p.cout ? c : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        sc[g][h] = p.scale ? *(const float4*)(p.scale + cc + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);
        bi[g][h] = p.bias ? *(const float4*)(p.bias + cc + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if constexpr (PRE) {
    // the add / mask operands of two fragments in flight (conv_common.h, epi_fetch)
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += 2) {
      uint4 av[2][NG], mv[2][NG];
#pragma unroll
      for (int u = 0; u < 2; ++u) epi_fetch<bf16_t, NG>(p, (size_t)fpos[i0 + u], cbase, av[u], mv[u]);      // (a clamped group: valid memory, not stored)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!fvalid[i0 + u]) continue;
        float v[NG][EPL];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int e = 0; e < EPL; ++e) v[g][e] = acc[(g * EPL + e) >> 2][i0 + u][(g * EPL + e) & 3];
        finish_store_row_ops<bf16_t, NG>(p, (size_t)fpos[i0 + u], nullptr, cbase, v, sc, bi, av[u], mv[u]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!fvalid[i]) continue;
      float v[NG][EPL];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[g][e] = acc[(g * EPL + e) >> 2][i][(g * EPL + e) & 3];
      finish_store_row<bf16_t, NG>(p, (size_t)fpos[i], nullptr, cbase, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
flk_tile flk_choose_tile(int To, int Ho, int Wo, int kt, int kh, int kw, int st, int sh, int sw, int max_rows, int max_halo) {
  if (max_halo <= 0) max_halo = FLK_MAX_HALO;      // (conv_pc.hip asks for 512-row tiles; conv_igemm_kernel's callers for <= FLK_ROWS / FLK_MAX_HALO)
  if (max_rows <= 0) max_rows = FLK_ROWS;
  flk_tile best{1, 1, 1};
  double best_eff = -1.0;
  long best_halo = 0;
  for (int Tt = 1; Tt <= To && Tt <= max_rows; ++Tt)
    for (int Ht = 1; Ht <= Ho && Tt * Ht <= max_rows; ++Ht) {
      const int wmax = max_rows / (Tt * Ht) < Wo ? max_rows / (Tt * Ht) : Wo;
      for (int Wt = 1; Wt <= wmax; ++Wt) {
        const long halo = (long)((Tt - 1) * st + kt) * ((Ht - 1) * sh + kh) * ((Wt - 1) * sw + kw);
        if (halo > max_halo) continue;
        const long tiles = (long)((To + Tt - 1) / Tt) * ((Ho + Ht - 1) / Ht) * ((Wo + Wt - 1) / Wt);
        // a workgroup occupies a CU slot whatever its row count: utilisation = useful rows / (tiles * 256);
        // staging cost grows with the halo (per slab) while MFMA work grows with rows * taps
        const int rows = Tt * Ht * Wt;
        const double eff = (double)To * Ho * Wo / ((double)tiles * max_rows);
        const double score = eff / (1.0 + 2.0 * (double)halo / ((double)max_rows * kt * kh * kw));
        (void)rows;
        if (score > best_eff + 1e-9 || (score > best_eff - 1e-9 && (Wt > best.Wt || (Wt == best.Wt && halo < best_halo)))) {
          best_eff = score; best = flk_tile{Tt, Ht, Wt}; best_halo = halo;
        }
      }
    }
  return best;
}

template <typename T, int NF, int WN, int MODE>
static int launch(const ConvKP& kp, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set[FLK_MAX_DEVICES] = {};      // per device: one process may drive several GPUs
  if (int rc = flk_raise_lds_limit((const void*)conv_igemm_kernel<T, NF, WN, MODE>, 96 * 1024, attr_set)) return rc;
  FLK_LAUNCH_KERNEL((conv_igemm_kernel<T, NF, WN, MODE>), grid, dim3(256), lds, s, kp);
  flk_last_kernel_tag = "conv_igemm_kernel";
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// Bank-conflict-free fragment reads.  A wave reads a position fragment with ds_read_b128: lane (q, m) fetches the 16-byte slot of tile
// row 16 i + m in plane q, and the LDS serves the 16 lanes of one q together (16 x 16 bytes = all 64 banks) -- in one pass iff their
// slots differ mod 16, in two when any two collide.  With rows enumerated along w, then h, a fragment of an 8-wide tile is two runs of 8
// slots one halo row (Wh = 10) apart: slots 0..7 and 10..17 -> 0 / 16 and 1 / 17 collide, every such read takes twice the LDS cycles
// (rocprofv3 on Conv3d_2c: SQ_LDS_BANK_CONFLICT = 30 % of SQ_LDS_IDX_ACTIVE, 4.7 cycles per position-fragment read).  Enumerated along w,
// then T, the runs are one FRAME apart, and the frame pitch FP can be padded to any residue at <= 15 slots per frame: FP = Wt (mod 16)
// makes the runs of a fragment consecutive mod 16.  Chosen per launch: the (enumeration, pad) with the fewest extra passes over the tile's
// fragments, the unpadded w-h-T form on ties, and never at the price of a resident workgroup.  Same products, same K order per output:
// the results do not change.
static void pick_halo_layout(ConvKP& kp, size_t ring_bytes, int max_resident) {
  if (kp.P <= 256) return;                     // (the two-image path of small halos keeps its 256-slot images)
  const int cells = kp.Hh * kp.Wh;
  auto extra_passes = [&](int tfast, int FP) { return conv_halo_extra_passes(kp, tfast, FP); };
  auto lds_of = [&](int P) { return 4 * ((size_t)(P + 63) / 64 * 1024) + 64 + ring_bytes; };      // (images of whole 64-cell DMA blocks)
  const size_t cap = 160 * 1024;
  const size_t res0 = std::min<size_t>(max_resident, cap / lds_of(kp.P));
  int best = extra_passes(0, cells), best_t = 0, best_fp = cells;
  for (int pad = 0; pad < 16 && best > 0; ++pad)
    for (int tfast = 0; tfast < 2 && best > 0; ++tfast) {
      const int FP = cells + pad, P = (kp.Th - 1) * FP + cells;
      if (P > FLK_MAX_HALO || std::min<size_t>(max_resident, cap / lds_of(P)) < res0) continue;
      const int c = extra_passes(tfast, FP);
      if (c < best) { best = c; best_t = tfast; best_fp = FP; }
    }
  kp.tfast = best_t; kp.FP = best_fp; kp.P = (kp.Th - 1) * best_fp + cells;
}

constexpr int FLK_MAX_KSPLIT = 8;
// FLK_CONV_PC=0: the large 3x3x3 layers stay on conv_igemm_kernel (A/B of the producer / consumer kernel, conv_pc.hip; read once); 1: single
// launches only (Conv3d_2c_3x3), 2 (default): the Mixed_3* ring groups as well
static int pc_route_on() { static const int on = getenv("FLK_CONV_PC") ? atoi(getenv("FLK_CONV_PC")) : 2; return on; }

static int launch_any(const ConvKP& kp, dim3 grid, size_t lds, hipStream_t s, int dtype, int nf, int wn, int mode);
static bool dbg_on() { static const bool d = getenv("FLK_CONV_DBG") != nullptr; return d; }      // print every launch's layout
// FLK_CONV_KSPLIT=k: every split-K-eligible launch in k slices (tests: the split path on small shapes); 0 = the heuristic
static int ksplit_forced() { static const int k = getenv("FLK_CONV_KSPLIT") ? atoi(getenv("FLK_CONV_KSPLIT")) : 0; return k; }

static bool splitk_eligible(const flk_conv_args* a, const flk_conv_weights* w) {
  return !w->stem4 && w->nslab >= 2 && !a->pos_bias && a->ost == 1 && a->osh == 1 && a->osw == 1 && a->oot == 0 && a->ooh == 0 && a->oow == 0 &&
         a->To == a->OT && a->Ho == a->OH && a->Wo == a->OW;
}
// Slices for this convolution (1 = no split).  Split-K pays where even 256-row tiles leave three quarters of the CUs without a
// workgroup AND the K loop is long (3x3x3 over >= 4 slabs: Mixed_5*, the late VideoResNet layers): measured on the Mixed_5c
// 3x3x3 pair 49 -> 37 us forward, 88 -> 35 us data-gradient.  1x1x1 GEMMs (K loop of <= 26 steps) and the 25088-position
// Mixed_4* layers lose (the fp32 partial sums cost more than the idle CUs): never split.
static int plan_ksplit(const flk_conv_args* a, const flk_conv_weights* w) {
  if (!splitk_eligible(a, w) || w->ntaps == 1 || w->nslab < 4) return 1;
  const int nf = w->nf;
  const flk_tile t = flk_choose_tile(a->To, a->Ho, a->Wo, a->kt, a->kh, a->kw, a->st, a->sh, a->sw, nf == 2 ? 192 : FLK_ROWS,
                                     nf == 2 ? 768 : FLK_MAX_HALO);
  const long wgs = (long)a->B * ((a->To + t.Tt - 1) / t.Tt) * ((a->Ho + t.Ht - 1) / t.Ht) * ((a->Wo + t.Wt - 1) / t.Wt) * (w->cout_frags / nf);
  // up to 64 workgroups: always; 65 .. 128 (two slices): only K loops of >= 100 steps.  The second rule is round 4's -- the 90-frame clips'
  // Mixed_5* data-gradients sit at 72 workgroups x 270-324 steps and ran unsplit (0.077-0.092 ms against 0.032-0.040 for the 64-frame
  // clips' 48 workgroups), as do Mixed_5*'s forward Branch_1 at 65 and r2plus1d_18's layer3 / layer4 halves at bs 8.  Measured, three
  // rounds on one box (gpurun_out/sk_ab.log, sk_ab3.log): I3D T = 90 bs 8 7.75 -> 7.64 ms, r2plus1d_18 bs 8 4.33 -> 4.29, I3D T = 64 bs 8
  // 5.596 -> 5.584, the batch-1 plans unchanged (without the step bound they lose 0.01-0.02 ms to the extra finish kernels).
  constexpr long maxwg = 128, minsteps = 100;
  if (wgs > maxwg || (wgs > 64 && (long)w->nslab * w->ntaps < minsteps)) return 1;
  int ks = (int)(256 / wgs);
  if (ks < 2) ks = 2;
  ks = ks > FLK_MAX_KSPLIT ? FLK_MAX_KSPLIT : ks;
  return ks > w->nslab ? w->nslab : ks;
}

extern "C" int64_t flk_conv_splitk_bytes(const flk_conv_args* a, const flk_conv_weights* w) {
  if (!a || !w || !splitk_eligible(a, w)) return 0;
  const int ks = ksplit_forced() ? FLK_MAX_KSPLIT : plan_ksplit(a, w);
  return ks > 1 ? (int64_t)ks * a->B * a->OT * a->OH * a->OW * a->cout * (int64_t)sizeof(float) : 0;
}

// what conv3d_impl decided for a launch (plan-only calls: the members of a grouped launch)
struct ConvPlan { ConvKP kp; dim3 grid; size_t lds; int nf, wn, mode; };

// force_wn: 0 = heuristic, else 1 / 2 / 4.  force_da: -1 = heuristic, 0 = LDS weight ring, 1 = direct A.  force_ks: 0 = heuristic.
// plan != nullptr: validate and plan only -- nothing is launched, the decisions land in *plan (no split-K, no LDS-DMA GEMM route).
static int conv3d_impl(const flk_conv_args* a, const flk_conv_weights* w, int dtype, void* stream, int force_wn, int force_da, int force_ks = 0,
                       ConvPlan* plan = nullptr) {
  FLK_REQUIRE(a && w && w->dev, "flk_conv3d: null argument");
  FLK_REQUIRE(dtype == w->dtype, "flk_conv3d: dtype %d != packed weight dtype %d", dtype, w->dtype);
  FLK_REQUIRE(a->kt == w->kt && a->kh == w->kh && a->kw == w->kw && a->cin == w->cin && a->cout == w->cout,
              "flk_conv3d: args (%dx%dx%d, %d->%d) do not match weights (%dx%dx%d, %d->%d)", a->kt, a->kh, a->kw,
              a->cin, a->cout, w->kt, w->kh, w->kw, w->cin, w->cout);
  FLK_REQUIRE(a->cin % 8 == 0 && a->cout % 8 == 0 && a->in_ld % 8 == 0 && a->in_coff % 8 == 0 &&
                  a->out_ld % 8 == 0 && a->out_coff % 8 == 0,
              "flk_conv3d: channel counts / strides / offsets must be multiples of 8");
  FLK_REQUIRE(!a->add || (a->add_ld % 8 == 0 && a->add_coff % 8 == 0), "flk_conv3d: add ld/coff % 8");
  FLK_REQUIRE(!a->mask || (a->mask_ld % 8 == 0 && a->mask_coff % 8 == 0), "flk_conv3d: mask ld/coff % 8");
  FLK_REQUIRE((a->in2 || a->in_coff + a->cin <= a->in_ld) && (a->out2 || a->out_coff + a->cout <= a->out_ld), "flk_conv3d: slice exceeds ld");
  FLK_REQUIRE(a->B > 0 && a->To > 0 && a->Ho > 0 && a->Wo > 0 && a->st > 0 && a->sh > 0 && a->sw > 0 &&
                  a->ost > 0 && a->osh > 0 && a->osw > 0, "flk_conv3d: bad dims");
  FLK_REQUIRE((a->To - 1) * a->ost + a->oot < a->OT && (a->Ho - 1) * a->osh + a->ooh < a->OH &&
                  (a->Wo - 1) * a->osw + a->oow < a->OW && a->oot >= 0 && a->ooh >= 0 && a->oow >= 0,
              "flk_conv3d: logical output grid exceeds the physical output");
  FLK_REQUIRE(!w->stem4 || (a->st == 1 && a->sh == 1 && a->sw == 1 && !a->in2 && w->nslab == 1),
              "flk_conv3d: folded-stem weights need a stride-1, single-segment 4x4x4x32 convolution");
  const size_t esz = flk_esize(dtype);
  FLK_REQUIRE((size_t)a->B * a->Ti * a->Hi * a->Wi * a->in_ld < (1ull << 31) &&
                  (size_t)a->B * a->OT * a->OH * a->OW * a->out_ld < (1ull << 31),
              "flk_conv3d: tensor too large for 32-bit element offsets");
  (void)esz;

  ConvKP kp{};
  kp.in = (const char*)a->in; kp.w = (const char*)w->dev; kp.out = (char*)a->out;
  kp.scale = a->scale; kp.bias = a->bias; kp.add = (const char*)a->add; kp.mask = (const char*)a->mask;
  kp.pos_bias = a->pos_bias; kp.pos_bias_bstride = (long)a->pos_bias_bstride;
  FLK_REQUIRE(!a->pos_bias || (a->Ho >= 4 && a->Wo >= 4 && !a->out2), "flk_conv3d: pos_bias needs Ho, Wo >= 4 and a single output segment");
  kp.in_ld = a->in_ld; kp.in_coff = a->in_coff; kp.cin = a->cin;
  kp.B = a->B; kp.Ti = a->Ti; kp.Hi = a->Hi; kp.Wi = a->Wi;
  kp.kt = a->kt; kp.kh = a->kh; kp.kw = a->kw; kp.st = a->st; kp.sh = a->sh; kp.sw = a->sw;
  kp.pt = a->pt; kp.ph = a->ph; kp.pw = a->pw;
  kp.To = a->To; kp.Ho = a->Ho; kp.Wo = a->Wo;
  kp.out_ld = a->out_ld; kp.out_coff = a->out_coff; kp.cout = a->cout;
  kp.OT = a->OT; kp.OH = a->OH; kp.OW = a->OW;
  kp.ost = a->ost; kp.osh = a->osh; kp.osw = a->osw; kp.oot = a->oot; kp.ooh = a->ooh; kp.oow = a->oow;
  kp.add_ld = a->add_ld; kp.add_coff = a->add_coff; kp.mask_ld = a->mask_ld; kp.mask_coff = a->mask_coff;
  kp.relu = a->relu;
  // wave layout: start from 256-row tiles; while the launch has fewer workgroups than CUs, halve
  // the rows per workgroup (waves then split the channel tile instead).  bf16 needs >= 2 fragments per wave.
  const int nf = w->nf;
  const int ntile_n = w->cout_frags / nf;
  int wn = 1;
  // split-K decision first: a split launch keeps 256-row tiles (wn = 1) and gets its parallelism from the K slices
  int ksplit = 1;
  if (a->splitk_ws && splitk_eligible(a, w) && !plan) {
    ksplit = ksplit_forced() ? ksplit_forced() : force_ks > 0 ? force_ks : plan_ksplit(a, w);
    ksplit = ksplit > w->nslab ? w->nslab : ksplit > FLK_MAX_KSPLIT ? FLK_MAX_KSPLIT : ksplit < 1 ? 1 : ksplit;
  }
  // narrow channel tiles (nf = 2) are latency-bound: keep their halo <= 768 so that 3 workgroups fit a CU's LDS
  // the folded stem (mode 4): 192-row tiles (three computing waves, the fourth only stages) with a halo <= 704 keep the
  // workgroup at 53 KiB of LDS, i.e. THREE per CU instead of two: measured 0.62 vs 0.655 ms (4x6x8 vs 4x8x8 tiles)
  const int max_halo = w->stem4 ? 704 : nf == 2 ? 768 : FLK_MAX_HALO;
  // narrow channel tiles (nf = 2) likewise run faster on 192-row tiles (measured -4...-8 % on every nf = 2 layer; nf = 4 / 8
  // layers lose 10-15 % with them)
  const int max_rows = (w->stem4 || nf == 2) ? 192 : FLK_ROWS;
  flk_tile t = flk_choose_tile(a->To, a->Ho, a->Wo, a->kt, a->kh, a->kw, a->st, a->sh, a->sw, max_rows, max_halo);
  {
    const int wn_max = nf == 6 ? 1 : dtype == FLK_BF16 ? nf / 2 : nf;      // NFW >= 2 (bf16) / 1 (fp32); 96-channel tiles: ring kernels, wn = 1 only
    const int force = force_wn > 0 ? force_wn : 0;
    while (true) {
      const long wgs = (long)a->B * ((a->To + t.Tt - 1) / t.Tt) * ((a->Ho + t.Ht - 1) / t.Ht) * ((a->Wo + t.Wt - 1) / t.Wt) * ntile_n;
      // (threshold re-measured at the round-4 kernels: 64 / 128 / 256 / 384 / 512 workgroups -> 5.715 / 5.628 / 5.586 / 5.619 / 5.607 ms per step)
      const bool more = w->stem4 ? false : force ? wn < force : (wgs < 256 && ksplit == 1);   // fewer workgroups than CUs (mode 4 is written for wn = 1)
      if (!more || wn * 2 > 4 || wn * 2 > wn_max) break;
      wn *= 2;
      t = flk_choose_tile(a->To, a->Ho, a->Wo, a->kt, a->kh, a->kw, a->st, a->sh, a->sw, FLK_ROWS / wn, max_halo);
    }
  }
  kp.Tt = t.Tt; kp.Ht = t.Ht; kp.Wt = t.Wt; kp.rows = t.Tt * t.Ht * t.Wt;
  kp.nTt = (a->To + t.Tt - 1) / t.Tt; kp.nTh = (a->Ho + t.Ht - 1) / t.Ht; kp.nTw = (a->Wo + t.Wt - 1) / t.Wt;
  kp.Th = (t.Tt - 1) * a->st + a->kt; kp.Hh = (t.Ht - 1) * a->sh + a->kh; kp.Wh = (t.Wt - 1) * a->sw + a->kw;
  kp.P = kp.Th * kp.Hh * kp.Wh;
  FLK_REQUIRE(kp.P <= FLK_MAX_HALO && kp.rows <= FLK_ROWS / wn, "flk_conv3d: no tile fits (halo %d)", kp.P);
  kp.FP = kp.Hh * kp.Wh; kp.tfast = 0;
  kp.plane_b = (kp.P * 16 + 255) / 256 * 256;          // (pick_halo_layout may pad the image once the weight path is known)
  kp.nslab = w->nslab; kp.ntaps = w->ntaps; kp.cout_frags = w->cout_frags;
  if (a->in2) {
    FLK_REQUIRE(w->cin_split == a->cin1 && a->cin1 > 0 && a->cin1 < a->cin, "flk_conv3d: in2 given but weights were packed with "
                "cin_split %d (args cin1 %d)", w->cin_split, a->cin1);
    FLK_REQUIRE(a->in2_ld % 8 == 0 && a->in2_coff % 8 == 0 && a->in2_coff + a->cin - a->cin1 <= a->in2_ld &&
                    a->in_coff + a->cin1 <= a->in_ld, "flk_conv3d: bad in2 slice");
    kp.in2 = (const char*)a->in2; kp.in2_ld = a->in2_ld; kp.in2_coff = a->in2_coff; kp.cin1 = a->cin1; kp.nslab1 = w->nslab1;
  } else {
    FLK_REQUIRE(w->cin_split == 0, "flk_conv3d: weights packed for two input segments but in2 is NULL");
    kp.in2 = kp.in; kp.in2_ld = a->in_ld; kp.in2_coff = a->in_coff; kp.cin1 = a->cin; kp.nslab1 = w->nslab;
  }
  if (a->out2) {
    FLK_REQUIRE(a->cout1 > 0 && a->cout1 < a->cout && a->cout1 % 8 == 0 && a->out2_ld % 8 == 0 && a->out2_coff % 8 == 0 &&
                    a->out2_coff + a->cout - a->cout1 <= a->out2_ld && a->out_coff + a->cout1 <= a->out_ld,
                "flk_conv3d: bad out2 slice");
    FLK_REQUIRE(!a->add && !a->mask, "flk_conv3d: out2 cannot be combined with add / mask");
    kp.out2 = (char*)a->out2; kp.out2_ld = a->out2_ld; kp.out2_coff = a->out2_coff; kp.cout1 = a->cout1;
  } else {
    kp.out2 = kp.out; kp.cout1 = a->cout;
  }
  auto magic = [](int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); };
  kp.m_HW = magic(kp.FP); kp.m_Wh = magic(kp.Wh); kp.m_hw = magic((kp.tfast ? kp.Tt : kp.Ht) * kp.Wt); kp.m_Wt = magic(kp.Wt);
  kp.ntile_n = ntile_n;
  const long ptiles = (long)a->B * kp.nTt * kp.nTh * kp.nTw;
  // position tiles in contiguous chunks per XCD (halo-sharing neighbours in one L2): measured 6.82 -> 6.76 ms per step, conv kernels
  // 6.32 -> 6.23 ms serial.  1x1x1 launches have no halo to share: the interleaved order (tile i on XCD i % 8).
  kp.xcd_chunk = kp.ntaps > 1 ? (int)((ptiles + 7) / 8) : 0;
  const long gx = (ptiles + 7) / 8 * 8 * ntile_n;
  FLK_REQUIRE(gx < (1l << 31), "flk_conv3d: grid too large");
  // Deterministic split-K for launches that cannot fill the chip with output tiles (Mixed_5*: 3136 positions): the input-channel
  // slabs are divided over blockIdx.y, every slice writes raw fp32 partial sums and a second launch adds them in slice order
  // and runs the epilogue.  Needs the caller's workspace (flk_conv_args.splitk_ws) and a logical == physical output grid.
  kp.ksplit = 1;
  if (ksplit > 1) {
    const size_t npos = (size_t)a->B * a->OT * a->OH * a->OW;
    FLK_REQUIRE((size_t)ksplit * npos * a->cout * sizeof(float) <= (size_t)a->splitk_ws_bytes, "flk_conv3d: split-K workspace too small "
                "(%zu bytes needed, flk_conv_splitk_bytes)", (size_t)ksplit * npos * a->cout * sizeof(float));
    kp.ksplit = ksplit; kp.part = (float*)a->splitk_ws; kp.part_ld = a->cout; kp.npos = (unsigned)npos;
  }
  dim3 grid((unsigned)gx, (unsigned)kp.ksplit);
  hipStream_t s = (hipStream_t)stream;
  // Weight path.  Direct-A (modes 1/2) wherever a K step holds few MFMAs per wave and the waves would otherwise stall
  // on the per-step barrier: all WN >= 2 layouts.  Everything else shares the weights through the LDS ring (mode 0 / 5 / 6).
  // Until round 4 narrow channel tiles (nf <= 4) on grids of at most two workgroups per CU (<= 512) took direct-A as well; with the
  // row-ahead ring kernels and the conflict-free halo images the ring wins there too (bs 8, four interleaved rounds on one box, threshold 512 /
  // 448 / 336 / 256 / 0 workgroups: 5.694 / 5.685 / 5.674 / 5.658 / 5.644 ms per step; the other configurations neutral to -0.5 %).
  int mode = 0;
  {
    const bool k1 = kp.ntaps == 1 && kp.P <= 256;
    // direct A needs <= 4 fragments per wave (and >= 2 in bf16); the ring kernels are instantiated for wn == 1 only
    const int nfw = nf / wn;
    const bool da_ok = nf != 6 && nfw <= 4 && (dtype != FLK_BF16 || nfw >= 2), ring_ok = wn == 1;
    bool da = wn >= 2;
    if (force_da == 0 && ring_ok) da = false;
    if (force_da == 1 && da_ok) da = true;
    if (w->stem4) mode = 4;
    else if (da) mode = k1 ? 2 : 1;
    else if (k1 && kp.nslab >= 4) mode = 3;             // (2-3 slabs: the clamped tail loads would outweigh the prefetch)
  }
  // (3,1,1) convolutions, stride 1, pad 1 (the temporal half of a (2+1)D unit) whose frames split into 16-position chunks: the LDS-DMA
  // ring over whole-T tiles (conv_t3_dma_kernel).  Measured (r2plus1d_18, bs 8, same box): halo kernels 4.72 ms per step; R = 2 ring slots
  // (60 KB of LDS at nf 4: two workgroups per CU) 4.42; R = 3 / 4 (one per CU, two / three slabs in flight) 4.82 / 4.89 -- once more the
  // second resident workgroup is worth more than the deeper pipeline
  {
    const bool flat3 = a->kt == 3 && a->kh == 1 && a->kw == 1 && a->st == 1 && a->sh == 1 && a->sw == 1 && a->pt == 1 && a->ph == 0 && a->pw == 0 &&
                       a->ost == 1 && a->osh == 1 && a->osw == 1 && a->oot == 0 && a->ooh == 0 && a->oow == 0 && a->To == a->Ti && a->Ho == a->Hi &&
                       a->Wo == a->Wi && a->OT == a->To && a->OH == a->Ho && a->OW == a->Wo;
    const long npos = (long)a->B * a->To * a->Ho * a->Wo;
    const int Tn = a->Ti, hw = a->Hi * a->Wi;
    // (the kernel builds its global byte offsets in 32 bits: position * in_ld * 2 + channel offset must stay below 2^32)
    const bool off32 = ((unsigned long long)npos * (unsigned)a->in_ld + (unsigned)a->in_coff + 32ull) * 2ull < (1ull << 32);
    if (!plan && off32 && dtype == FLK_BF16 && flat3 && !a->in2 && !a->out2 && !a->pos_bias && kp.ksplit == 1 && force_wn == 0 && force_da < 0 &&
        (nf == 8 || nf == 4) && npos >= 2048 && npos < (1l << 23) && (Tn == 2 || Tn == 4 || Tn == 8 || Tn == 16) && hw % 16 == 0) {
      const int Gn = 16 / Tn;
      const long pt = (long)a->B * ((hw / 16 + Gn - 1) / Gn);
      dim3 g((unsigned)((pt + 7) / 8 * 8 * ntile_n));
      static bool attr_t3[2][FLK_MAX_DEVICES] = {};
      constexpr int Rr = 2;
      const size_t l5 = (size_t)Rr * ((16 + 2 * Gn) * 1024 + 3 * nf * 1024);
      if (dbg_on()) fprintf(stderr, "conv 3x1x1 cin %d cout %d positions %ld | nf %d LDS-DMA ring over whole-T tiles, R %d, lds %zu, wgs %ld\n", a->cin, a->cout, npos, nf, Rr, l5, pt * ntile_n);
#define FLK_LAUNCH_T3(NFv, Rv, idx)                                                                                                \
      if (nf == NFv && Rr == Rv && l5 <= 160 * 1024) {                                                                              \
        if (int rc = flk_raise_lds_limit((const void*)conv_t3_dma_kernel<NFv, Rv>, 160 * 1024, attr_t3[idx])) return rc;           \
        FLK_LAUNCH_KERNEL((conv_t3_dma_kernel<NFv, Rv>), g, dim3(256), l5, s, kp);                                                 \
        flk_last_kernel_tag = "conv_t3_dma_kernel";                                                                                 \
        FLK_CHECK_HIP(hipGetLastError());                                                                                           \
        return FLK_OK;                                                                                                              \
      }
      FLK_LAUNCH_T3(4, 2, 0) FLK_LAUNCH_T3(8, 2, 1)
#undef FLK_LAUNCH_T3
    }
  }
  // 1x1x1 GEMMs over a flat position grid: both operands through the LDS-DMA ring (conv1x1_dma_kernel); what it cannot take goes to modes 2 / 3.
  {
    const bool flat = a->st == 1 && a->sh == 1 && a->sw == 1 && a->pt == 0 && a->ph == 0 && a->pw == 0 && a->ost == 1 && a->osh == 1 && a->osw == 1 &&
                      a->oot == 0 && a->ooh == 0 && a->oow == 0 && a->To == a->Ti && a->Ho == a->Hi && a->Wo == a->Wi && a->OT == a->To &&
                      a->OH == a->Ho && a->OW == a->Wo;
    const long npos = (long)a->B * a->To * a->Ho * a->Wo;
    // (32-bit global byte offsets in the kernel, over both input segments)
    const unsigned ld_max = (unsigned)(a->in2 && a->in2_ld > a->in_ld ? a->in2_ld : a->in_ld), co_max = (unsigned)(a->in2 && a->in2_coff > a->in_coff ? a->in2_coff : a->in_coff);
    const bool off32 = ((unsigned long long)npos * ld_max + co_max + 32ull) * 2ull < (1ull << 32);
    if (!plan && off32 && dtype == FLK_BF16 && kp.ntaps == 1 && flat && !a->pos_bias && kp.ksplit == 1 && force_wn == 0 && force_da < 0 &&
        (nf == 8 || nf == 6 || nf == 4) && kp.nslab >= 2 && npos >= 2048 && npos < (1l << 23)) {
      kp.npos = (unsigned)npos;
      const long pt = (npos + 255) / 256;
      dim3 g((unsigned)((pt + 7) / 8 * 8 * ntile_n));
      static bool attr_set[3][FLK_MAX_DEVICES] = {};
      // (3 ring slots: a 6-slot ring for the launches with at most one workgroup per CU -- Mixed_5*, 13 position tiles -- measured the same)
      if (dbg_on()) fprintf(stderr, "conv 1x1x1 cin %d cout %d positions %ld | nf %d LDS-DMA ring, wgs %ld\n", a->cin, a->cout, npos, nf, pt * ntile_n);
#define FLK_LAUNCH_DMA(NFv, idx)                                                                                                   \
      if (nf == NFv) {                                                                                                              \
        constexpr int Rv = 3;                                                                                                       \
        const size_t l5 = (size_t)Rv * (16384 + NFv * 1024);                                                                        \
        if (int rc = flk_raise_lds_limit((const void*)conv1x1_dma_kernel<NFv, Rv>, 96 * 1024, attr_set[idx])) return rc;           \
        FLK_LAUNCH_KERNEL((conv1x1_dma_kernel<NFv, Rv>), g, dim3(256), l5, s, kp);                                                 \
        flk_last_kernel_tag = "conv1x1_dma_kernel";                                                                                 \
        FLK_CHECK_HIP(hipGetLastError());                                                                                           \
        return FLK_OK;                                                                                                              \
      }
      FLK_LAUNCH_DMA(8, 0) FLK_LAUNCH_DMA(6, 1) FLK_LAUNCH_DMA(4, 2)
#undef FLK_LAUNCH_DMA
    }
  }
  // ring kernels of three-tap rows (kw = 3) with a large halo: the weights a row ahead (mode 5).  Measured,
  // same box, two rounds: Conv3d_2c 0.2594 -> 0.2541 ms forward, 0.2670 -> 0.2551 data-gradient; Mixed_3c Branch_1 0.2485 -> 0.2412 /
  // 0.2607 -> 0.2509; the (1,3,3) 64 -> 144 layer 0.1413 -> 0.1370 / 0.1163 -> 0.1103; 160 -> 320 at 25 088 positions 0.0945 -> 0.0921; the
  // step 5.82-5.89 -> 5.78 ms.  128- and 96-channel tiles carry two 16-byte pieces per thread and step (six registers quads in flight,
  // 248 / 208 VGPRs, two workgroups per CU as in mode 0): Mixed_3b Branch_1 96 -> 128 0.1498-0.1536 -> 0.1464-0.1485 ms, its data-gradient on
  // 96-channel tiles 0.158 -> 0.160 (alone: slower), the step 5.75 -> 5.70 (128 only) -> 5.68 ms (both; four A/B pairs each, same box).
  if (mode == 0 && dtype == FLK_BF16 && a->kw == 3 && kp.P > 256 && (nf == 8 || nf == 6 || nf == 4 || nf == 2)) mode = 5;
  // two halo images for small halos; the LDS weight ring only in mode 0 / 3 / 4 / 5
  const size_t ring_bytes = (mode == 0 || mode == 3 || mode == 4 || mode == 5) ? 2 * (size_t)nf * 1024 : 0;
  if (!w->stem4) {
    // workgroups per CU the registers allow: the ring kernels with <= 64-channel tiles hold three, everything else two
    pick_halo_layout(kp, ring_bytes, (mode == 0 || mode == 5) && nf <= 4 ? 3 : 2);
    // (large halos of the multi-tap forms are staged by LDS-DMA, 64 cells per wave-instruction: planes of whole blocks)
    kp.plane_b = kp.P > 256 && (mode == 0 || mode == 1 || mode == 5) ? (kp.P + 63) / 64 * 1024 : (kp.P * 16 + 255) / 256 * 256;
    kp.m_HW = magic(kp.FP); kp.m_hw = magic((kp.tfast ? kp.Tt : kp.Ht) * kp.Wt);
  }
  const size_t lds = (kp.P <= 256 ? 2 : 1) * (4 * (size_t)kp.plane_b + 64) + ring_bytes;
  {
    const bool dbg = dbg_on();
    if (dbg)
      fprintf(stderr, "conv %dx%dx%d s%d%d%d cin %d cout %d out %dx%dx%dx%d | nf %d wn %d tile %dx%dx%d rows %d halo %d (frame pitch %d + %d, rows %s) wgs %ld mode %d lds %zu\n",
              a->kt, a->kh, a->kw, a->st, a->sh, a->sw, a->cin, a->cout, a->B, a->To, a->Ho, a->Wo, nf, wn, kp.Tt, kp.Ht, kp.Wt,
              kp.rows, kp.P, kp.Hh * kp.Wh, kp.FP - kp.Hh * kp.Wh, kp.tfast ? "w-T-h" : "w-h-T", ptiles * ntile_n, mode, lds);
    if (dbg && kp.ksplit > 1) fprintf(stderr, "   split-K x%d (%d slabs)\n", kp.ksplit, kp.nslab);
  }
  if (plan) {
    kp.wn = wn;
    plan->kp = kp; plan->grid = grid; plan->lds = lds; plan->nf = nf; plan->wn = wn; plan->mode = mode;
    return FLK_OK;
  }
  if (kp.ksplit > 1) {
    const int rc = launch_any(kp, grid, lds, s, dtype, nf, wn, mode);
    if (rc) return rc;
    const int epl = dtype == FLK_BF16 ? 8 : 4;
    const size_t n = (size_t)kp.npos * ((a->cout + epl - 1) / epl);
    if (dtype == FLK_BF16) FLK_LAUNCH_KERNEL(conv_splitk_finish_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kp);
    else FLK_LAUNCH_KERNEL(conv_splitk_finish_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kp);
    FLK_CHECK_HIP(hipGetLastError());
    return FLK_OK;
  }
  return launch_any(kp, grid, lds, s, dtype, nf, wn, mode);
}

static int launch_any(const ConvKP& kp, dim3 grid, size_t lds, hipStream_t s, int dtype, int nf, int wn, int mode) {
#define FLK_LAUNCH0(TT, NFv, WNv)                                                           \
  if (nf == NFv && wn == WNv && mode == 0) return launch<TT, NFv, WNv, 0>(kp, grid, lds, s); \
  if (nf == NFv && wn == WNv && mode == 3) return launch<TT, NFv, WNv, 3>(kp, grid, lds, s)
#define FLK_LAUNCHD(TT, NFv, WNv)                                                         \
  if (nf == NFv && wn == WNv && mode == 1) return launch<TT, NFv, WNv, 1>(kp, grid, lds, s); \
  if (nf == NFv && wn == WNv && mode == 2) return launch<TT, NFv, WNv, 2>(kp, grid, lds, s)
  if (dtype == FLK_BF16) {
    if (mode == 4 && wn == 1 && nf == 4) return launch<bf16_t, 4, 1, 4>(kp, grid, lds, s);
    if (mode == 4 && wn == 1 && nf == 8) return launch<bf16_t, 8, 1, 4>(kp, grid, lds, s);
    if (mode == 4 && wn == 1 && nf == 2) return launch<bf16_t, 2, 1, 4>(kp, grid, lds, s);
    // ring write behind the barrier (mode 6): measured on <= 64-channel tiles only (Conv3d_2c 0.2525 / 0.2467 -> 0.2478 / 0.2402 ms forward /
    // data-gradient, Mixed_3c Branch_1 0.2348 -> 0.2308, 160 -> 320 at 25 088 positions 0.0955 -> 0.0922, the (1,3,3) 64 -> 144 layer 0.1398 ->
    // 0.1315; 128- and 96-channel tiles the same: they stay mode 5)
    if (mode == 5 && wn == 1 && nf == 2) return launch<bf16_t, 2, 1, 6>(kp, grid, lds, s);
    if (mode == 5 && wn == 1 && nf == 4) return launch<bf16_t, 4, 1, 6>(kp, grid, lds, s);
    if (mode == 5 && wn == 1 && nf == 8) return launch<bf16_t, 8, 1, 5>(kp, grid, lds, s);
    if (mode == 5 && wn == 1 && nf == 6) return launch<bf16_t, 6, 1, 5>(kp, grid, lds, s);
    FLK_LAUNCH0(bf16_t, 2, 1); FLK_LAUNCH0(bf16_t, 4, 1); FLK_LAUNCH0(bf16_t, 8, 1); FLK_LAUNCH0(bf16_t, 6, 1);
    FLK_LAUNCHD(bf16_t, 2, 1); FLK_LAUNCHD(bf16_t, 4, 1); FLK_LAUNCHD(bf16_t, 4, 2);
    FLK_LAUNCHD(bf16_t, 8, 2); FLK_LAUNCHD(bf16_t, 8, 4);
  } else if (dtype == FLK_F32) {
    FLK_LAUNCH0(float, 2, 1); FLK_LAUNCH0(float, 4, 1); FLK_LAUNCH0(float, 8, 1);
    FLK_LAUNCHD(float, 2, 1); FLK_LAUNCHD(float, 4, 1); FLK_LAUNCHD(float, 2, 2); FLK_LAUNCHD(float, 4, 2);
    FLK_LAUNCHD(float, 4, 4); FLK_LAUNCHD(float, 8, 2); FLK_LAUNCHD(float, 8, 4);
  }
#undef FLK_LAUNCH0
#undef FLK_LAUNCHD
  flk_set_error("flk_conv3d: unsupported dtype %d / nf %d / wn %d / mode %d", dtype, nf, wn, mode);
  return FLK_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// Grouped launch: n <= 3 convolutions (bf16, more than one tap, no split-K) in one grid of conv_igemm_group_kernel<NFW>.  Every member
// is planned as a launch of its own would be, with direct-A weights and wn = (its weights' nf) / nfw waves along N, so member i's blocks
// compute exactly what flk_conv3d would have with that layout.  Members in the order given: put the longest K loops first.
// plan of a grouped launch: member table, LDS size, grid, body mode (1 direct-A, 0 / 5 / 6 ring forms)
static int group_plan(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int nfw, int ring, int dtype, ConvGroupKP& g, size_t& lds,
                      long& total, int& gmode) {
  FLK_REQUIRE(a && w && n >= 1 && n <= FLK_MAX_GROUP, "flk_conv3d_group: 1..%d members", FLK_MAX_GROUP);
  FLK_REQUIRE(dtype == FLK_BF16 && (ring ? (nfw == 4 || nfw == 8) : (nfw == 2 || nfw == 4)),
              "flk_conv3d_group: bf16; 2 or 4 channel fragments per wave (direct-A members) or channel tiles of 4 or 8 fragments (ring members)");
  g = ConvGroupKP{};
  lds = 0;
  total = 0;
  bool all5 = ring != 0;
  for (int i = 0; i <= FLK_MAX_GROUP; ++i) g.start[i] = 0x7fffffff;
  for (int i = 0; i < n; ++i) {
    FLK_REQUIRE(a[i] && w[i] && w[i]->dev, "flk_conv3d_group: null member %d", i);
    if (ring) FLK_REQUIRE(w[i]->nf == nfw, "flk_conv3d_group: ring member %d packed with nf %d, the group's tile is %d", i, w[i]->nf, nfw);
    else FLK_REQUIRE(w[i]->nf % nfw == 0 && (w[i]->nf / nfw == 1 || w[i]->nf / nfw == 2 || w[i]->nf / nfw == 4),
                     "flk_conv3d_group: member %d packed with nf %d, not 1 / 2 / 4 waves of %d fragments", i, w[i]->nf, nfw);
    FLK_REQUIRE(w[i]->ntaps > 1 && !w[i]->stem4 && !a[i]->pos_bias, "flk_conv3d_group: member %d is not a multi-tap convolution", i);
    ConvPlan pl{};
    if (int rc = conv3d_impl(a[i], w[i], dtype, nullptr, ring ? 1 : w[i]->nf / nfw, ring ? 0 : 1, 0, &pl)) return rc;
    // (ring members: a member planned for mode 5 -- the ring with the weights a row ahead -- has mode 0's arguments and LDS layout; the group
    //  runs the mode-5 body when every member was planned so, the mode-0 body otherwise)
    all5 = all5 && pl.mode == 5;
    FLK_REQUIRE((ring ? (pl.mode == 0 || pl.mode == 5) : pl.mode == 1) && pl.wn * nfw == (ring ? nfw : pl.nf) && pl.kp.ksplit == 1 && pl.grid.y == 1,
                "flk_conv3d_group: member %d planned as mode %d, wn %d", i, pl.mode, pl.wn);
    g.m[i] = pl.kp;
    g.start[i] = (int)total;
    total += pl.grid.x;
    lds = pl.lds > lds ? pl.lds : lds;
    if (dbg_on())
      fprintf(stderr, "group member %d: conv %dx%dx%d cin %d cout %d out %dx%dx%dx%d | %s nfw %d wn %d tile %dx%dx%d rows %d halo %d wgs %u lds %zu\n", i,
              a[i]->kt, a[i]->kh, a[i]->kw, a[i]->cin, a[i]->cout, a[i]->B, a[i]->To, a[i]->Ho, a[i]->Wo, ring ? "ring" : "direct-A", nfw, pl.wn, pl.kp.Tt,
              pl.kp.Ht, pl.kp.Wt, pl.kp.rows, pl.kp.P, pl.grid.x, pl.lds);
  }
  FLK_REQUIRE(total < (1l << 31), "flk_conv3d_group: grid too large");
  gmode = ring ? (all5 ? (nfw == 4 ? 6 : 5) : 0) : 1;      // (ring write behind the barrier: the 64-channel tiles)
  return FLK_OK;
}

// validation + planning of flk_conv3d_group without a launch: a plan builder calls it once per group it emits, so that a layout the group
// kernel cannot run fails when the plan is BUILT, not as FLK_EINVAL on every step
extern "C" int flk_conv3d_group_check(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int nfw, int ring, int dtype) {
  ConvGroupKP g; size_t lds; long total; int gmode;
  return group_plan(a, w, n, nfw, ring, dtype, g, lds, total, gmode);
}

extern "C" int flk_conv3d_group(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int nfw, int ring, int dtype, void* stream) {
  // ring groups of 64-channel tiles on large grids (Branch_1 + Branch_2 of Mixed_3b / 3c at the benchmark batch): the persistent producer / consumer
  // kernel takes all members in one launch -- bitwise the same outputs
  if (pc_route_on() >= 2 && ring && nfw == 4 && dtype == FLK_BF16 && a && w && n >= 1 && flk_conv3d_pc_worthwhile(a, w, n, dtype)) return flk_conv3d_pc(a, w, n, dtype, stream);
  ConvGroupKP g; size_t lds; long total; int gmode;
  if (int rc = group_plan(a, w, n, nfw, ring, dtype, g, lds, total, gmode)) return rc;
  hipStream_t s = (hipStream_t)stream;
  static bool attr[7][FLK_MAX_DEVICES] = {};
#define FLK_LAUNCH_GROUP(NFWv, MODEv, idx)                                                                                           \
  if (nfw == NFWv && gmode == MODEv) {                                                                                                \
    if (int rc = flk_raise_lds_limit((const void*)conv_igemm_group_kernel<bf16_t, NFWv, MODEv>, 96 * 1024, attr[idx])) return rc;    \
    FLK_LAUNCH_KERNEL((conv_igemm_group_kernel<bf16_t, NFWv, MODEv>), dim3((unsigned)total), dim3(256), lds, s, g);                  \
  }
  FLK_LAUNCH_GROUP(2, 1, 0) FLK_LAUNCH_GROUP(4, 1, 1) FLK_LAUNCH_GROUP(4, 0, 2) FLK_LAUNCH_GROUP(8, 0, 3) FLK_LAUNCH_GROUP(4, 5, 4) FLK_LAUNCH_GROUP(8, 5, 5) FLK_LAUNCH_GROUP(4, 6, 6)
#undef FLK_LAUNCH_GROUP
  flk_last_kernel_tag = "conv_igemm_group_kernel";
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// the (waves along N, weight path) flk_conv3d's heuristics pick for this geometry with weights packed at `nf` (plan builders decide how to
// pack a grouped launch's members before any weights exist)
extern "C" int flk_conv_layout_query(const flk_conv_args* a, int nf, int dtype, int force_da, int* wn_out, int* mode_out) {
  FLK_REQUIRE(a && wn_out && mode_out && nf > 0, "flk_conv_layout_query: bad argument");
  flk_conv_weights w{};
  w.dev = (void*)1; w.kt = a->kt; w.kh = a->kh; w.kw = a->kw; w.cin = a->cin; w.cout = a->cout; w.dtype = dtype; w.nf = nf;
  const int epl = dtype == FLK_BF16 ? 8 : 4;
  w.nslab = (a->cin + 4 * epl - 1) / (4 * epl); w.ntaps = a->kt * a->kh * a->kw;
  w.cout_frags = ((a->cout + 15) / 16 + nf - 1) / nf * nf; w.nslab1 = w.nslab;
  ConvPlan pl{};
  if (int rc = conv3d_impl(a, &w, dtype, nullptr, 0, force_da, 0, &pl)) return rc;
  *wn_out = pl.wn; *mode_out = pl.mode;
  return FLK_OK;
}

// ------------------------------------------------------------------------------------------------
// Autotuning.  The layout heuristics above (rows per workgroup, weight path) are good for the big layers and mediocre for
// the many mid-sized ones, whose best layout depends on how their grid lands on 256 CUs.  In tuning mode every
// flk_conv3d call times its candidate layouts on its own operands (3 runs each after a warm-up, HIP events on the
// caller's stream), remembers the winner in the weights object under the call's geometry and runs it; later calls with
// the same geometry reuse it.  The arithmetic does not depend on the layout (same K order per output), so tuning
// changes speed only.
static thread_local int g_tuning = 0;
extern "C" int flk_conv_set_autotune(int on) { g_tuning = on != 0; return FLK_OK; }

extern "C" int flk_conv3d(const flk_conv_args* a, const flk_conv_weights* w, int dtype, void* stream) {
  if (!a || !w) return conv3d_impl(a, w, dtype, stream, 0, -1);
  // the large 3x3x3 stride-1 layers (Conv3d_2c_3x3 at the benchmark batch): the persistent producer / consumer kernel -- bitwise the same outputs
  if (pc_route_on() && !g_tuning && dtype == FLK_BF16 && (w->ntaps == 27 || (w->ntaps == 9 && w->kt == 1)) && w->nf == 4 && !a->splitk_ws && flk_conv3d_pc_worthwhile(&a, &w, 1, dtype))
    return flk_conv3d_pc(&a, &w, 1, dtype, stream);
  for (const flk_conv_weights::Tuned& tn : w->tuned)
    if (tn.B == a->B && tn.To == a->To && tn.Ho == a->Ho && tn.Wo == a->Wo) return conv3d_impl(a, w, dtype, stream, tn.wn, tn.da);
  if (!g_tuning || w->stem4) return conv3d_impl(a, w, dtype, stream, 0, -1);
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  FLK_CHECK_HIP(hipEventCreate(&e0));
  FLK_CHECK_HIP(hipEventCreate(&e1));
  const int nf = w->nf, wn_max = nf == 6 ? 1 : dtype == FLK_BF16 ? nf / 2 : nf;
  struct Cand { int wn, da; };
  std::vector<Cand> cands;
  cands.push_back({0, -1});                                       // the heuristic's choice
  for (int wn = 1; wn <= 4 && wn <= (wn_max < 1 ? 1 : wn_max); wn *= 2) {
    const int nfw = nf / wn;
    if (wn == 1) cands.push_back({1, 0});
    if (nf != 6 && nfw <= 4 && (dtype != FLK_BF16 || nfw >= 2)) cands.push_back({wn, 1});
  }
  float best_ms = 1e30f;
  Cand best = cands[0];
  for (const Cand& c : cands) {
    int rc = conv3d_impl(a, w, dtype, stream, c.wn, c.da);        // warm-up (also validates the candidate)
    if (rc) continue;
    FLK_CHECK_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < 3; ++r) (void)conv3d_impl(a, w, dtype, stream, c.wn, c.da);
    FLK_CHECK_HIP(hipEventRecord(e1, s));
    FLK_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FLK_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best_ms * 0.97f) { best_ms = ms; best = c; }         // candidates are listed heuristic-first: keep it on ties
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  w->tuned.push_back({a->B, a->To, a->Ho, a->Wo, best.wn, best.da});
  return conv3d_impl(a, w, dtype, stream, best.wn, best.da);
}

// 3x3x3 and 1x3x3 stride-1 convolution (Unit3D forward and data-gradient of the large I3D layers: Conv3d_2c_3x3 and the Mixed_3* Branch_1 /
// Branch_2 units, i3d.py:183-186, 200-209, 229-238; the 3x3x3 layers of r3d_18 / mc3_18 and the spatial halves of r2plus1d_18's layer1 units
// -- torchvision Conv2Plus1D, model.py:421 -- at large batches) with WAVE-SPECIALISED producers -- bf16 only, round 5.
//
// conv_igemm_kernel (conv_igemm.hip) runs 256-thread workgroups in which every wave loads, stages and multiplies; three of them share a
// CU, and its ablations (DESIGN_LOG.md, "ring write behind the barrier") show a launch costing the SUM of its memory phase and its MFMA
// phase: a workgroup spends a third of its life outside the tap loop (halo staging, epilogue) and inside it a step lasts ~1100 cycles for
// 768 cycles of MFMA issue.  Here ONE persistent 512-thread workgroup per CU walks a list of (position tile, channel tile) items:
//   * waves 0-3 (one per SIMD) are CONSUMERS: ds_read_b128 + MFMA only.  A consumer owns 16 NI positions x 64 channels (NI = 7: 448-row
//     tiles, 28 MFMAs per 11 fragment reads; NI = 8: 512 rows).  Registers: 16 NI accumulators; the weight fragments in three rotating
//     sets (read one step ahead, right behind the barrier that publishes them); the position fragments in ONE set -- the MFMAs run position-
//     fragment-major and a fragment's register is refilled for the next step one group of four MFMAs behind its last use.  The K loop is
//     (slab, frame of taps) x a fully unrolled body of nine steps: a step holds its barrier, its reads, eight address adds and its MFMAs,
//     one filler behind each MFMA; no condition -- an item's last step reads the first weights of the NEXT item.  A consumer leaves the tap
//     loop only for its epilogue (conv_igemm_kernel's arithmetic; one straight-line form per feature set; scale / bias through LDS);
//   * waves 4-5 stream the WEIGHTS: per K step (32 input channels x one tap) the tile's 4 KiB of MFMA A fragments, by LDS-DMA
//     (global_load_lds_dwordx4) into a ring of R = 6 slots, D = 4 steps ahead of the consumers, behind a counted s_waitcnt vmcnt, across
//     item boundaries; with an item's first weights, the batch-norm scale / bias of its 64 channels;
//   * waves 6-7 stage the HALO box of the NEXT 32-channel slab (or of the next item's first slab) into the second of two LDS images while the
//     consumers multiply out of the first, by LDS-DMA as well: one wave-instruction fills 64 consecutive cells of one chunk plane, every lane
//     with its own source address (a block of zeros for padding); two per step and wave, nothing in the last third of a slab.
// One s_barrier per K step, joined by all eight waves, is the only synchronisation: the barrier of step k publishes the weights of step
// k + 1 (landed: the streaming waves waited for them) and, before a slab's first read, its halo image; it frees the ring slot of step k - 1
// and, one step into a slab, the image of the slab before it.  Producers arrive early and wait.
// Same products, same K order per output (slab-major, taps t-h-w) and the same epilogue as conv_igemm_kernel: bitwise its results.
// Measured (MI355X, round 5): Conv3d_2c forward at half the benchmark batch 0.239 (conv_igemm_kernel) -> 0.218 (register-staged
// halo) -> 0.203 ms (LDS-DMA halo: 1 300 TFLOP/s), its data-gradient 0.233 -> 0.215 -> 0.188 (1 420); in-kernel clock 1.9-2.1 GHz, 605-620
// cycles per K step for 448 of MFMA issue; SQ MFMA busy 0.56 (conv_igemm_kernel 0.41-0.44).
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <map>
#include <type_traits>
#include "flk_internal.h"
#include "conv_common.h"

constexpr int PC_MAX_MEMBERS = 3;
constexpr int PC_R = 6;             // weight ring slots of 4 KiB (one K step of a 64-channel tile)
constexpr int PC_D = 4;             // weight look-ahead in K steps; PC_R >= PC_D + 2 (a slot is re-filled two barriers after its step)
constexpr int PC_MAX_HALO = 1024;   // halo slots per image (16 DMA blocks of 64 per chunk plane): 2 x (4 planes x 16 KiB + 64) + 6 x 4 KiB + 4 KiB (scale / bias) = 159 872 B of the 160 KiB
constexpr int PC_THREADS = 512;
constexpr int PC_NBLK = 16;         // DMA blocks (64 halo positions) per chunk plane
constexpr int PC_ROW = 9;            // K steps per frame of taps (3 x 3); a slab is kt of them (kt = 3: Unit3D; kt = 1: the spatial half of a (2+1)D unit)
static_assert(PC_R >= PC_D + 2, "ring too short for the look-ahead");
static_assert(PC_NBLK * 64 >= PC_MAX_HALO, "halo blocks do not cover the image");

// timing experiments (-DPC_ABLATE=bits builds only, tools/build_variant.py --src conv_pc.hip; WRONG results): 1 no MFMAs, 2 consumers at priority 0,
// 4 no halo staging, 8 no weight DMA, 16 no epilogue stores, 32 no position-fragment reads, 64 no weight-fragment reads, 256 no staggered start,
// 512 no halo loads (the staging waves still compute and write), 1024 no halo writes (they still load).  The product build compiles every PAB() to true.
#ifdef PC_ABLATE
#define PAB(bit) (!((PC_ABLATE) & (bit)))
#else
#define PAB(bit) true
#endif

// (bit 128: no barriers at all -- racy, WRONG results, every address still valid: what the loop costs without its synchronisation)
#define PC_BARRIER() do { if (PAB(128)) __builtin_amdgcn_s_barrier(); } while (0)

#ifdef PC_STAMP
// diagnostic build (-DPC_STAMP, tools/build_variant.py): consumer wave 0 of every workgroup stamps the K loop of its SECOND item -- shader
// cycles (s_memtime), 100 MHz real time (s_memrealtime), K steps -- into a buffer nothing else reads; flk_pc_stamps_read copies it out.
// In-kernel clock = cycles / realtime x 100 MHz; cycles per step against the 512 (NI = 8) of MFMA issue.
__device__ unsigned long long pc_stamps[512][8];
#endif

// 16 bytes of zeros in device memory: the source of every halo cell that holds no data
// (conv_common.h flk_zero16)

struct PcKP {
  ConvKP m[PC_MAX_MEMBERS];          // members of the launch (a grouped launch: Branch_1 and Branch_2 of an Inception block)
  int nmem;
  int cnt[PC_MAX_MEMBERS];           // items of member i per XCD: its xcd_chunk position tiles x its channel tiles
  int per_xcd;                       // sum of cnt
  int slots;                         // workgroups per XCD (gridDim.x / 8)
  int halo_bytes;                    // one LDS halo image (the largest member's)
};

// The items of a workgroup, in the order every wave of it walks them: XCD x (= blockIdx.x % 8 under round-robin dispatch; speed only)
// owns, per member, the position tiles [x * chunk, (x + 1) * chunk) with all their channel tiles back to back; its `slots` workgroups
// take the XCD-local indices slot, slot + slots, ...: at any time they work on neighbouring tiles (shared halos, one L2).
struct PcIter {
  int q, stride;
  int mi, ptile, ntile;
  __device__ __forceinline__ bool next(const PcKP& kp, int xcd) {
    while (q < kp.per_xcd) {
      int r = q, m = 0;
      q += stride;
      if (kp.nmem > 1 && r >= kp.cnt[0]) {
        r -= kp.cnt[0]; m = 1;
        if (kp.nmem > 2 && r >= kp.cnt[1]) { r -= kp.cnt[1]; m = 2; }
      }
      const ConvKP& p = kp.m[m];
      const int k = r / p.ntile_n, pt = xcd * p.xcd_chunk + k;
      if (pt >= p.B * p.nTt * p.nTh * p.nTw) continue;        // tail of the last XCD's chunk
      mi = m; ptile = pt; ntile = r - k * p.ntile_n;
      return true;
    }
    return false;
  }
};

__device__ static inline void pc_tile_origin(const ConvKP& p, int ptile, int& b, int& ot0, int& oh0, int& ow0) {
  int bid = ptile;
  const int tw = bid % p.nTw; bid /= p.nTw;
  const int th = bid % p.nTh; bid /= p.nTh;
  const int tt = bid % p.nTt;
  b = bid / p.nTt;
  ot0 = tt * p.Tt; oh0 = th * p.Ht; ow0 = tw * p.Wt;
}

typedef unsigned pc_u32x4 __attribute__((ext_vector_type(4)));

// 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction): every product below has operands < 2^24 -- rows and halo slots
// < 2^11, magic numbers < 2^20, positions < 2^24 (the host checks), channel strides < 2^13
__device__ static inline int pc_mul24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }
__device__ static inline int pc_fdiv(int x, unsigned magic) { return (int)(__umul24((unsigned)x, magic) >> 20); }
// loads / stores through GLOBAL (not flat) instructions: a pointer that went through pc_uniform has lost its address space
typedef __attribute__((address_space(1))) pc_u32x4 pc_gu32x4;
__device__ static inline uint4 pc_ld16(const char* p) { return __builtin_bit_cast(uint4, *(const pc_gu32x4*)(size_t)p); }
__device__ static inline void pc_st16(char* p, const uint4& v) { *(pc_gu32x4*)(size_t)p = __builtin_bit_cast(pc_u32x4, v); }

// a wave-uniform pointer, as far as the compiler is concerned too: it goes into the "s" operand of an asm statement (the scalar base of a SADDR
// load), and an "s" constraint on a value hipcc's divergence analysis could not prove uniform is instantiated with VECTOR registers
__device__ static inline int pc_u(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ static inline const char* pc_uniform(const char* p) {
  const unsigned long long a = (unsigned long long)(size_t)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (const char*)(size_t)(((unsigned long long)hi << 32) | lo);
}

// NI = position fragments per consumer wave: tiles of 64 NI rows (8: 8x8x8 boxes of the 56x56 layers; 7: 16x4x7 boxes of the 28x28 ones)
template <int NI>
__global__ __launch_bounds__(PC_THREADS, 2) void conv_pc_kernel(const PcKP kp) {
  typedef Prec<bf16_t> PR;
  typedef typename PR::frag frag;
  constexpr int EPL = 8, NFW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  char* const ring = smem + 2 * kp.halo_bytes;
  char* const sbarea = ring + PC_R * 4096;              // two 2-KiB areas (item parity): [1 KiB piece: scale of the item's 64 channels in its first 256 B | 1 KiB piece: bias]

  if (wave < 4) {
    // =================================================== consumers ===================================================
    const int q = lane >> 4, m = lane & 15;
    PcIter it{slot, kp.slots, 0, 0, 0};
    int rslot = 0;        // ring slot of the next step to be read
    int gslab = 0;        // slabs consumed so far by this workgroup: halo image = slab parity
    int nitem = 0;        // items consumed so far: parity = scale / bias area
    bool first_item = true;
    frag a0[NFW];         // weight fragments of an item's first step: read by the previous item's last step (by the prologue for the first item)
    if (PAB(2)) __builtin_amdgcn_s_setprio(1);      // the partner wave on this SIMD is a producer: its vector instructions take the leftover issue slots
    // The member's parameters, copied into registers when the MEMBER changes (a launch has one to three; a workgroup's items run member by
    // member).  Read in place -- kp.m[mi].field with a run-time mi -- every use is a scalar load + s_waitcnt of its own, and hipcc sinks
    // those loads into the innermost conditional blocks instead of hoisting them: the first build of this kernel spent 28 us per item in its
    // epilogue and 37 us per slab in the halo staging that way (4 x the whole conv_igemm_kernel launch).  pc_u / pc_uniform
    // (v_readfirstlane) make each copy a value the compiler cannot re-materialise from memory.  The rows' cells in the halo image depend on
    // the member's tile geometry only: computed with the copy (per ITEM the set-up was ~1 000 cycles, 3 % of a 54-step item).
    ConvKP p{};
    int cur_mi = -1, halo_bytes = 0, inner = 1;
    int rowpos[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) rowpos[i] = 0;
    auto row_cell24 = [&](int r, int& rt, int& rh, int& rw) {      // conv_common.h row_cell with 24-bit multiplies
      const int o = pc_fdiv(r, p.m_hw), rem = r - pc_mul24(o, inner);
      const int i = pc_fdiv(rem, p.m_Wt);
      rw = rem - pc_mul24(i, p.Wt);
      rt = p.tfast ? i : o;
      rh = p.tfast ? o : i;
    };
    while (it.next(kp, xcd)) {
#ifdef PC_STAMP
      const unsigned long long sti = __builtin_amdgcn_s_memtime();
#endif
      if (it.mi != cur_mi) {
        cur_mi = it.mi;
        const ConvKP& pk = kp.m[it.mi];
        p.rows = pc_u(pk.rows); p.FP = pc_u(pk.FP); p.Wh = pc_u(pk.Wh); p.plane_b = pc_u(pk.plane_b); p.tfast = pc_u(pk.tfast);
        p.Tt = pc_u(pk.Tt); p.Ht = pc_u(pk.Ht); p.Wt = pc_u(pk.Wt); p.m_hw = (unsigned)pc_u((int)pk.m_hw); p.m_Wt = (unsigned)pc_u((int)pk.m_Wt);
        p.nTt = pc_u(pk.nTt); p.nTh = pc_u(pk.nTh); p.nTw = pc_u(pk.nTw); p.nslab = pc_u(pk.nslab); p.kt = pc_u(pk.kt);
        p.To = pc_u(pk.To); p.Ho = pc_u(pk.Ho); p.Wo = pc_u(pk.Wo); p.OT = pc_u(pk.OT); p.OH = pc_u(pk.OH); p.OW = pc_u(pk.OW);
        p.out = (char*)pc_uniform(pk.out); p.out2 = p.out; p.out_ld = pc_u(pk.out_ld); p.out_coff = pc_u(pk.out_coff); p.cout = pc_u(pk.cout); p.cout1 = p.cout;
        p.scale = (const float*)pc_uniform((const char*)pk.scale); p.bias = (const float*)pc_uniform((const char*)pk.bias);
        p.add = pc_uniform(pk.add); p.add_ld = pc_u(pk.add_ld); p.add_coff = pc_u(pk.add_coff);
        p.mask = pc_uniform(pk.mask); p.mask_ld = pc_u(pk.mask_ld); p.mask_coff = pc_u(pk.mask_coff); p.relu = pc_u(pk.relu);
        halo_bytes = pc_u(kp.halo_bytes);
        inner = (p.tfast ? p.Tt : p.Ht) * p.Wt;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = wave * 16 * NI + i * 16 + m;
          int pos = 0;
          if (r < p.rows) {
            int rt, rh, rw;
            row_cell24(r, rt, rh, rw);
            pos = pc_mul24(rt, p.FP) + pc_mul24(rh, p.Wh) + rw;
          }
          rowpos[i] = pos * 16 + plane_off(q, p.plane_b);
        }
      }
      int b, ot0, oh0, ow0;
      pc_tile_origin(p, it.ptile, b, ot0, oh0, ow0);
      f32x4 acc[NFW][NI];
#pragma unroll
      for (int f = 0; f < NFW; ++f)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int fp16 = p.FP * 16, wh16 = p.Wh * 16;
      const int nsteps_item = p.nslab * PC_ROW * p.kt;
      // Registers: 128 accumulators; the weight fragments in THREE rotating sets (a step's four are used by all its MFMAs; the next step's are
      // read right behind the barrier that publishes them; three sets because the loop body below is nine steps long -- no copies, no
      // renaming at the back edge); the position fragments in ONE -- the MFMAs run position-fragment-major, and fragment i of the next step
      // is requested into bq[i] as soon as the four MFMAs that read it have been issued (its image was published long ago: the halo of a
      // slab is complete eight steps before the slab begins).
      // The K loop is (slab, dt) x a body of NINE steps (dh, dw unrolled): every tap offset inside a body is a constant added to the
      // body's base, so a step holds nothing but its barrier, 12 fragment reads (+ 8 address adds) and 32 MFMAs.  (The first form of
      // this loop advanced a (dt, dh, dw) cursor per step: ~40 scalar instructions and four branches at the loop's back edge, where no
      // MFMA is in flight -- the consumers issued instructions for 590 cycles of a 1320-cycle step, the matrix pipe was busy for 512.)
      frag a1[NFW], a2[NFW], bq[NI];      // (a0: declared outside the item loop -- it arrives holding this item's first weights)
      auto read_a = [&](frag (&d)[NFW]) {
        const char* const wr = ring + rslot * 4096 + lane * 16;
#pragma unroll
        for (int f = 0; f < NFW; ++f) d[f] = PAB(64) ? *(const frag*)(wr + f * 1024) : frag{};
        rslot = rslot + 1 == PC_R ? 0 : rslot + 1;
      };
      // one K step: the barrier that publishes the NEXT step's weights, their read (into an), then the MFMAs of the current step (weights
      // ac) position-fragment by position-fragment, each followed by the request for the next step's fragment (image + tap offset nb).
      // No condition anywhere: the last step of an item reads the first weights of the NEXT item (the streaming waves run across items;
      // after the workgroup's last item: a slot nobody uses) and position fragments nobody uses (the next item reads its own)
      auto step = [&](const frag (&ac)[NFW], frag (&an)[NFW], int cb, int nb) {
        PC_BARRIER();
        asm volatile("" : "+s"(nb), "+s"(cb));            // (one scalar + eight vector adds per step, not 24 precomputed address registers per body)
        __builtin_amdgcn_sched_barrier(0);
        // ONE filler behind every MFMA.  A v_mfma_f32_16x16x32_bf16 occupies the matrix pipe for 16 cycles and the SIMD's issue port for 8 of
        // them: the wave can issue about one other instruction per MFMA for free, and everything beyond that stalls the pipe -- with the
        // reads grouped behind every fourth MFMA a step cost its MFMA time PLUS its other instructions' time (337 ns for 200 ns of MFMAs).
        // The refill of a position-fragment register is requested one group (four MFMAs) behind the MFMAs that read it: fragment i - 1
        // of the NEXT step (tap offset nb) behind group i, the last fragment of THIS step (tap offset cb) behind group 0 -- it is not needed
        // before the last group.  The next step's weights: one fragment behind each of the first four groups.
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
          for (int f = 0; f < NFW; ++f) if (PAB(1)) PR::mma(ac[f], bq[i], acc[f][i]);
          if (i == 0) bq[NI - 1] = PAB(32) ? *(const frag*)(smem + (cb + rowpos[NI - 1])) : frag{};
          else bq[i - 1] = PAB(32) ? *(const frag*)(smem + (nb + rowpos[i - 1])) : frag{};
          if (i < NFW) an[i] = PAB(64) ? *(const frag*)(ring + rslot * 4096 + lane * 16 + i * 1024) : frag{};
          // per group: MFMA, address add, MFMA, position-fragment read, MFMA, weight-fragment read (groups 0-3), MFMA
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < NFW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        rslot = rslot + 1 == PC_R ? 0 : rslot + 1;
        __builtin_amdgcn_sched_barrier(0);
      };
      if (first_item) {
        first_item = false;
        // Workgroups that get one item FEWER than the busiest (the last round is not full) start late, by a different fraction of an item
        // each.  All workgroups of a launch run the same items in lock-step, so their epilogues -- 57 KB of stores per CU -- hit the memory
        // system together: measured 5 200 cycles per epilogue = 14.7 MB per round at ~6 TB/s, the chip's write bandwidth, with the matrix
        // pipes idle (in-kernel stamps, -DPC_STAMP); a workgroup alone stores its tile several times faster.  The late starters' epilogues
        // fall between the others'; their slack pays for it (they would have idled at the end).
        {
          const int nmax = (kp.per_xcd + kp.slots - 1) / kp.slots, rem = kp.per_xcd - (nmax - 1) * kp.slots;      // slots [0, rem) run nmax items
          if (PAB(256) && slot >= rem && nmax > 1) {
            const int late = kp.slots - rem, idx = slot - rem;
            const int item_cycles = nsteps_item * 620 + 6000;
            const int delay = (int)(((long long)item_cycles * (2 * idx + 1)) / (2 * late));
            for (int c = 0; c < delay; c += 1024) __builtin_amdgcn_s_sleep(16);
          }
        }
        PC_BARRIER();                     // b_0: the first weights and the first halo image of this workgroup
        read_a(a0);
      }
      {
        const int hb = (gslab & 1) * halo_bytes;
#pragma unroll
        for (int i = 0; i < NI; ++i) bq[i] = PAB(32) ? *(const frag*)(smem + (hb + rowpos[i])) : frag{};
      }
      const int nslab = p.nslab, kt = p.kt;
#ifdef PC_STAMP
      const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
      for (int sl = 0; sl < nslab; ++sl) {
        const int img = ((gslab + sl) & 1) * halo_bytes;
#pragma unroll 1
        for (int dt = 0; dt < kt; ++dt) {
          const int hb = img + dt * fp16;
          // the first step of the next body: the next frame of taps, or the next slab (the other image); behind the item's last body: any valid address
          const int nb9 = dt + 1 < kt ? hb + fp16 : ((gslab + sl + 1) & 1) * halo_bytes;
          step(a0, a1, hb, hb + 16);
          step(a1, a2, hb + 16, hb + 32);
          step(a2, a0, hb + 32, hb + wh16);
          step(a0, a1, hb + wh16, hb + wh16 + 16);
          step(a1, a2, hb + wh16 + 16, hb + wh16 + 32);
          step(a2, a0, hb + wh16 + 32, hb + 2 * wh16);
          step(a0, a1, hb + 2 * wh16, hb + 2 * wh16 + 16);
          step(a1, a2, hb + 2 * wh16 + 16, hb + 2 * wh16 + 32);
          step(a2, a0, hb + 2 * wh16 + 32, nb9);
        }
      }
#ifdef PC_STAMP
      const unsigned long long ste = __builtin_amdgcn_s_memtime();
      if (wave == 0 && lane == 0 && nitem == 1 && blockIdx.x < 512) {
        pc_stamps[blockIdx.x][0] = ste - st0; pc_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - sr0;
        pc_stamps[blockIdx.x][2] = (unsigned long long)(nslab * PC_ROW * kt); pc_stamps[blockIdx.x][3] = (unsigned long long)NI;
        pc_stamps[blockIdx.x][4] = st0 - sti;             // item set-up: parameter copy, row decode, first position fragments
      }
#endif
      gslab += p.nslab;
      if constexpr (NI == 8) {
        // 128 accumulators: the epilogue has no registers to carry the rows' cells to the next item (92 spilled): this instance re-derives
        // the member state per item, as every instance did at first
#pragma unroll
        for (int i = 0; i < NI; ++i) rowpos[i] = 0;
        cur_mi = -1;
      }

      // ---- epilogue: lane = position m of fragment i, lane group q owns EPL channels of each of the two 32-channel store groups.  The arithmetic is
      // conv_igemm_kernel's (conv_common.h finish_store: acc * scale, rounded, + bias, rounded; + add; ReLU; mask; bf16), in the same order; the
      // two forms the plan uses are straight-line code chosen once per item -- forward (scale, bias, ReLU) and data-gradient (ReLU mask of the
      // layer's input) -- everything else goes through finish_store_row_pre.  (That general form alone was 2 700 instructions per item: a
      // run-time branch per feature and store group, flat instead of global accesses: 3.8 us per item, a sixth of the launch.)
      constexpr int NG = 4 * NFW / EPL;
      const int cbase = it.ntile * 16 * NFW + q * EPL;
      const bool hs = p.scale != nullptr, hbias = p.bias != nullptr, hadd = p.add != nullptr, hmask = p.mask != nullptr;
      const int ohw = p.OH * p.OW;
      const int obase = ((b * p.OT + ot0) * p.OH + oh0) * p.OW + ow0;      // output position of the tile's origin
      // output row of tile row r, or -1 (beyond the tile's rows / the tensor)
      auto out_row = [&](int i) -> int {
        int r = wave * 16 * NI + i * 16 + m;
        asm volatile("" : "+v"(r));      // (recomputed here: shared with the row decode in front of the K loop, hipcc keeps 8 x (cell, output offset) live across it -- spills)
        int rt, rh, rw;
        row_cell24(r, rt, rh, rw);
        const bool ok = r < p.rows && ot0 + rt < p.To && oh0 + rh < p.Ho && ow0 + rw < p.Wo;
        return ok ? obase + pc_mul24(rt, ohw) + pc_mul24(rh, p.OW) + rw : -1;
      };
      // one straight-line form per feature set (compile-time flags; chosen once per item), rows in batches: the add / mask operands of a
      // batch are all requested before the first is used (row by row, behind the branch that skips rows outside the tensor, every row paid its
      // own memory round trip)
      auto epi = [&](auto HS, auto ADD, auto RELU, auto MASK) {
        constexpr bool kHS = decltype(HS)::value, kADD = decltype(ADD)::value, kRELU = decltype(RELU)::value, kMASK = decltype(MASK)::value;
        constexpr int HALF = kADD && kMASK ? 2 : kADD || kMASK ? 4 : NI;      // rows per batch: its add / mask operands are in flight together
        // scale / bias of the lane's 16 channels: brought into LDS by the weight-streaming waves with the item's first weights (a global load
        // here would expose an L2 round trip per item with the matrix pipe idle)
        float4 sc[kHS ? NG : 1][2], bi[kHS ? NG : 1][2];
        if constexpr (kHS) {
          const char* const sa = sbarea + (nitem & 1) * 2048 + q * EPL * 4;
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              sc[g][h] = *(const float4*)(sa + g * 4 * EPL * 4 + h * 16);
              bi[g][h] = *(const float4*)(sa + 1024 + g * 4 * EPL * 4 + h * 16);
            }
        }
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += HALF) {
          int orow[HALF];
          uint4 av[kADD ? HALF : 1][NG], mv[kMASK ? HALF : 1][NG];
#pragma unroll
          for (int u = 0; u < HALF; ++u) {
            if (i0 + u >= NI) continue;
            orow[u] = out_row(i0 + u);
            const size_t orw = (size_t)(unsigned)(orow[u] < 0 ? 0 : orow[u]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              const int go = cbase + g * 4 * EPL < p.cout ? g * 4 * EPL * 2 : 0;
              if constexpr (kADD) av[u][g] = pc_ld16(p.add + (orw * p.add_ld + p.add_coff + cbase) * 2 + go);
              if constexpr (kMASK) mv[u][g] = pc_ld16(p.mask + (orw * p.mask_ld + p.mask_coff + cbase) * 2 + go);
            }
          }
#pragma unroll
          for (int u = 0; u < HALF; ++u) {
            if (i0 + u >= NI) continue;
            const int i = i0 + u;
            if (orow[u] < 0) continue;
            char* const dst = p.out + ((size_t)(unsigned)orow[u] * p.out_ld + p.out_coff + cbase) * 2;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              if (cbase + g * 4 * EPL >= p.cout) continue;
              float v[EPL], t[EPL];
#pragma unroll
              for (int e = 0; e < EPL; ++e) {
                v[e] = acc[(g * EPL + e) >> 2][i][(g * EPL + e) & 3];
                if constexpr (kHS) {
                  const float scv = e < 4 ? (&sc[g][0].x)[e] : (&sc[g][1].x)[e - 4], biv = e < 4 ? (&bi[g][0].x)[e] : (&bi[g][1].x)[e - 4];
                  v[e] = epi_scale_bias(v[e], scv, biv, true, true);
                }
              }
              if constexpr (kADD) {
                PR::to_f32(av[u][g], t);
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] += t[e];
              }
              if constexpr (kRELU) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] = fmaxf(v[e], 0.f);
              }
              if constexpr (kMASK) {
                PR::to_f32(mv[u][g], t);
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] = t[e] > 0.f ? v[e] : 0.f;
              }
              pc_st16(dst + g * 4 * EPL * 2, PR::from_f32(v));
            }
          }
        }
      };
      typedef std::true_type Y;
      typedef std::false_type N;
      const bool relu = p.relu != 0, hsb = hs && hbias;
      if (!PAB(16)) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int f = 0; f < NFW; ++f) asm volatile("" :: "v"(acc[f][i]));
      } else if (hsb && !hadd && relu && !hmask) epi(Y{}, N{}, Y{}, N{});       // Unit3D / conv + BN + ReLU forward
      else if (hsb && hadd && relu && !hmask) epi(Y{}, Y{}, Y{}, N{});          // ... with the block's residual (VideoResNet conv2)
      else if (hsb && !hadd && !relu && !hmask) epi(Y{}, N{}, N{}, N{});
      else if (!hs && !hbias && !hadd && !relu && hmask) epi(N{}, N{}, N{}, Y{});      // data-gradient: ReLU mask of the layer's input
      else if (!hs && !hbias && hadd && !relu && hmask) epi(N{}, Y{}, N{}, Y{});       // ... accumulated onto the shortcut's gradient
      else if (!hs && !hbias && hadd && !relu && !hmask) epi(N{}, Y{}, N{}, N{});
      else if (!hs && !hbias && !hadd && !relu && !hmask) epi(N{}, N{}, N{}, N{});
      else {
        // anything else (scale without bias, ...): conv_igemm_kernel's general epilogue
        float4 sg[NG][2], bg[NG][2];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int c = cbase + g * 4 * EPL, cc = c < p.cout ? c : 0;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            sg[g][h] = hs ? *(const float4*)(p.scale + cc + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);
            bg[g][h] = hbias ? *(const float4*)(p.bias + cc + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int orow = out_row(i);
          if (orow < 0) continue;
          float v[NG][EPL];
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[g][e] = acc[(g * EPL + e) >> 2][i][(g * EPL + e) & 3];
          finish_store_row_pre<bf16_t, NG>(p, (size_t)(unsigned)orow, cbase, v, sg, bg);
        }
      }
#ifdef PC_STAMP
      if (wave == 0 && lane == 0 && nitem == 1 && blockIdx.x < 512) pc_stamps[blockIdx.x][5] = __builtin_amdgcn_s_memtime() - ste;      // the epilogue
      if (wave == 0 && lane == 0 && nitem == 2 && blockIdx.x < 512) pc_stamps[blockIdx.x][6] = sti;                                      // (item 2's start: against item 1's end below)
      if (wave == 0 && lane == 0 && nitem == 1 && blockIdx.x < 512) pc_stamps[blockIdx.x][7] = __builtin_amdgcn_s_memtime();
#endif
      ++nitem;
    }
    return;
  }

  if (wave < 6) {
    // ============================================ weight streaming (LDS-DMA) ============================================
    const int w2 = wave - 4;
    // K steps of this workgroup = barriers every wave executes
    int total = 0;
    {
      PcIter c{slot, kp.slots, 0, 0, 0};
      while (c.next(kp, xcd)) total += kp.m[c.mi].nslab * PC_ROW * kp.m[c.mi].kt;
    }
    PcIter it{slot, kp.slots, 0, 0, 0};
    bool have = it.next(kp, xcd);
    int cstep = 0, cn = 0, islot = 0, citem = 0;
    const char* wt = nullptr;
    const char *scp = nullptr, *bip = nullptr;
    size_t wstep = 0;
    int sbmax = 0;
    // (member fields: read when the member changes, not per item -- an item switch sits between two barriers every wave waits at)
    int w_mi = -1, m_cn = 0, m_cout = 0;
    size_t m_wstep = 0;
    const char *m_w = nullptr, *m_sc = nullptr, *m_bi = nullptr;
    auto setup = [&]() {
      if (!have) return;
      if (it.mi != w_mi) {
        w_mi = it.mi;
        const ConvKP& p = kp.m[it.mi];
        m_cn = pc_u(p.nslab) * PC_ROW * pc_u(p.kt);
        m_wstep = (size_t)pc_u(p.cout_frags) * 1024;
        m_w = pc_uniform(p.w); m_sc = pc_uniform((const char*)p.scale); m_bi = pc_uniform((const char*)p.bias);
        m_cout = pc_u(p.cout);
      }
      cn = m_cn; wstep = m_wstep;
      wt = m_w + (size_t)it.ntile * 4 * 1024;
      scp = m_sc; bip = m_bi;
      sbmax = (m_cout - it.ntile * 64) * 4 - 16;      // last readable 16-byte piece of the tile's scale / bias rows (cout % 8 == 0)
    };
    setup();
    const unsigned ring0 = lds_addr32(ring);
    const unsigned voff0 = (unsigned)((2 * w2) * 1024 + lane * 16);
    auto issue = [&]() -> bool {                        // the two pieces this wave moves of the cursor's step; advance the cursor
      if (!have) return false;
      const char* const sb = pc_uniform(wt + (size_t)cstep * wstep);
      const unsigned lb = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring0 + (unsigned)(islot * 4096 + (2 * w2) * 1024)));
      if (cstep == 0 && w2 == 0 && scp && bip) {
        // with an item's first weights: scale and bias of its 64 channels into the scale / bias area of the item's parity, one piece each (lanes
        // 0-15 carry the 256 bytes, the others re-read them into the piece's unused part; channels past cout: clamped, never used).  Extra pieces
        // in this wave's queue only make the counted waits below stricter: the six youngest then reach less far back.
        const int l16 = lane & 15;
        const unsigned vo = (unsigned)(it.ntile * 256 + (l16 * 16 < sbmax ? l16 * 16 : (sbmax > 0 ? sbmax : 0)));
        const unsigned ab = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_addr32(sbarea) + (unsigned)((citem & 1) * 2048)));
        glds16(vo, scp, ab);
        glds16(vo, bip, ab + 1024u);
      }
      if (PAB(8)) {
        glds16(voff0, sb, lb);
        glds16(voff0 + 1024u, sb, lb + 1024u);
      }
      islot = islot + 1 == PC_R ? 0 : islot + 1;
      if (++cstep == cn) { have = it.next(kp, xcd); cstep = 0; ++citem; setup(); }
      return true;
    };
    bool more = true;
#pragma unroll
    for (int j = 0; j < PC_D; ++j) more = issue();
    // the pieces of step 0 have landed once at most those of steps 1 .. D - 1 are outstanding (fewer steps than D: everything)
    if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PC_D - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k < total; ++k) {
      PC_BARRIER();                       // b_k: step k's weights are published; slot (k + D) % R is free
      more = issue();                                     // step k + D
      // before b_{k+1}: step k + 1 has landed (steps k + 2 .. k + D may stay in flight); in the tail nothing newer is issued: drain
      if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PC_D - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (total) PC_BARRIER();              // b_total: the barrier of the consumers' last step (it publishes nothing)
    return;
  }

  // =================================================== halo staging ===================================================
  {
    // LDS-DMA: one wave-instruction writes 64 consecutive 16-byte cells of ONE chunk plane -- halo positions 64 j .. 64 j + 63, each lane
    // with its own 64-bit source address (the position's 16 bytes of the slab's chunk, or a block of zeros for padding, for the cells
    // between two frames and for chunks past cin).  Wave 6 fills planes 0 and 1, wave 7 planes 2 and 3: block j of both planes at step
    // j + 1 of the slab before (2 DMAs and ~10 vector instructions per step; nothing after step 16).  The first form of this role moved
    // every piece through registers (asm load -> counted wait -> zero select -> ds_write_b128, 66 pieces per slab and wave): with the
    // index arithmetic cached per position tile that staging still cost the launch 16 % (-DPC_ABLATE=4: 0.218 -> 0.182 ms; loads alone
    // 6 %, the selects and LDS writes alone 8 % -- a ds_write_b128 holds its SIMD's path to the LDS for 13 cycles, which the consumer on
    // that SIMD needs for its fragment reads).
    const int hwv = wave - 6;                             // planes 2 hwv, 2 hwv + 1
    struct Slab { int b, it0, ih0, iw0, s, img, P, FP, cells, Wh, Ti, Hi, Wi, in_ld, cin, plane_b, key, nblk; unsigned m_HW, m_Wh; const char* base; };
    // the member's part of a Slab: read when the member changes
    Slab mem{};
    int mem_mi = -1, m_nTt = 1, m_nTh = 1, m_nTw = 1, m_Tt = 1, m_Ht = 1, m_Wt = 1, m_pt = 0, m_ph = 0, m_pw = 0;
    auto slab_of = [&](const PcIter& c, int s, int img) {
      if (c.mi != mem_mi) {
        mem_mi = c.mi;
        const ConvKP& p = kp.m[c.mi];
        mem.P = pc_u(p.P); mem.FP = pc_u(p.FP); mem.Wh = pc_u(p.Wh); mem.cells = pc_u(p.Hh) * mem.Wh; mem.Ti = pc_u(p.Ti); mem.Hi = pc_u(p.Hi); mem.Wi = pc_u(p.Wi);
        mem.in_ld = pc_u(p.in_ld); mem.cin = pc_u(p.cin); mem.plane_b = pc_u(p.plane_b); mem.m_HW = (unsigned)pc_u((int)p.m_HW); mem.m_Wh = (unsigned)pc_u((int)p.m_Wh);
        mem.nblk = (mem.P + 63) >> 6;
        mem.base = pc_uniform(p.in + (size_t)p.in_coff * 2);
        m_nTt = pc_u(p.nTt); m_nTh = pc_u(p.nTh); m_nTw = pc_u(p.nTw); m_Tt = pc_u(p.Tt); m_Ht = pc_u(p.Ht); m_Wt = pc_u(p.Wt);
        m_pt = pc_u(p.pt); m_ph = pc_u(p.ph); m_pw = pc_u(p.pw);
      }
      Slab z = mem;
      z.s = s; z.img = img; z.key = (c.mi << 24) | c.ptile;      // (position tiles < 2^24: the host checks the positions)
      ConvKP t{};
      t.nTt = m_nTt; t.nTh = m_nTh; t.nTw = m_nTw; t.Tt = m_Tt; t.Ht = m_Ht; t.Wt = m_Wt;
      int ot0, oh0, ow0;
      pc_tile_origin(t, c.ptile, z.b, ot0, oh0, ow0);
      z.it0 = ot0 - m_pt; z.ih0 = oh0 - m_ph; z.iw0 = ow0 - m_pw;
      return z;
    };
    // halo position hp -> linear input position, or -1: no data (padding, a cell between two frames, beyond the image)
    auto cell_g = [&](const Slab& z, int hp) -> int {
      const int a = pc_fdiv(hp, z.m_HW), rem = hp - pc_mul24(a, z.FP);
      const int bq = pc_fdiv(rem, z.m_Wh), c = rem - pc_mul24(bq, z.Wh);
      const int it = z.it0 + a, ih = z.ih0 + bq, iw = z.iw0 + c;
      const bool ok = hp < z.P && rem < z.cells && (unsigned)it < (unsigned)z.Ti && (unsigned)ih < (unsigned)z.Hi && (unsigned)iw < (unsigned)z.Wi;
      const int g = pc_mul24(pc_mul24(z.b * z.Ti + it, z.Hi) + ih, z.Wi) + iw;
      return ok ? g : -1;
    };
    // Which input position a cell holds depends on the position tile only, not on the slab or the channel tile: the lane's byte offset of
    // block j's cell (from the member's first channel) is computed for the FIRST slab staged of a position tile and kept for its other
    // slabs and the channel tiles that follow; bit j of `live` = that cell holds data.
    unsigned o0 = 0u, o1 = 0u, o2 = 0u, o3 = 0u, o4 = 0u, o5 = 0u, o6 = 0u, o7 = 0u, o8 = 0u, o9 = 0u, o10 = 0u, o11 = 0u, o12 = 0u, o13 = 0u, o14 = 0u, o15 = 0u;
    unsigned live = 0u;
    Slab z{};
    z.key = -1;
    bool fresh = true;
    const char* const zeros = (const char*)flk_zero16;
    const int halo_b = pc_u(kp.halo_bytes);
    const unsigned lds0 = lds_addr32(smem);
    // what each image holds: (position tile, slab).  A member of TWO slabs alternates its images in step with its channel tiles -- the slab
    // the next channel tile starts with is still there (Conv3d_2c forward: 64 -> 192, three channel tiles per position tile: two stagings
    // in three skipped; the (1,3,3) 64 -> 144 of r2plus1d_18 likewise)
    int held0 = -1, held1 = -1;
    auto begin_slab = [&](const Slab& zn) {
      fresh = pc_u(zn.key != z.key) != 0;
      z = zn;
      const int tag = (z.key << 3) | (z.s & 7);             // (members of more than 8 slabs never find their slab again anyway)
      const int held = z.img ? held1 : held0;
      if (held == tag && z.s < 8) z.nblk = 0;
      if (z.img) held1 = tag; else held0 = tag;
    };
    // block j of this wave's two planes, slab z
#define PC_BLK(j)                                                                                                              \
    if (PAB(4) && (j) < z.nblk) {                                                                                              \
      if (fresh) {                                                                                                             \
        const int g = cell_g(z, 64 * (j) + lane);                                                                              \
        o##j = __umul24((unsigned)(g >= 0 ? g : 0), (unsigned)z.in_ld) * 2u;                                                   \
        live = (live & ~(1u << (j))) | ((g >= 0 ? 1u : 0u) << (j));                                                            \
      }                                                                                                                        \
      const bool lv = (live >> (j)) & 1u;                                                                                      \
      const unsigned lb = lds0 + (unsigned)(z.img * halo_b + (j) * 1024);                                             \
      _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                                                                       \
        const int c = 2 * hwv + k2;                                                                                            \
        const bool chv = z.s * 32 + c * EPL < z.cin;                                                                           \
        const char* const src = (lv && chv) ? z.base + (size_t)o##j + (size_t)((z.s * 32 + c * EPL) * 2) : zeros;             \
        if (PAB(512)) glds16_v64(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lb + (unsigned)plane_off(c, z.plane_b)))); \
      }                                                                                                                        \
    }
    PcIter it{slot, kp.slots, 0, 0, 0};                   // the item being consumed
    if (!it.next(kp, xcd)) return;
    PcIter nx = it;                                       // the item after it
    bool have_nx = nx.next(kp, xcd);
    int gslab = 0;
    {
      // the first slab of the workgroup: staged before b_0, synchronously
      begin_slab(slab_of(it, 0, 0));
      PC_BLK(0) PC_BLK(1) PC_BLK(2) PC_BLK(3) PC_BLK(4) PC_BLK(5) PC_BLK(6) PC_BLK(7) PC_BLK(8) PC_BLK(9) PC_BLK(10) PC_BLK(11) PC_BLK(12) PC_BLK(13) PC_BLK(14) PC_BLK(15)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    while (true) {
      const int nslab = pc_u(kp.m[it.mi].nslab), kt_it = pc_u(kp.m[it.mi].kt);      // slabs of the item being consumed; 27 or 9 K steps per slab
      for (int s = 0; s < nslab; ++s, ++gslab) {
        // slab (it, s) is being consumed out of image gslab & 1; the slab after it goes into the other image while that happens: issued
        // from step 1 on (the image's last reader, the slab before this one, has been read to its end by everyone behind the barrier of
        // step 1) and complete -- vmcnt(0) -- some steps before its first reader.  Straight-line per slab length (a run-time step loop that
        // picks its blocks cost the 27-step layers 40 %: 181 spilled scalars, v_readlane in front of every compare).
        const bool has_next = s + 1 < nslab || have_nx;
        PC_BARRIER();                                     // step 0
        if (has_next) begin_slab(s + 1 < nslab ? slab_of(it, s + 1, (gslab + 1) & 1) : slab_of(nx, 0, (gslab + 1) & 1));      // (behind the barrier: nobody waits for it)
        else z.nblk = 0;                                  // (the workgroup's last slab: nothing to stage)
        if (kt_it == 3) {
          // 27 steps: one block (two DMAs) per step, complete at step 22
          PC_BARRIER(); PC_BLK(0)                         // step 1
          PC_BARRIER(); PC_BLK(1)
          PC_BARRIER(); PC_BLK(2)
          PC_BARRIER(); PC_BLK(3)
          PC_BARRIER(); PC_BLK(4)
          PC_BARRIER(); PC_BLK(5)
          PC_BARRIER(); PC_BLK(6)
          PC_BARRIER(); PC_BLK(7)
          PC_BARRIER(); PC_BLK(8)
          PC_BARRIER(); PC_BLK(9)
          PC_BARRIER(); PC_BLK(10)
          PC_BARRIER(); PC_BLK(11)
          PC_BARRIER(); PC_BLK(12)
          PC_BARRIER(); PC_BLK(13)
          PC_BARRIER(); PC_BLK(14)
          PC_BARRIER(); PC_BLK(15)                        // step 16
          PC_BARRIER();                                   // step 17
          PC_BARRIER();
          PC_BARRIER();
          PC_BARRIER();
          PC_BARRIER();
          PC_BARRIER();                                   // step 22
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PC_BARRIER();                                   // step 23
          PC_BARRIER();
          PC_BARRIER();
          PC_BARRIER();                                   // step 26
        } else {
          // 9 steps (one frame of taps): four blocks per step, complete at step 7
          PC_BARRIER(); PC_BLK(0) PC_BLK(1) PC_BLK(2) PC_BLK(3)              // step 1
          PC_BARRIER(); PC_BLK(4) PC_BLK(5) PC_BLK(6) PC_BLK(7)
          PC_BARRIER(); PC_BLK(8) PC_BLK(9) PC_BLK(10) PC_BLK(11)
          PC_BARRIER(); PC_BLK(12) PC_BLK(13) PC_BLK(14) PC_BLK(15)          // step 4
          PC_BARRIER();
          PC_BARRIER();
          PC_BARRIER();                                   // step 7
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PC_BARRIER();                                   // step 8
        }
      }
      if (!have_nx) break;
      it = nx;
      have_nx = nx.next(kp, xcd);
    }
#undef PC_BLK
    PC_BARRIER();                         // b_total: the barrier of the consumers' last step
  }
}

// ------------------------------------------------------------------------------------------------
// Host side.  A member is eligible when: bf16; 3x3x3 taps, stride 1, logical == physical output grid; one input and one output segment; no
// position-class bias; weights packed with 64-channel tiles (nf = 4).
static const char* pc_ineligible(const flk_conv_args* a, const flk_conv_weights* w, int dtype) {
  if (dtype != FLK_BF16 || w->dtype != FLK_BF16) return "bf16 only";
  if (!((a->kt == 3 || a->kt == 1) && a->kh == 3 && a->kw == 3 && w->kt == a->kt && w->kh == 3 && w->kw == 3)) return "3x3x3 or 1x3x3 taps only";
  if (!(a->st == 1 && a->sh == 1 && a->sw == 1 && a->ost == 1 && a->osh == 1 && a->osw == 1 && a->oot == 0 && a->ooh == 0 && a->oow == 0 &&
        a->To == a->OT && a->Ho == a->OH && a->Wo == a->OW)) return "stride 1 with a logical == physical output grid only";
  if (a->in2 || a->out2 || a->pos_bias || w->cin_split || w->stem4) return "one input segment, one output segment, no position-class bias";
  if (w->nf != 4) return "weights must be packed with 64-channel tiles (nf = 4)";
  if (a->cin != w->cin || a->cout != w->cout) return "arguments do not match the packed weights";
  return nullptr;
}

static int pc_plan_member(const flk_conv_args* a, const flk_conv_weights* w, int max_rows, ConvKP& kp, int& ni, int force_fp = 0, int force_tfast = 0) {
  kp = ConvKP{};
  kp.in = (const char*)a->in; kp.w = (const char*)w->dev; kp.out = (char*)a->out; kp.in2 = kp.in; kp.out2 = kp.out;
  kp.scale = a->scale; kp.bias = a->bias; kp.add = (const char*)a->add; kp.mask = (const char*)a->mask;
  kp.in_ld = a->in_ld; kp.in_coff = a->in_coff; kp.cin = a->cin; kp.in2_ld = a->in_ld; kp.in2_coff = a->in_coff; kp.cin1 = a->cin;
  kp.B = a->B; kp.Ti = a->Ti; kp.Hi = a->Hi; kp.Wi = a->Wi;
  kp.kt = a->kt; kp.kh = kp.kw = 3; kp.st = kp.sh = kp.sw = 1; kp.pt = a->pt; kp.ph = a->ph; kp.pw = a->pw;
  kp.To = a->To; kp.Ho = a->Ho; kp.Wo = a->Wo;
  kp.out_ld = a->out_ld; kp.out_coff = a->out_coff; kp.cout = a->cout; kp.cout1 = a->cout;
  kp.OT = a->OT; kp.OH = a->OH; kp.OW = a->OW; kp.ost = kp.osh = kp.osw = 1;
  kp.add_ld = a->add_ld; kp.add_coff = a->add_coff; kp.mask_ld = a->mask_ld; kp.mask_coff = a->mask_coff; kp.relu = a->relu;
  kp.nslab = w->nslab; kp.nslab1 = w->nslab; kp.ntaps = w->ntaps; kp.cout_frags = w->cout_frags; kp.ntile_n = w->cout_frags / 4;
  kp.ksplit = 1; kp.wn = 1;
  // the tile: the box of <= max_rows rows flk_choose_tile scores best under this kernel's halo budget
  const flk_tile t = flk_choose_tile(a->To, a->Ho, a->Wo, a->kt, 3, 3, 1, 1, 1, max_rows, 1008);      // (unpadded cells; the padded image is checked below)
  kp.Tt = t.Tt; kp.Ht = t.Ht; kp.Wt = t.Wt; kp.rows = t.Tt * t.Ht * t.Wt;
  kp.nTt = (a->To + t.Tt - 1) / t.Tt; kp.nTh = (a->Ho + t.Ht - 1) / t.Ht; kp.nTw = (a->Wo + t.Wt - 1) / t.Wt;
  kp.Th = t.Tt + a->kt - 1; kp.Hh = t.Ht + 2; kp.Wh = t.Wt + 2;
  const int cells = kp.Hh * kp.Wh;
  // frame pitch / row enumeration with the fewest extra LDS passes per position-fragment read (conv_halo_extra_passes) that fits the image
  int best = force_fp ? 0 : conv_halo_extra_passes(kp, 0, cells), best_t = force_fp ? force_tfast : 0, best_fp = force_fp ? force_fp : cells;
  for (int pad = 0; pad < 16 && best > 0; ++pad)
    for (int tfast = 0; tfast < 2 && best > 0; ++tfast) {
      const int FP = cells + pad;
      if ((kp.Th - 1) * FP + cells > PC_MAX_HALO) continue;
      const int c = conv_halo_extra_passes(kp, tfast, FP);
      if (c < best) { best = c; best_t = tfast; best_fp = FP; }
    }
  kp.tfast = best_t; kp.FP = best_fp; kp.P = (kp.Th - 1) * best_fp + cells;
  FLK_REQUIRE(kp.P <= PC_MAX_HALO && kp.rows <= 512, "flk_conv3d_pc: no tile fits (halo %d)", kp.P);
  kp.plane_b = (kp.P + 63) / 64 * 1024;      // whole DMA blocks: a block's 64 cells never reach into the next plane
  auto magic = [](int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); };
  kp.m_HW = magic(kp.FP); kp.m_Wh = magic(kp.Wh); kp.m_hw = magic((kp.tfast ? kp.Tt : kp.Ht) * kp.Wt); kp.m_Wt = magic(kp.Wt);
  const long ptiles = (long)a->B * kp.nTt * kp.nTh * kp.nTw;
  kp.xcd_chunk = (int)((ptiles + 7) / 8);
  ni = (kp.rows + 63) / 64;
  return FLK_OK;
}

static bool pc_dbg() { static const bool d = getenv("FLK_CONV_DBG") != nullptr; return d; }

static int pc_cus() {
  static int cus[FLK_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  const bool tracked = dev >= 0 && dev < FLK_MAX_DEVICES;
  int ncu = tracked ? cus[dev] : 0;
  if (!ncu) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
    if (tracked) cus[dev] = ncu;
  }
  return ncu;
}

// The launch for n members: validation, then the tile size.  A workgroup's life is its items back to back, an item costs its K steps at the
// launch's fragments-per-wave (NI, the largest member's: the template argument) plus an epilogue of about seven full steps, and the launch
// lasts as long as its busiest workgroup: with few rounds of long items the rounding of items / workgroups decides (Conv3d_2c's
// data-gradient at half the batch: 784 tiles of 8x8x8 on 256 workgroups are FOUR rounds of 162 steps for 3.06 rounds of work; 896 tiles of
// 8x8x7 are four rounds of seven eighths the length).  Candidates: boxes of at most 512, 448 and 384 rows; the cheapest busiest workgroup wins.
// *eff = the launch's useful share of (busiest workgroup x workgroups) under that model.
static int pc_plan(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int dtype, PcKP& best, int& ni_best, double* eff, double* steps = nullptr) {
  FLK_REQUIRE(a && w && n >= 1 && n <= PC_MAX_MEMBERS, "flk_conv3d_pc: 1..%d members", PC_MAX_MEMBERS);
  for (int i = 0; i < n; ++i) {
    FLK_REQUIRE(a[i] && w[i] && w[i]->dev, "flk_conv3d_pc: null member %d", i);
    if (const char* why = pc_ineligible(a[i], w[i], dtype)) { flk_set_error("flk_conv3d_pc: member %d: %s", i, why); return FLK_EINVAL; }
    FLK_REQUIRE(a[i]->cin % 8 == 0 && a[i]->cout % 8 == 0 && a[i]->in_ld % 8 == 0 && a[i]->in_coff % 8 == 0 && a[i]->out_ld % 8 == 0 && a[i]->out_coff % 8 == 0 &&
                    a[i]->in_coff + a[i]->cin <= a[i]->in_ld && a[i]->out_coff + a[i]->cout <= a[i]->out_ld,
                "flk_conv3d_pc: channel counts / strides / offsets must be multiples of 8 and slices inside their rows");
    FLK_REQUIRE(!a[i]->add || (a[i]->add_ld % 8 == 0 && a[i]->add_coff % 8 == 0), "flk_conv3d_pc: add ld/coff % 8");
    FLK_REQUIRE(!a[i]->mask || (a[i]->mask_ld % 8 == 0 && a[i]->mask_coff % 8 == 0), "flk_conv3d_pc: mask ld/coff % 8");
    FLK_REQUIRE(a[i]->B > 0 && a[i]->To > 0 && a[i]->Ho > 0 && a[i]->Wo > 0, "flk_conv3d_pc: bad dims");
    FLK_REQUIRE((long)a[i]->B * a[i]->Ti * a[i]->Hi * a[i]->Wi < (1l << 24) && (long)a[i]->B * a[i]->OT * a[i]->OH * a[i]->OW < (1l << 24) && a[i]->in_ld < 4096 &&
                    a[i]->out_ld < (1 << 24) && a[i]->mask_ld < (1 << 24),
                "flk_conv3d_pc: more than 2^24 positions (the kernel multiplies with 24-bit instructions)");
    FLK_REQUIRE((size_t)a[i]->B * a[i]->Ti * a[i]->Hi * a[i]->Wi * a[i]->in_ld < (1ull << 31) && (size_t)a[i]->B * a[i]->OT * a[i]->OH * a[i]->OW * a[i]->out_ld < (1ull << 31),
                "flk_conv3d_pc: tensor too large for 32-bit element offsets");
  }
  const int ncu = pc_cus();
  // the search below (three tile sizes x the bank-conflict model over 32 image layouts) costs tens of microseconds: its outcome depends on the
  // geometry only and is remembered per geometry (a plan launches the same few geometries every iteration)
  struct Geo { int max_rows, fp[PC_MAX_MEMBERS], tfast[PC_MAX_MEMBERS]; double eff; };
  static thread_local std::map<std::array<int, 4 + 2 * PC_MAX_MEMBERS>, Geo> cache;
  std::array<int, 4 + 2 * PC_MAX_MEMBERS> key{};
  key[0] = a[0]->B; key[1] = a[0]->To; key[2] = a[0]->Ho; key[3] = a[0]->Wo;
  for (int i = 0; i < n; ++i) { key[4 + 2 * i] = a[i]->cin; key[5 + 2 * i] = a[i]->cout + (a[i]->To != a[0]->To || a[i]->Ho != a[0]->Ho || a[i]->Wo != a[0]->Wo || a[i]->B != a[0]->B ? 1 << 20 : 0); }
  for (int i = 1; i < n; ++i)
    FLK_REQUIRE(a[i]->B == a[0]->B && a[i]->To == a[0]->To && a[i]->Ho == a[0]->Ho && a[i]->Wo == a[0]->Wo, "flk_conv3d_pc: the members of a launch share one output grid");
  const auto hit = cache.find(key);
  double best_cost = 1e300;
  const int cand[3] = {512, 448, 384};
  for (int max_rows : cand) {
    if (hit != cache.end() && max_rows != hit->second.max_rows) continue;
    PcKP kp{};
    kp.nmem = n;
    int ni_max = 0;
    for (int i = 0; i < n; ++i) {
      int ni = 0;
      if (int rc = hit != cache.end() ? pc_plan_member(a[i], w[i], max_rows, kp.m[i], ni, hit->second.fp[i], hit->second.tfast[i])
                                      : pc_plan_member(a[i], w[i], max_rows, kp.m[i], ni)) return rc;
      ni_max = std::max(ni_max, ni);
      kp.cnt[i] = kp.m[i].xcd_chunk * kp.m[i].ntile_n;
      kp.per_xcd += kp.cnt[i];
      kp.halo_bytes = std::max(kp.halo_bytes, 4 * kp.m[i].plane_b + 64);
    }
    ni_max = ni_max <= 7 ? 7 : 8;
    // one workgroup per CU (the grid a multiple of 8: blockIdx.x % 8 labels the XCD), never more workgroups than an XCD has items
    kp.slots = std::max(1, std::min(ncu / 8, kp.per_xcd));
    // busiest workgroup of an XCD with a full chunk: slot j takes the XCD-local items j, j + slots, ...
    double busiest = 0, useful = 0;
    for (int j = 0; j < kp.slots; ++j) {
      double t = 0;
      for (int q = j; q < kp.per_xcd; q += kp.slots) {
        int r = q, mi = 0;
        while (mi + 1 < n && r >= kp.cnt[mi]) r -= kp.cnt[mi++];
        t += (kp.m[mi].nslab * PC_ROW * kp.m[mi].kt + 7) * (ni_max / 8.0);
      }
      busiest = std::max(busiest, t);
    }
    for (int i = 0; i < n; ++i) useful += (double)a[i]->B * a[i]->To * a[i]->Ho * a[i]->Wo / 512.0 * kp.m[i].ntile_n * kp.m[i].nslab * PC_ROW * kp.m[i].kt / 8.0;
    if (busiest < best_cost) {
      best_cost = busiest; best = kp; ni_best = ni_max;
      if (eff) *eff = useful / (busiest * kp.slots);
      if (steps) *steps = busiest;
    }
  }
  if (hit == cache.end()) {
    Geo g{};
    g.max_rows = best.m[0].rows <= 384 ? 384 : best.m[0].rows <= 448 ? 448 : 512;
    for (int i = 0; i < n; ++i) { g.fp[i] = best.m[i].FP; g.tfast[i] = best.m[i].tfast; }
    // (max_rows as the candidate that produced the plan: re-derive it from the winning plan by re-planning member 0)
    for (int max_rows : cand) {
      ConvKP t{}; int ni = 0;
      if (pc_plan_member(a[0], w[0], max_rows, t, ni, best.m[0].FP, best.m[0].tfast) == FLK_OK && t.Tt == best.m[0].Tt && t.Ht == best.m[0].Ht && t.Wt == best.m[0].Wt) { g.max_rows = max_rows; break; }
    }
    cache[key] = g;
  }
  return FLK_OK;
}

// n <= 3 eligible convolutions in one persistent launch (one: a plain layer; two: Branch_1 + Branch_2 of an Inception block).  Members in
// the order given: put the longest K loops first -- the short members' items then fill the last round of the long ones.
extern "C" int flk_conv3d_pc(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int dtype, void* stream) {
  PcKP kp{};
  int ni = 8;
  double eff = 0;
  if (int rc = pc_plan(a, w, n, dtype, kp, ni, &eff)) return rc;
  if (pc_dbg())
    for (int i = 0; i < n; ++i)
      fprintf(stderr, "pc member %d: conv 3x3x3 cin %d cout %d out %dx%dx%dx%d | tile %dx%dx%d rows %d halo %d (frame pitch %d + %d, rows %s) items per XCD %d (%d slabs)\n", i,
              a[i]->cin, a[i]->cout, a[i]->B, a[i]->To, a[i]->Ho, a[i]->Wo, kp.m[i].Tt, kp.m[i].Ht, kp.m[i].Wt, kp.m[i].rows, kp.m[i].P, kp.m[i].Hh * kp.m[i].Wh,
              kp.m[i].FP - kp.m[i].Hh * kp.m[i].Wh, kp.m[i].tfast ? "w-T-h" : "w-h-T", kp.cnt[i], kp.m[i].nslab);
  const size_t lds = 2 * (size_t)kp.halo_bytes + (size_t)PC_R * 4096 + 4096;
  FLK_REQUIRE(lds <= 160 * 1024, "flk_conv3d_pc: %zu bytes of LDS", lds);
  hipStream_t s = (hipStream_t)stream;
  static bool attr[2][FLK_MAX_DEVICES] = {};
  const dim3 grid((unsigned)(8 * kp.slots));
  if (pc_dbg()) fprintf(stderr, "pc launch: %d members, %d items per XCD on %d workgroups each, NI %d, lds %zu, modelled efficiency %.2f\n", n, kp.per_xcd, kp.slots, ni, lds, eff);
  if (ni <= 7) {
    if (int rc = flk_raise_lds_limit((const void*)conv_pc_kernel<7>, 160 * 1024, attr[0])) return rc;
    FLK_LAUNCH_KERNEL(conv_pc_kernel<7>, grid, dim3(PC_THREADS), lds, s, kp);
  } else {
    if (int rc = flk_raise_lds_limit((const void*)conv_pc_kernel<8>, 160 * 1024, attr[1])) return rc;
    FLK_LAUNCH_KERNEL(conv_pc_kernel<8>, grid, dim3(PC_THREADS), lds, s, kp);
  }
  flk_last_kernel_tag = "conv_pc_kernel";
  FLK_CHECK_HIP(hipGetLastError());
  return FLK_OK;
}

// Should a plan send these convolutions to flk_conv3d_pc?  Eligible members, a modelled efficiency (pc_plan: useful share of busiest workgroup x
// workgroups, the channel tiles' padding included) of at least 0.65 and a busiest workgroup of at least 150 full K steps (~0.3 us each).  The
// persistent kernel pays where it keeps the chip busy for a while; one 160-KB workgroup per CU cannot start beside the previous kernel's tail and
// its first halo image and weights are staged with nothing to hide them.  With the register-staged halo of the first form the bounds were 0.8 / 300
// (Conv3d_2c at batch 1, ~160 steps, lost 0.05 ms per launch); since the LDS-DMA staging (cheaper start, a third of the producers' work) the
// same sweep reads: I3D bs 8 5.38 -> 5.35 ms per step, I3D bs 1 2.05 -> 1.88, mc3_18 bs 16 4.47 -> 4.37, r2plus1d_18 bs 8 4.17 -> 4.09 (its (1,3,3)
// layers), and the same within noise for 0.5-0.65 / 60-180.  FLK_PC_MIN_STEPS / FLK_PC_MIN_EFF override the bounds for A/B runs.
extern "C" int flk_conv3d_pc_worthwhile(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int dtype) {
  if (!a || !w || n < 1 || n > PC_MAX_MEMBERS) return 0;
  for (int i = 0; i < n; ++i)
    if (!a[i] || !w[i] || pc_ineligible(a[i], w[i], dtype)) return 0;
  PcKP kp{};
  int ni = 8;
  double eff = 0;
  double steps = 0;
  if (pc_plan(a, w, n, dtype, kp, ni, &eff, &steps) != FLK_OK) return 0;
  static const double min_steps = getenv("FLK_PC_MIN_STEPS") ? atof(getenv("FLK_PC_MIN_STEPS")) : 150.0;
  static const double min_eff = getenv("FLK_PC_MIN_EFF") ? atof(getenv("FLK_PC_MIN_EFF")) : 0.65;
  return eff >= min_eff && steps >= min_steps;
}

// The launch flk_conv3d_pc would make for these geometries, without weights or a device (plan builders, tests): member 0's tile, the
// fragments per consumer wave, and the model's two figures -- useful share of (busiest workgroup x workgroups), K steps of the busiest
// workgroup.  Returns what flk_conv3d_pc_worthwhile decides (1 / 0), or a negative FLK_E* code.
extern "C" int flk_conv3d_pc_query(const flk_conv_args* const* a, int n, int dtype, int* tile3, int* ni_out, double* eff_out, double* steps_out) {
  FLK_REQUIRE(a && n >= 1 && n <= PC_MAX_MEMBERS, "flk_conv3d_pc_query: 1..%d members", PC_MAX_MEMBERS);
  flk_conv_weights fw[PC_MAX_MEMBERS];
  const flk_conv_weights* wp[PC_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) {
    FLK_REQUIRE(a[i], "flk_conv3d_pc_query: null member %d", i);
    flk_conv_weights& w = fw[i];
    w.dev = (void*)1; w.kt = a[i]->kt; w.kh = a[i]->kh; w.kw = a[i]->kw; w.cin = a[i]->cin; w.cout = a[i]->cout; w.dtype = dtype; w.nf = 4;
    w.nslab = (a[i]->cin + 31) / 32; w.ntaps = a[i]->kt * a[i]->kh * a[i]->kw; w.cout_frags = ((a[i]->cout + 15) / 16 + 3) / 4 * 4; w.nslab1 = w.nslab;
    wp[i] = &w;
    if (const char* why = pc_ineligible(a[i], &w, dtype)) { flk_set_error("flk_conv3d_pc_query: member %d: %s", i, why); return FLK_EINVAL; }
  }
  PcKP kp{};
  int ni = 8;
  double eff = 0, steps = 0;
  if (int rc = pc_plan(a, wp, n, dtype, kp, ni, &eff, &steps)) return rc;
  if (tile3) { tile3[0] = kp.m[0].Tt; tile3[1] = kp.m[0].Ht; tile3[2] = kp.m[0].Wt; }
  if (ni_out) *ni_out = ni;
  if (eff_out) *eff_out = eff;
  if (steps_out) *steps_out = steps;
  return flk_conv3d_pc_worthwhile(a, wp, n, dtype);
}

// does flk_conv3d_pc take this convolution (0) -- or why not (a static string; plan builders decide packing and launch form with it)
extern "C" const char* flk_conv3d_pc_why_not(const flk_conv_args* a, const flk_conv_weights* w, int dtype) {
  if (!a || !w) return "null argument";
  return pc_ineligible(a, w, dtype);
}

#ifdef PC_STAMP
extern "C" int flk_pc_stamps_read(unsigned long long* out, int n) {
  FLK_CHECK_HIP(hipDeviceSynchronize());
  FLK_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_stamps), sizeof(unsigned long long) * 8 * (n < 512 ? n : 512)));
  return FLK_OK;
}
#endif

// Whole-network plan: victim classifier forward + backward-to-input as a flat list of kernel launches
// over pre-planned, resident buffers (static shapes; nothing is allocated or synchronised per step).
//
// I3D topology follows reference i3d.py:144-479; every Unit3D (i3d.py:51-71) is one flk_conv3d launch
// with BN(inference, no gamma, eps 1e-3) folded into an fp32 scale/bias epilogue + ReLU, writing its
// channel slice of the Inception concat buffer.  Backward is data-gradient only (only the perturbation
// is trainable: i3d_adversarial_main_single_video_npy.py:82): gradient buffers mirror the activation
// buffers and hold d(loss)/d(pre-ReLU output) ("G"), the ReLU mask being applied by the PRODUCING
// kernel's epilogue and the BN scale folded into the transposed weights.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <map>
#include <memory>
#include "flk_internal.h"

int flk_head_forward(const void* y, int ld, int coff, int C, int B, int Tn, int HW, const float* wt, const float* W,
                     const float* bias, int N, float* feat, float* logits, int dtype, hipStream_t s);
int flk_head_backward(const void* y, int ld, int coff, void* gy, int gld, int gcoff, int C, int B, int Tn, int HW,
                      const float* wt, const float* W, int N, const float* dlogits, float* dfeat, int use_mask, int dtype,
                      hipStream_t s);

// stem_grad.hip: the fused delta-gradient on a slice of the batch / its stage 2 (flk_stem_delta_grad = mask + one slice + stage 2)
int flk_stem_delta_grad_part(const flk_apply_args* a, int b0, int nb, const void* G, int g_ld, const float* wf_dev, float* scratch, hipStream_t s);
int flk_stem_delta_grad_finish(const flk_apply_args* a, int nb, float* scratch, float* gdelta, hipStream_t s);

namespace {

enum OpKind { K_CONV = 0, K_POOL = 1, K_HEAD = 2, K_OTHER = 3, K_FORK = 4, K_JOIN = 5 };
const char* kKindName[] = {"conv", "pool", "head", "other", "fork", "join"};
constexpr int kSideStreams = 2;   // independent Inception branches run on the caller's stream + 2 side streams

struct Op {
  std::string name;
  int kind;
  double flops;   // algorithmic flops (2*MAC) for conv ops, 0 otherwise
  double bytes;   // algorithmic HBM bytes (compulsory traffic) for memory-bound ops
  std::function<int(hipStream_t)> run;
  int nlaunch = 0;   // kernel launches of the operator's last run (run_ops counts them: single-launch operators can carry a stop event)
  int lane = 0;   // 0 = the caller's stream, 1..kSideStreams = side streams (between a fork and its join)
  int mask = ~0;  // fork: which side streams start here (bit l = side stream l); the others keep what they were doing
};

struct Act {      // channels-last activation tensor [B,T,H,W,ld]
  void* p = nullptr;
  int T = 0, H = 0, W = 0, ld = 0;
  size_t numel(int B) const { return (size_t)B * T * H * W * ld; }
};

struct ConvLayer {
  std::string name;
  int kt, kh, kw, cin, cout;
  std::vector<float> w;        // canonical [kt][kh][kw][cin][cout]
  std::vector<float> scale, bias;
  flk_conv_weights* wf = nullptr;
  flk_conv_weights* wb = nullptr;
  float* d_scale = nullptr;
  float* d_bias = nullptr;
  // generic (strided) layers: stride, symmetric padding, and the parity classes of the data-gradient
  int st = 1, sh = 1, sw = 1, pt = 0, ph = 0, pw = 0;
  struct BwdClass { flk_conv_weights* w; int kt, kh, kw, pbt, pbh, pbw, ot, oh, ow; };
  std::vector<BwdClass> bcls;
};

inline void same_pad(int n, int k, int s, int& out, int& before) {
  out = (n + s - 1) / s;
  int tot = (out - 1) * s + k - n;
  if (tot < 0) tot = 0;
  before = tot / 2;
}

// nf6: 96-channel tiles (bf16 ring kernels, wn = 1 only: launches with >= 256 position tiles) are a candidate -- they remove the padding
// of 96- and 176-channel outputs (25 % of the MFMAs of a 96-channel layer in a 128-wide tile).  Measured per layer (bs 8): they win
// where they replace a padded 128-wide tile (Mixed_3b/Branch_1 data-gradient, N = 96: 0.177 -> 0.165 ms; 3b fused GEMM, N = 176:
// 0.075 -> 0.068; 3c/Branch_2, N = 96: 0.067 -> 0.060) and LOSE against three exact 64-wide tiles (Conv3d_2c, N = 192: 0.510 ->
// 0.588 ms) -- hence the 1.15 weight, which keeps N = 192 on nf = 4
int choose_nf(int cout, int taps, bool nf6 = false) {
  int best = 8;
  double best_cost = 1e30;
  const int cand[4] = {8, 6, 4, 2};
  for (int nf : cand) {
    if (nf == 6 && !nf6) continue;
    const int padded = (cout + 16 * nf - 1) / (16 * nf) * (16 * nf);
    const int ntiles = padded / (16 * nf);
    // MFMA work ~ padded*taps (narrow tiles re-read the activation fragments more often per MFMA);
    // every extra N tile re-stages the halo
    const double cost = (double)padded * taps * (nf == 2 ? 1.35 : nf == 4 ? 1.1 : nf == 6 ? 1.15 : 1.0) + 64.0 * ntiles;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = nf; }
  }
  return best;
}

}  // namespace

struct flk_net {
  int arch = 0, dtype = 0, B = 0, T = 0, H = 0, W = 0, device = 0;
  int num_classes = 400;
  bool finalized = false, fwd_done = false;
  std::map<std::string, std::vector<float>> weights;
  std::vector<std::unique_ptr<ConvLayer>> convs;
  std::vector<void*> allocs;
  std::vector<void*> pool_gemm_weights;      // flk_pool_gemm_weights_create handles (fused Branch_3 backward)
  size_t alloc_bytes = 0;
  std::vector<Op> fwd, bwd;
  std::map<std::string, std::pair<Act, int>> named;   // endpoint name -> (tensor, channels)
  // bound per call
  const void* x_in = nullptr;
  void* gx_in = nullptr;
  float* logits_out = nullptr;
  const float* dlogits_in = nullptr;
  // fused stem delta-gradient (stem_grad.hip): fp32 weights, index of the stem data-gradient op it replaces, the stem's gradient buffer
  float* d_stem_wf = nullptr;
  int stem_dgrad_op = -1;
  Act stem_G;
  // set by flk_net_backward_delta for the duration of its run: the delta-gradient GEMM of clips [b0, b0 + nb), launched by the half-batch
  // operators "Conv3d_1a_7x7/dgrad/half" of the split stem segment on their own streams (nullptr: those operators launch nothing)
  std::function<int(int, int, hipStream_t)> delta_part;
  int stem_halves = 1;         // parts the backward stem segment was emitted in
  // exact perturbation path of the stem's forward in bf16 (flk_net_forward_flicker): class sums of the weights, per-call table
  float* d_stem_sums = nullptr;
  float* d_stem_tab = nullptr;
  const float* cur_pos_bias = nullptr;
  int64_t cur_pos_bias_bstride = 0;
  const flk_apply_args* cur_apply = nullptr;      // flk_net_forward_apply: the stem operators apply their batch slice first
  // the perturbation apply of clips [b0, b0 + nb) into the plan's input tensor, on stream s (no-op outside flk_net_forward_apply)
  flk_apply_args apply_args_slice(int b0, int nb) const {
    flk_apply_args sl = *cur_apply;
    sl.x = (const char*)sl.x + (size_t)b0 * sl.T * sl.H * sl.W * 3 * (sl.x_is_u8 ? 1 : 4);
    sl.B = nb;
    if (sl.delta_per_clip) {
      sl.delta += (size_t)b0 * sl.T * 3;
      if (sl.dclip_dev) sl.dclip_dev += b0;
    }
    return sl;
  }
  int apply_slice(int b0, int nb, void* dst, hipStream_t s) const {
    if (!cur_apply) return FLK_OK;
    const flk_apply_args sl = apply_args_slice(b0, nb);
    return flk_perturb_apply_s2d(&sl, dst, dtype, s);
  }
  // I3D in bf16, flk_net_forward_apply on a uint8 clip with the centred flicker perturbation: the stem reads the clip itself
  // (stem_fwd.hip: perturbation apply + 7x7x7 / 2 convolution in one kernel, 37 K steps instead of apply + 49); the space-to-depth
  // tensor is then neither written nor read.  FLK_STEM_U8=0 (read per call): the two-kernel path.
  flk_conv_weights* stem_u8_w = nullptr;
  flk_conv_weights* stem_hilo_w = nullptr;      // VideoResNet stems in bf16: forward weights over the two-bf16-numbers-per-value input (fold_t = 4)
  int in_ch = 16;                               // VideoResNet plans: channels of the input tensor (16, or 32 in bf16)
  bool stem_from_u8() const {
    return stem_u8_w && cur_apply && cur_pos_bias && cur_apply->x_is_u8 && cur_apply->center == 1 && !cur_apply->delta_dense &&
           !(getenv("FLK_STEM_U8") && atoi(getenv("FLK_STEM_U8")) == 0);
  }
  // head
  float *d_fcw = nullptr, *d_fcb = nullptr, *d_wt = nullptr, *d_feat = nullptr, *d_dfeat = nullptr;
  // profiling
  hipStream_t side[kSideStreams] = {nullptr, nullptr};
  hipStream_t mask_stream = nullptr;   // the stem clip-mask pre-pass has a stream of its own: on a branch lane it delayed that lane's first kernels
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[kSideStreams] = {nullptr, nullptr};
  hipEvent_t ev_main[2] = {nullptr, nullptr};     // "the caller's stream up to its last kernel before a join" (forks that follow a join directly)
  hipEvent_t ev_mask_fork = nullptr, ev_mask_done = nullptr;     // the stem clip-mask pre-pass runs beside the backward pass
  // flk_net_prepare_backward_delta: the clip mask of the coming flk_net_backward_delta is already being computed (into `premask_scratch`,
  // for these arguments) on mask_stream
  bool premask = false;
  float* premask_scratch = nullptr;
  flk_apply_args premask_args{};
  bool multi_stream = true;
  bool ext_events = true;      // fork / join events ride on kernels as stop events (FLK_EXT_EVENTS=0 at finalize: marker packets)
  bool tuning = false;
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_fwd, ev_bwd;
  std::vector<const char*> tag_fwd, tag_bwd;      // kernel each profiled launch went to (flk_last_kernel_tag)
  bool ev_fwd_valid = false, ev_bwd_valid = false;

  int esz() const { return flk_esize(dtype); }

  int dmalloc(void** p, size_t bytes, bool zero = false) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { flk_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return FLK_ENOMEM; }
    allocs.push_back(*p);
    alloc_bytes += bytes;
    if (zero) FLK_CHECK_HIP(hipMemset(*p, 0, bytes));
    return FLK_OK;
  }
  // workspace for deterministic split-K where flk_conv3d would split this launch (flk_conv_splitk_bytes: the Mixed_5* 3x3x3 layers,
  // the late VideoResNet layers); one per operator, because the branches of a block run concurrently.  bf16 (the performance
  // mode) only: in fp32, the parity mode, every output keeps ONE summation order whatever the batch size and launch layout
  // (tests compare plans of different batch sizes at fp32 summation-order noise).
  void attach_splitk(flk_conv_args& a, const flk_conv_weights* w) {
    if (dtype != FLK_BF16) return;
    const bool off = getenv("FLK_NO_SPLITK") && atoi(getenv("FLK_NO_SPLITK"));      // read per plan build
    const int64_t nb = off ? 0 : flk_conv_splitk_bytes(&a, w);
    void* ws = nullptr;
    if (nb > 0 && dmalloc(&ws, (size_t)nb) == FLK_OK) { a.splitk_ws = ws; a.splitk_ws_bytes = nb; }
  }
  int new_act(Act& a, int T_, int H_, int W_, int ld, bool zero = false) {
    a.T = T_; a.H = H_; a.W = W_; a.ld = ld;
    return dmalloc(&a.p, a.numel(B) * esz(), zero);
  }
  int upload(float** d, const std::vector<float>& h) {
    int rc = dmalloc((void**)d, h.size() * sizeof(float));
    if (rc) return rc;
    FLK_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return FLK_OK;
  }
  const std::vector<float>* find(const std::string& name, size_t numel) {
    auto it = weights.find(name);
    if (it == weights.end()) { flk_set_error("flk_net_finalize: missing weight '%s'", name.c_str()); return nullptr; }
    if (it->second.size() != numel) {
      flk_set_error("flk_net_finalize: weight '%s' has %zu elements, expected %zu", name.c_str(), it->second.size(), numel);
      return nullptr;
    }
    return &it->second;
  }

  // ---- conv layer construction -----------------------------------------------------------------
  // Unit3D parameters from the checkpoint names (kinetics_i3d_utils.py:41-62)
  int make_unit3d(const std::string& unit, int kt, int kh, int kw, int cin, int cout, ConvLayer** out) {
    const std::string pre = "RGB/inception_i3d/" + unit;
    auto* w = find(pre + "/conv_3d/w", (size_t)kt * kh * kw * cin * cout);
    auto* beta = find(pre + "/batch_norm/beta", cout);
    auto* mean = find(pre + "/batch_norm/moving_mean", cout);
    auto* var = find(pre + "/batch_norm/moving_variance", cout);
    if (!w || !beta || !mean || !var) return FLK_EINVAL;
    auto L = std::make_unique<ConvLayer>();
    L->name = unit; L->kt = kt; L->kh = kh; L->kw = kw; L->cin = cin; L->cout = cout;
    L->w = *w;
    L->scale.resize(cout); L->bias.resize(cout);
    for (int c = 0; c < cout; ++c) {
      const float a = 1.0f / sqrtf((*var)[c] + 1e-3f);      // snt.BatchNorm(eps=1e-3, scale=False)
      L->scale[c] = a;
      L->bias[c] = (*beta)[c] - (*mean)[c] * a;
    }
    *out = L.get();
    convs.push_back(std::move(L));
    return FLK_OK;
  }
  // pack forward + data-gradient operators and upload the epilogue vectors
  // 96-channel tiles are candidates for launches of `rows` output positions that keep 256-row workgroups (>= 256 position tiles), bf16
  bool nf6_ok(long rows) const {
    return dtype == FLK_BF16 && rows >= 256L * 256;
  }
  // (bf16 1x1x1 layers: at least 64-channel tiles, so that a 32-channel Branch_3 convolution goes through the LDS-DMA ring kernel too --
  // it is memory-bound, the padded MFMAs are free: 25 -> 22 us for 192 -> 32 at 200 704 positions)
  int nf_for(int cout, int taps, long rows) const {
    const int nf = choose_nf(cout, taps, nf6_ok(rows));
    // 1x1x1 GEMMs on 8 .. 255 position tiles (Mixed_4* / Mixed_5* at bs 8): 64-channel tiles, twice the workgroups of the 128-wide
    // choice -- these launches are paced by the step chain of ONE workgroup per CU (DMA issue -> LDS reads -> MFMAs, one wave per
    // SIMD), not by bytes: same total time alone, the step -0.035 ms (they overlap their neighbours better)
    if (dtype == FLK_BF16 && taps == 1 && rows >= 2048 && rows < 256L * 256 && cout > 64) return 4;
    // (the same for the 3x3x3 layers of those blocks: 6.122-6.124 vs 6.132-6.135 ms, within the noise; 96-wide: 6.32 ms -- not done)
    // multi-tap layers on >= 256 position tiles (Mixed_3*): 64- instead of 128-channel tiles -- since the ring write moved behind the
    // barrier on 64-channel tiles (conv_igemm.hip, mode 6) they are the faster kernels at equal padding (tools/nf_sweep.py, round 4:
    // 96 -> 128 at 8x32x28x28 0.134 vs 0.148 ms, 192 -> 128 0.245 vs 0.263)
    if (dtype == FLK_BF16 && taps > 1 && nf == 8 && rows >= 256L * 256) return 4;
    return (dtype == FLK_BF16 && taps == 1 && nf == 2) ? 4 : nf;
  }
  // (nf_f / nf_b > 0: the channel-tile width of the forward / data-gradient operator is given -- members of a grouped launch)
  int pack(ConvLayer* L, long rows = 0, int nf_f = 0, int nf_b = 0) {
    const int taps = L->kt * L->kh * L->kw;
    int rc = flk_conv_weights_create_impl(L->w.data(), L->kt, L->kh, L->kw, L->cin, L->cout, nullptr, 0, dtype,
                                          nf_f > 0 ? nf_f : nf_for(L->cout, taps, rows), 0, &L->wf);
    if (rc) return rc;
    rc = flk_conv_weights_create_impl(L->w.data(), L->kt, L->kh, L->kw, L->cin, L->cout, L->scale.data(), 1, dtype,
                                      nf_b > 0 ? nf_b : nf_for(L->cin, taps, rows), 0, &L->wb);
    if (rc) return rc;
    if ((rc = upload(&L->d_scale, L->scale))) return rc;
    if ((rc = upload(&L->d_bias, L->bias))) return rc;
    return FLK_OK;
  }

  // ---- batch slices: the emitters below build their operator for clips [bs_b0, bs_b0 + bs_nb) of every tensor (default: all).
  // The I3D stem segment is emitted once per half of the batch on two streams, so that the HBM-bound pools / 1x1x1 of one half
  // run beside the MFMA-bound convolutions of the other (per-sample arithmetic is unchanged: tiles never span clips).
  int bs_b0 = 0, bs_nb = 0;         // bs_nb == 0: the whole batch
  int nbatch() const { return bs_nb ? bs_nb : B; }
  char* bp(const Act& a) const { return (char*)a.p + (size_t)bs_b0 * a.T * a.H * a.W * a.ld * esz(); }
  const char* bp(const void* p, const Act& geom, int ld) const { return p ? (const char*)p + (size_t)bs_b0 * geom.T * geom.H * geom.W * ld * esz() : nullptr; }

  // compulsory HBM bytes of a convolution launch: input once, output once, plus the epilogue's add / mask operands (reported beside
  // the flops by the per-layer profile: the 1x1x1 layers are bound by these, the 3x3x3 / 7x7x7 ones by MFMA)
  double conv_bytes(const flk_conv_args& a) const {
    const double in = (double)a.B * a.Ti * a.Hi * a.Wi * a.cin, out = (double)a.B * a.To * a.Ho * a.Wo * a.cout;
    return (in + out * (1 + (a.add != nullptr) + (a.mask != nullptr))) * esz();
  }

  // ---- op emitters -------------------------------------------------------------------------------
  // forward Unit3D: out[:, coff:coff+cout] = relu(conv(in[:, in_coff:in_coff+cin]) * scale + bias), stride 1 SAME
  void emit_conv_fwd(ConvLayer* L, const Act& in, int in_coff, const Act& out, int out_coff, const void* const* in_ptr = nullptr) {
    flk_conv_args a{};
    a.in = bp(in); a.in_ld = in.ld; a.in_coff = in_coff; a.cin = L->cin;
    a.B = nbatch(); a.Ti = in.T; a.Hi = in.H; a.Wi = in.W;
    a.kt = L->kt; a.kh = L->kh; a.kw = L->kw; a.st = a.sh = a.sw = 1;
    a.pt = (L->kt - 1) / 2; a.ph = (L->kh - 1) / 2; a.pw = (L->kw - 1) / 2;
    a.To = out.T; a.Ho = out.H; a.Wo = out.W;
    a.out = bp(out); a.out_ld = out.ld; a.out_coff = out_coff; a.cout = L->cout;
    a.OT = out.T; a.OH = out.H; a.OW = out.W; a.ost = a.osh = a.osw = 1;
    a.scale = L->d_scale; a.bias = L->d_bias; a.relu = 1;
    const double macs = (double)nbatch() * out.T * out.H * out.W * L->kt * L->kh * L->kw * L->cin * L->cout;
    flk_conv_weights* wf = L->wf;
    const int dt = dtype;
    attach_splitk(a, wf);
    fwd.push_back(Op{L->name, K_CONV, 2.0 * macs, conv_bytes(a), [a, wf, dt, in_ptr](hipStream_t s) mutable {
                       if (in_ptr) a.in = *in_ptr;
                       return flk_conv3d(&a, wf, dt, s);
                     }});
  }
  // data gradient of a stride-1 SAME Unit3D: gin[:, gin_coff..] = (conv_T(G[:, g_coff..]) + add) masked
  void emit_conv_bwd(ConvLayer* L, const Act& G, int g_coff, const Act& gin, int gin_coff, const void* add, int add_ld,
                     int add_coff, const Act* mask, int mask_coff, void* const* out_ptr = nullptr) {
    flk_conv_args a{};
    a.in = bp(G); a.in_ld = G.ld; a.in_coff = g_coff; a.cin = L->cout;
    a.B = nbatch(); a.Ti = G.T; a.Hi = G.H; a.Wi = G.W;
    a.kt = L->kt; a.kh = L->kh; a.kw = L->kw; a.st = a.sh = a.sw = 1;
    a.pt = L->kt - 1 - (L->kt - 1) / 2; a.ph = L->kh - 1 - (L->kh - 1) / 2; a.pw = L->kw - 1 - (L->kw - 1) / 2;
    a.To = gin.T; a.Ho = gin.H; a.Wo = gin.W;
    a.out = bp(gin); a.out_ld = gin.ld; a.out_coff = gin_coff; a.cout = L->cin;
    a.OT = gin.T; a.OH = gin.H; a.OW = gin.W; a.ost = a.osh = a.osw = 1;
    a.add = bp(add, gin, add_ld); a.add_ld = add_ld; a.add_coff = add_coff;
    if (mask) { a.mask = bp(*mask); a.mask_ld = mask->ld; a.mask_coff = mask_coff; }
    const double macs = (double)nbatch() * gin.T * gin.H * gin.W * L->kt * L->kh * L->kw * L->cin * L->cout;
    flk_conv_weights* wb = L->wb;
    const int dt = dtype;
    attach_splitk(a, wb);
    bwd.push_back(Op{L->name + "/dgrad", K_CONV, 2.0 * macs, conv_bytes(a), [a, wb, dt, out_ptr](hipStream_t s) mutable {
                       if (out_ptr) a.out = *out_ptr;
                       return flk_conv3d(&a, wb, dt, s);
                     }});
  }
  struct PoolRec { flk_pool_args a; uint8_t* idx_base = nullptr; };
  int emit_pool_fwd(const std::string& name, const Act& in, int C, int kt, int kh, int kw, int st, int sh, int sw, Act& out,
                    PoolRec& rec, bool relu_input = false) {
    flk_pool_args a{};
    a.relu_input = relu_input;
    a.in = bp(in); a.in_ld = in.ld; a.in_coff = 0; a.C = C;
    a.B = nbatch(); a.Ti = in.T; a.Hi = in.H; a.Wi = in.W;
    a.kt = kt; a.kh = kh; a.kw = kw; a.st = st; a.sh = sh; a.sw = sw;
    same_pad(in.T, kt, st, a.To, a.pt); same_pad(in.H, kh, sh, a.Ho, a.ph); same_pad(in.W, kw, sw, a.Wo, a.pw);
    int rc;
    if (!out.p && (rc = new_act(out, a.To, a.Ho, a.Wo, C))) return rc;      // (a later batch slice reuses the buffers of the first:
    if (!rec.idx_base) {                                                      //  its PoolRec arrives with idx_base already set)
      void* idx = nullptr;
      if ((rc = dmalloc(&idx, (size_t)B * a.To * a.Ho * a.Wo * C))) return rc;
      rec.idx_base = (uint8_t*)idx;
    }
    a.out = bp(out); a.out_ld = C; a.out_coff = 0; a.idx = rec.idx_base + (size_t)bs_b0 * a.To * a.Ho * a.Wo * C;
    rec.a = a;
    const int dt = dtype;
    const double bytes = ((double)in.numel(nbatch()) + out.numel(nbatch())) * esz() + (double)nbatch() * a.To * a.Ho * a.Wo * C;
    fwd.push_back(Op{name, K_POOL, 0.0, bytes, [a, dt](hipStream_t s) { return flk_maxpool3d_fwd(&a, dt, s); }});
    return FLK_OK;
  }
  void emit_pool_bwd(const std::string& name, const PoolRec& rec, const Act& gout, const Act& gin, const Act* mask) {
    const flk_pool_args a = rec.a;
    const int dt = dtype;
    const void* mp = mask ? mask->p : nullptr;
    const int mld = mask ? mask->ld : 0;
    const int nb = a.B;                                  // the slice the forward operator was built for (bs_b0 must match)
    const double bytes = ((double)gout.numel(nb) + gin.numel(nb) + (mask ? gin.numel(nb) : 0)) * esz() + (double)nb * a.To * a.Ho * a.Wo * a.C;
    const void* gop = bp(gout);
    void* gip = bp(gin);
    if (mask) mp = bp(*mask);
    bwd.push_back(Op{name + "/grad", K_POOL, 0.0, bytes, [a, dt, gop, gip, gout, gin, mp, mld](hipStream_t s) {
                       return flk_maxpool3d_bwd(&a, gop, gout.ld, 0, gip, gin.ld, 0, mp, mld, 0, dt, s);
                     }});
  }

  // ---- grouped launches: Branch_1's and Branch_2's 3x3x3 units of an Inception block in ONE grid (flk_conv3d_group) ----
  // Layout of a group over a [B,T,H,W] grid: the LARGE member (cin1 -> cout1) keeps the layout flk_conv3d's heuristics give it with
  // direct-A weights (nf1 channels x wn1 waves); that fixes the fragments per wave nfw = nf1 / wn1 every member shares.  The small member
  // (cout2 <= 128) takes the narrowest tile of wn2 in {1, 2, 4} waves x nfw fragments that holds its channels -- with the large member
  // filling the chip it no longer has to manufacture workgroups out of 64-row tiles.  nfw = 0: no group (fp32, a geometry the group
  // kernel has no instance for).
  struct GroupLayout { int nfw = 0, nf1 = 0, nf2 = 0, ring = 0; };
  GroupLayout plan_group(const Act& geom, int cin1, int cout1, int cout2) const {
    GroupLayout g;
    if (dtype != FLK_BF16) return g;
    flk_conv_args a{};
    a.B = B; a.Ti = a.To = a.OT = geom.T; a.Hi = a.Ho = a.OH = geom.H; a.Wi = a.Wo = a.OW = geom.W;
    a.kt = a.kh = a.kw = 3; a.st = a.sh = a.sw = 1; a.pt = a.ph = a.pw = 1; a.ost = a.osh = a.osw = 1;
    a.cin = cin1; a.cout = cout1; a.in_ld = cin1; a.out_ld = cout1;
    const long rows = (long)B * geom.T * geom.H * geom.W;
    int nf1 = nf_for(cout1, 27, rows);
    if (nf1 == 6) nf1 = 4;
    int wn1 = 1, mode = 0;
    // a large member the heuristics give the LDS weight ring (Mixed_4e / 4f forward: 560 workgroups) keeps the ring: as a direct-A member
    // of a group it ran 0.120 / 0.122 ms against 0.092 + 0.019 / 0.091 + 0.022 ms for the two launches
    // (such a block not grouped: 5.889 ms per step; as direct-A members: slower on the same box; as a RING group, both members on the ring's
    //  256-row tiles with Branch_1's channel tile -- what happens here: 5.871)
    if (flk_conv_layout_query(&a, nf1, dtype, -1, &wn1, &mode) != FLK_OK || mode != 1) {
      if ((mode == 0 || mode == 5) && wn1 == 1 && (nf1 == 4 || nf1 == 8)) { g.nfw = nf1; g.nf1 = nf1; g.nf2 = nf1; g.ring = 1; }
      return g;
    }
    if (flk_conv_layout_query(&a, nf1, dtype, 1, &wn1, &mode) != FLK_OK || mode != 1) return g;
    // measured (bs 8, same box, interleaved runs): no groups 5.94-5.96 ms per step; Mixed_4* grouped with the heuristic's fragments per
    // wave 5.89-5.92; with FOUR fragments per wave everywhere (the data-gradients of Mixed_4b-d on 128- instead of 64-row tiles) 5.88;
    // Mixed_5* grouped as well 5.98-6.04 (their split-K launches are faster than any one-slice layout)
    constexpr int nfw = 4;
    if (nf1 < nfw) nf1 = nfw;
    if (nf1 / nfw > 4) nf1 = 4 * nfw;
    int wn2 = 1;
    while (wn2 < 4 && 16 * nfw * wn2 < cout2) wn2 *= 2;
    g.nfw = nfw; g.nf1 = nf1; g.nf2 = nfw * wn2;
    return g;
  }
  // arguments of a stride-1 SAME Unit3D forward / data-gradient (emit_conv_fwd / emit_conv_bwd without the launch)
  flk_conv_args conv_fwd_args(ConvLayer* L, const Act& in, int in_coff, const Act& out, int out_coff) const {
    flk_conv_args a{};
    a.in = bp(in); a.in_ld = in.ld; a.in_coff = in_coff; a.cin = L->cin;
    a.B = nbatch(); a.Ti = in.T; a.Hi = in.H; a.Wi = in.W;
    a.kt = L->kt; a.kh = L->kh; a.kw = L->kw; a.st = a.sh = a.sw = 1;
    a.pt = (L->kt - 1) / 2; a.ph = (L->kh - 1) / 2; a.pw = (L->kw - 1) / 2;
    a.To = out.T; a.Ho = out.H; a.Wo = out.W;
    a.out = bp(out); a.out_ld = out.ld; a.out_coff = out_coff; a.cout = L->cout;
    a.OT = out.T; a.OH = out.H; a.OW = out.W; a.ost = a.osh = a.osw = 1;
    a.scale = L->d_scale; a.bias = L->d_bias; a.relu = 1;
    return a;
  }
  flk_conv_args conv_bwd_args(ConvLayer* L, const Act& G, int g_coff, const Act& gin, int gin_coff, const Act* mask, int mask_coff) const {
    flk_conv_args a{};
    a.in = bp(G); a.in_ld = G.ld; a.in_coff = g_coff; a.cin = L->cout;
    a.B = nbatch(); a.Ti = G.T; a.Hi = G.H; a.Wi = G.W;
    a.kt = L->kt; a.kh = L->kh; a.kw = L->kw; a.st = a.sh = a.sw = 1;
    a.pt = L->kt - 1 - (L->kt - 1) / 2; a.ph = L->kh - 1 - (L->kh - 1) / 2; a.pw = L->kw - 1 - (L->kw - 1) / 2;
    a.To = gin.T; a.Ho = gin.H; a.Wo = gin.W;
    a.out = bp(gin); a.out_ld = gin.ld; a.out_coff = gin_coff; a.cout = L->cin;
    a.OT = gin.T; a.OH = gin.H; a.OW = gin.W; a.ost = a.osh = a.osw = 1;
    if (mask) { a.mask = bp(*mask); a.mask_ld = mask->ld; a.mask_coff = mask_coff; }
    return a;
  }
  // one operator = the two members in one launch (the large member first: its long K loops start first, the small member's
  // workgroups fill the tail)
  void emit_group(std::vector<Op>& ops, const std::string& name, const flk_conv_args& a1, const flk_conv_weights* w1, const flk_conv_args& a2,
                  const flk_conv_weights* w2, int nfw, int ring = 0) {
    auto macs = [](const flk_conv_args& a) { return (double)a.B * a.To * a.Ho * a.Wo * a.kt * a.kh * a.kw * a.cin * a.cout; };
    const int dt = dtype;
    {
      // plan_group decided layout and packing through flk_conv_layout_query; flk_conv3d_group re-plans every member at launch.  Checked
      // once, here: a disagreement fails (falls back) when the plan is built, not as FLK_EINVAL on every step.  The fallback -- the two
      // members as launches of their own, in line -- computes the same values from the weights as packed.
      const flk_conv_args* av[2] = {&a1, &a2};
      const flk_conv_weights* wv[2] = {w1, w2};
      if (flk_conv3d_group_check(av, wv, 2, nfw, ring, dt) != FLK_OK) {
        fprintf(stderr, "[flicker_hip] %s: grouped launch refused (%s); running its members as two launches\n", name.c_str(), flk_last_error());
        ops.push_back(Op{name + "/member0", K_CONV, 2.0 * macs(a1), conv_bytes(a1), [a1, w1, dt](hipStream_t s) { return flk_conv3d(&a1, w1, dt, s); }});
        ops.push_back(Op{name + "/member1", K_CONV, 2.0 * macs(a2), conv_bytes(a2), [a2, w2, dt](hipStream_t s) { return flk_conv3d(&a2, w2, dt, s); }});
        return;
      }
    }
    ops.push_back(Op{name, K_CONV, 2.0 * (macs(a1) + macs(a2)), conv_bytes(a1) + conv_bytes(a2), [a1, w1, a2, w2, nfw, ring, dt](hipStream_t s) {
                       const flk_conv_args* av[2] = {&a1, &a2};
                       const flk_conv_weights* wv[2] = {w1, w2};
                       return flk_conv3d_group(av, wv, 2, nfw, ring, dt, s);
                     }});
  }

  static void set_lane(std::vector<Op>& v, size_t from, int lane) { for (size_t i = from; i < v.size(); ++i) v[i].lane = lane; }
  static void push_sync(std::vector<Op>& v, int kind, int mask = ~0) {
    v.push_back(Op{kind == K_FORK ? "@fork" : "@join", kind, 0.0, 0.0, nullptr});
    v.back().mask = mask;
  }
  int build_i3d();
  int build_videoresnet();
  int make_conv_tv(const std::string& wname, const std::string& bnname, int cout, int cin, int kt, int kh, int kw,
                   int st_, int sh_, int sw_, int pt_, int ph_, int pw_, ConvLayer** out);
  int pack_generic(ConvLayer* L);
  void emit_gen_fwd(ConvLayer* L, const Act& in, const Act& out, bool relu, const Act* add);
  void emit_gen_bwd(ConvLayer* L, const Act& G, const Act& gin, const void* add, int add_ld, const Act* mask);
};

// ---------------------------------------------------------------------------------------------------
// I3D (i3d.py:144-479).
int flk_net::build_i3d() {
  FLK_REQUIRE(T % 2 == 0 && T >= 16, "I3D: T must be even and >= 16 (got %d)", T);
  FLK_REQUIRE(H == 224 && W == 224, "I3D: the Logits endpoint's 2x7x7 VALID avg-pool + squeeze (i3d.py:461-471) "
              "requires 224x224 input (got %dx%d)", H, W);
  int rc;
  std::vector<std::function<void()>> bwd_emit;   // backward emitters, run in reverse at the end

  // ---- stem: Conv3d_1a_7x7 (7x7x7 / 2, 3->64) as a 4x4x4 / 1 convolution over the space-to-depth clip (fold_t = 3) ----
  ConvLayer* stem7 = nullptr;
  if ((rc = make_unit3d("Conv3d_1a_7x7", 7, 7, 7, 3, 64, &stem7))) return rc;
  ConvLayer* stem = nullptr;
  {
    auto L = std::make_unique<ConvLayer>();
    L->name = "Conv3d_1a_7x7"; L->kt = L->kh = L->kw = 4; L->cin = 32; L->cout = 64;
    L->w.assign((size_t)64 * 32 * 64, 0.f);
    for (int kt = 0; kt < 7; ++kt) for (int kh = 0; kh < 7; ++kh) for (int kw = 0; kw < 7; ++kw)
      for (int c = 0; c < 3; ++c) {
        const int jt = kt >> 1, qt = kt & 1, jh = kh >> 1, qh = kh & 1, jw = kw >> 1, qw = kw & 1;
        const int ch = (qt * 2 + qh) * 8 + qw * 3 + c;      // fold_t = 3 layout: one (qt,qh) parity per 16-byte chunk
        const float* src = &stem7->w[((((size_t)kt * 7 + kh) * 7 + kw) * 3 + c) * 64];
        float* dst = &L->w[((((size_t)jt * 4 + jh) * 4 + jw) * 32 + ch) * 64];
        for (int co = 0; co < 64; ++co) dst[co] = src[co];
      }
    L->scale = stem7->scale; L->bias = stem7->bias;
    stem = L.get();
    convs.push_back(std::move(L));
  }
  {
    // forward: K steps assembled from the non-zero chunks of the folded taps (49 instead of 64; bf16); data-gradient: the
    // generic transposed operator over the same folded tensor
    if ((rc = flk_conv_weights_create_s2d_stem(stem->w.data(), 64, dtype, choose_nf(64, 64), &stem->wf))) return rc;
    if ((rc = flk_conv_weights_create_impl(stem->w.data(), 4, 4, 4, 32, 64, stem->scale.data(), 1, dtype, choose_nf(32, 64), 0, &stem->wb))) return rc;
    if ((rc = upload(&stem->d_scale, stem->scale)) || (rc = upload(&stem->d_bias, stem->bias))) return rc;
  }
  if (dtype == FLK_BF16) {
    if ((rc = flk_stem_delta_grad_weights_create(stem7->w.data(), stem7->scale.data(), &d_stem_wf))) return rc;
    if ((rc = flk_stem_delta_bias_weights_create(stem7->w.data(), stem7->scale.data(), &d_stem_sums))) return rc;
    if (H == 224 && W == 224 && (rc = flk_stem_fwd_u8_weights_create(stem7->w.data(), &stem_u8_w))) return rc;
    if ((rc = dmalloc((void**)&d_stem_tab, (size_t)B * (T / 2) * 16 * 64 * sizeof(float), true))) return rc;   // (one table per clip: per-clip perturbations)
  }
  const int T1 = T / 2, H1 = H / 2, W1 = W / 2;
  Act xin; xin.T = T1; xin.H = H1; xin.W = W1; xin.ld = 32;       // bound per call
  Act a1, G1;
  if ((rc = new_act(a1, T1, H1, W1, 64)) || (rc = new_act(G1, T1, H1, W1, 64))) return rc;
  named["Conv3d_1a_7x7"] = {a1, 64};
  named["grad:Conv3d_1a_7x7"] = {G1, 64};
  stem_G = G1;
  // ---- Conv3d_2b_1x1, Conv3d_2c_3x3 and the buffers of the segment ----
  ConvLayer *c2b = nullptr, *c2c = nullptr;
  const long rows2 = (long)(B >= 4 && B % 2 == 0 ? B / 2 : B) * T1 * (H1 / 2) * (W1 / 2);      // per launch (half-batches)
  if ((rc = make_unit3d("Conv3d_2b_1x1", 1, 1, 1, 64, 64, &c2b)) || (rc = pack(c2b, rows2))) return rc;
  if ((rc = make_unit3d("Conv3d_2c_3x3", 3, 3, 3, 64, 192, &c2c)) || (rc = pack(c2c, rows2))) return rc;
  Act p2a, Gp2a, a2b, G2b, a2c, G2c, p3a, Gp3a;
  // The segment up to Mixed_3b alternates MFMA-bound convolutions (Conv3d_1a, Conv3d_2c) with HBM-bound pools and a 1x1x1, one
  // kernel at a time.  With an even batch >= 4 it is emitted once per HALF of the batch, the halves on two streams: the pools of one
  // half run beside the convolutions of the other.
  const bool split = B >= 4 && B % 2 == 0;
  const int nhalf = split ? 2 : 1;
  PoolRec r2a[2], r3a[2];
  const double stem_macs = (double)(B / nhalf) * T1 * H1 * W1 * 343.0 * 3 * 64;   // algorithmic (7x7x7x3), not the padded 4x4x4x32
  // Measured (bs 8): both halves started together 6.94 ms per step against 6.99 unsplit; the second half started one kernel late
  // (so that a pool always meets a convolution) 7.00 -- a convolution and a pool slow each other about as much as they overlap,
  // the gain comes from the tails of the half-size launches filling each other.
  if (split) push_sync(fwd, K_FORK, 1);
  for (int h = 0; h < nhalf; ++h) {
    if (split) { bs_b0 = h * (B / 2); bs_nb = B / 2; }
    const size_t m0 = fwd.size();
    {
      // SAME padding of the 7/2 conv on an even size is (2,3) -> in s2d space taps j=0..3 read o-1+j: pad-before 1
      flk_conv_args a{};
      a.in_ld = 32; a.in_coff = 0; a.cin = 32; a.B = nbatch(); a.Ti = T1; a.Hi = H1; a.Wi = W1;
      a.kt = a.kh = a.kw = 4; a.st = a.sh = a.sw = 1; a.pt = a.ph = a.pw = 1;
      a.To = T1; a.Ho = H1; a.Wo = W1; a.out = bp(a1); a.out_ld = 64; a.cout = 64;
      a.OT = T1; a.OH = H1; a.OW = W1; a.ost = a.osh = a.osw = 1;
      a.scale = stem->d_scale; a.bias = stem->d_bias; a.relu = 1;
      flk_conv_weights* wf = stem->wf;
      const int dt = dtype;
      const size_t in_off = (size_t)bs_b0 * T1 * H1 * W1 * 32 * esz();
      const int b0 = bs_b0;
      fwd.push_back(Op{"Conv3d_1a_7x7", K_CONV, 2.0 * stem_macs, conv_bytes(a), [this, a, wf, dt, in_off, b0](hipStream_t s) mutable {
                         a.in = (const char*)x_in + in_off;
                         a.pos_bias_bstride = cur_pos_bias_bstride;
                         a.pos_bias = cur_pos_bias ? cur_pos_bias + (size_t)b0 * cur_pos_bias_bstride : nullptr;   // flk_net_forward_flicker: the perturbation enters here, in fp32
                         if (stem_from_u8()) {
                           const flk_apply_args sl = apply_args_slice(b0, a.B);
                           return flk_stem_fwd_u8(&sl, stem_u8_w, a.scale, a.bias, a.pos_bias, a.pos_bias_bstride, a.out, a.out_ld, s);
                         }
                         if (int rc = apply_slice(b0, a.B, (char*)x_in + in_off, s)) return rc;
                         return flk_conv3d(&a, wf, dt, s);
                       }});
    }
    // main pools read ReLU outputs whose gradient is masked by (input > 0): relu_input makes the mask read unnecessary
    if (h) { r2a[h].idx_base = r2a[0].idx_base; r3a[h].idx_base = r3a[0].idx_base; }
    if ((rc = emit_pool_fwd("MaxPool3d_2a_3x3", a1, 64, 1, 3, 3, 1, 2, 2, p2a, r2a[h], true))) return rc;
    if (h == 0) {
      if ((rc = new_act(Gp2a, p2a.T, p2a.H, p2a.W, 64))) return rc;
      if ((rc = new_act(a2b, p2a.T, p2a.H, p2a.W, 64)) || (rc = new_act(G2b, p2a.T, p2a.H, p2a.W, 64))) return rc;
      if ((rc = new_act(a2c, p2a.T, p2a.H, p2a.W, 192)) || (rc = new_act(G2c, p2a.T, p2a.H, p2a.W, 192))) return rc;
    }
    emit_conv_fwd(c2b, p2a, 0, a2b, 0);
    emit_conv_fwd(c2c, a2b, 0, a2c, 0);
    if ((rc = emit_pool_fwd("MaxPool3d_3a_3x3", a2c, 192, 1, 3, 3, 1, 2, 2, p3a, r3a[h], true))) return rc;
    if (h == 0 && (rc = new_act(Gp3a, p3a.T, p3a.H, p3a.W, 192))) return rc;
    set_lane(fwd, m0, h);
  }
  bs_b0 = bs_nb = 0;
  if (split) push_sync(fwd, K_JOIN, 1);
  named["MaxPool3d_2a_3x3"] = {p2a, 64};
  named["grad:MaxPool3d_2a_3x3"] = {Gp2a, 64};
  named["Conv3d_2b_1x1"] = {a2b, 64};
  named["Conv3d_2c_3x3"] = {a2c, 192};
  named["grad:Conv3d_2b_1x1"] = {G2b, 64};
  named["grad:Conv3d_2c_3x3"] = {G2c, 192};
  named["MaxPool3d_3a_3x3"] = {p3a, 192};
  named["grad:MaxPool3d_3a_3x3"] = {Gp3a, 192};
  {
    // backward of the segment (emitters run in reverse: this block's second emitter runs first)
    flk_conv_args g{};
    g.in = G1.p; g.in_ld = 64; g.cin = 64; g.B = B; g.Ti = T1; g.Hi = H1; g.Wi = W1;
    g.kt = g.kh = g.kw = 4; g.st = g.sh = g.sw = 1; g.pt = g.ph = g.pw = 2;   // k-1-pad
    g.To = T1; g.Ho = H1; g.Wo = W1; g.out_ld = 32; g.cout = 32;
    g.OT = T1; g.OH = H1; g.OW = W1; g.ost = g.osh = g.osw = 1;
    flk_conv_weights* wb = stem->wb;
    const int dt = dtype;
    const double macs = stem_macs * nhalf;
    bwd_emit.push_back([this, g, wb, dt, macs]() {
      bwd.push_back(Op{"Conv3d_1a_7x7/dgrad", K_CONV, 2.0 * macs, conv_bytes(g), [this, g, wb, dt](hipStream_t s) mutable {
                         g.out = gx_in;
                         return flk_conv3d(&g, wb, dt, s);
                       }});
    });
    const PoolRec r2a0 = r2a[0], r2a1 = r2a[1], r3a0 = r3a[0], r3a1 = r3a[1];
    bwd_emit.push_back([=]() {
      if (split) push_sync(bwd, K_FORK, 1);
      for (int h = 0; h < nhalf; ++h) {
        if (split) { bs_b0 = h * (B / 2); bs_nb = B / 2; }
        const size_t m0 = bwd.size();
        emit_pool_bwd("MaxPool3d_3a_3x3", h ? r3a1 : r3a0, Gp3a, G2c, nullptr);
        emit_conv_bwd(c2c, G2c, 0, G2b, 0, nullptr, 0, 0, &a2b, 0);
        emit_conv_bwd(c2b, G2b, 0, Gp2a, 0, nullptr, 0, 0, nullptr, 0);
        emit_pool_bwd("MaxPool3d_2a_3x3", h ? r2a1 : r2a0, Gp2a, G1, nullptr);
        if (split) {
          // the half's share of the fused stem delta-gradient, right behind the gradient it consumes: the MFMA-bound GEMM of one half
          // runs beside the HBM-bound tail (1x1x1 data-gradient, pool backward) of the other instead of after both
          // (flk_net_backward_delta; a no-op in every other run of this list)
          const int b0 = bs_b0, nb = bs_nb;
          bwd.push_back(Op{"Conv3d_1a_7x7/dgrad/half", K_CONV, 0.0, 0.0, [this, b0, nb](hipStream_t st) { return delta_part ? delta_part(b0, nb, st) : FLK_OK; }});
        }
        set_lane(bwd, m0, h);
      }
      bs_b0 = bs_nb = 0;
      if (split) push_sync(bwd, K_JOIN, 1);
      stem_halves = nhalf;
    });
  }

  // ---- Inception blocks ----
  struct Blk { const char* name; int c[6]; int pool_before; int pk[3]; int ps[3]; const char* pool_name; };
  const Blk blocks[] = {
      {"Mixed_3b", {64, 96, 128, 16, 32, 32}, 0, {0, 0, 0}, {0, 0, 0}, ""},
      {"Mixed_3c", {128, 128, 192, 32, 96, 64}, 0, {0, 0, 0}, {0, 0, 0}, ""},
      {"Mixed_4b", {192, 96, 208, 16, 48, 64}, 1, {3, 3, 3}, {2, 2, 2}, "MaxPool3d_4a_3x3"},
      {"Mixed_4c", {160, 112, 224, 24, 64, 64}, 0, {0, 0, 0}, {0, 0, 0}, ""},
      {"Mixed_4d", {128, 128, 256, 24, 64, 64}, 0, {0, 0, 0}, {0, 0, 0}, ""},
      {"Mixed_4e", {112, 144, 288, 32, 64, 64}, 0, {0, 0, 0}, {0, 0, 0}, ""},
      {"Mixed_4f", {256, 160, 320, 32, 128, 128}, 0, {0, 0, 0}, {0, 0, 0}, ""},
      {"Mixed_5b", {256, 160, 320, 32, 128, 128}, 1, {2, 2, 2}, {2, 2, 2}, "MaxPool3d_5a_2x2"},
      {"Mixed_5c", {384, 192, 384, 48, 128, 128}, 0, {0, 0, 0}, {0, 0, 0}, ""},
  };
  Act cur = p3a, Gcur = Gp3a;   // block input and its gradient buffer
  int cur_c = 192;
  bool cur_is_relu = false;     // is `cur` a ReLU output (so its gradient needs the producer's mask)?
  for (const Blk& bk : blocks) {
    const std::string bn = bk.name;
    if (bk.pool_before) {
      Act po, Gpo; PoolRec pr;
      if ((rc = emit_pool_fwd(bk.pool_name, cur, cur_c, bk.pk[0], bk.pk[1], bk.pk[2], bk.ps[0], bk.ps[1], bk.ps[2], po, pr, true))) return rc;
      if ((rc = new_act(Gpo, po.T, po.H, po.W, cur_c))) return rc;
      named[bk.pool_name] = {po, cur_c};
      named[std::string("grad:") + bk.pool_name] = {Gpo, cur_c};
      const Act prev = cur, Gprev = Gcur;
      const std::string pn = bk.pool_name;
      bwd_emit.push_back([this, pn, pr, Gpo, Gprev]() { emit_pool_bwd(pn, pr, Gpo, Gprev, nullptr); });
      (void)prev;
      cur = po; Gcur = Gpo; cur_is_relu = false;
    }
    const int c0 = bk.c[0], c1a = bk.c[1], c1b = bk.c[2], c2a = bk.c[3], c2b_ = bk.c[4], c3 = bk.c[5];
    const int cout_total = c0 + c1b + c2b_ + c3;
    ConvLayer *L0, *L1a, *L1b, *L2a, *L2b, *L3;
    const std::string b2name = bn == "Mixed_5b" ? "Conv3d_0a_3x3" : "Conv3d_0b_3x3";   // i3d.py:418
    if ((rc = make_unit3d(bn + "/Branch_0/Conv3d_0a_1x1", 1, 1, 1, cur_c, c0, &L0))) return rc;
    if ((rc = make_unit3d(bn + "/Branch_1/Conv3d_0a_1x1", 1, 1, 1, cur_c, c1a, &L1a))) return rc;
    const long rows_blk = (long)B * cur.T * cur.H * cur.W;
    // Branch_1 + Branch_2 3x3x3 units as ONE launch per pass (Mixed_4* / Mixed_5* at the benchmark batch: launches of < 256 x 256 positions,
    // where Branch_2 alone runs at 40-250 TFLOP/s on a 27-step K loop): group layouts of the forward pass and of the data-gradients.
    // Until the row-ahead ring kernels (conv_igemm.hip modes 5 / 6) the Mixed_3* blocks (>= 256 x 256 positions at bs 8) were faster as two
    // launches on two streams; as RING groups they save a fork / join pair per block and pass and Branch_2's 27-step workgroups fill
    // Branch_1's tail: 5.547 -> 5.502 ms per step.
    GroupLayout gf, gb;
    constexpr long group_min_rows = 8192;      // (below: Mixed_5*, 3136 positions at T = 64 and 4704 at T = 90 -- split-K launches)
    if (rows_blk >= group_min_rows) {
      gf = plan_group(cur, c1a, c1b, bk.c[4]);
      gb = plan_group(cur, c1b, c1a, bk.c[3]);
    }
    if ((rc = make_unit3d(bn + "/Branch_1/Conv3d_0b_3x3", 3, 3, 3, c1a, c1b, &L1b)) || (rc = pack(L1b, rows_blk, gf.nf1, gb.nf1))) return rc;
    if ((rc = make_unit3d(bn + "/Branch_2/Conv3d_0a_1x1", 1, 1, 1, cur_c, c2a, &L2a))) return rc;
    // The three 1x1x1 units reading the block input (i3d.py:197-207) run as ONE GEMM [b0 | b1a | b2a]: one launch, the
    // input read once; columns [0,c0) land in the concat buffer, the rest in `mid`.  Its data-gradient is one GEMM too,
    // with K gathered from the two gradient buffers.
    ConvLayer* Lf = nullptr;
    {
      const int cf = c0 + c1a + c2a;
      auto L = std::make_unique<ConvLayer>();
      L->name = bn + "/Branch_0+1+2/Conv3d_0a_1x1"; L->kt = L->kh = L->kw = 1; L->cin = cur_c; L->cout = cf;
      L->w.resize((size_t)cur_c * cf);
      for (int ci = 0; ci < cur_c; ++ci) {
        float* d = &L->w[(size_t)ci * cf];
        memcpy(d, &L0->w[(size_t)ci * c0], c0 * sizeof(float));
        memcpy(d + c0, &L1a->w[(size_t)ci * c1a], c1a * sizeof(float));
        memcpy(d + c0 + c1a, &L2a->w[(size_t)ci * c2a], c2a * sizeof(float));
      }
      for (ConvLayer* q : {L0, L1a, L2a}) {
        L->scale.insert(L->scale.end(), q->scale.begin(), q->scale.end());
        L->bias.insert(L->bias.end(), q->bias.begin(), q->bias.end());
      }
      std::vector<float> wT((size_t)cf * cur_c);
      for (int ci = 0; ci < cur_c; ++ci)
        for (int k = 0; k < cf; ++k) wT[(size_t)k * cur_c + ci] = L->w[(size_t)ci * cf + k];
      if ((rc = flk_conv_weights_create_impl(L->w.data(), 1, 1, 1, cur_c, cf, nullptr, 0, dtype, nf_for(cf, 1, rows_blk), 0, &L->wf))) return rc;
      if ((rc = flk_conv_weights_create_impl(wT.data(), 1, 1, 1, cf, cur_c, L->scale.data(), 0, dtype, nf_for(cur_c, 1, rows_blk), c0, &L->wb))) return rc;
      if ((rc = upload(&L->d_scale, L->scale)) || (rc = upload(&L->d_bias, L->bias))) return rc;
      Lf = L.get();
      convs.push_back(std::move(L));
    }
    if ((rc = make_unit3d(bn + "/Branch_2/" + b2name, 3, 3, 3, c2a, c2b_, &L2b)) || (rc = pack(L2b, rows_blk, gf.nf2, gb.nf2))) return rc;
    if ((rc = make_unit3d(bn + "/Branch_3/Conv3d_0b_1x1", 1, 1, 1, cur_c, c3, &L3)) || (rc = pack(L3, rows_blk))) return rc;
    Act out, Gout, mid, Gmid, pl, Gpl, gxa;
    if ((rc = new_act(out, cur.T, cur.H, cur.W, cout_total)) || (rc = new_act(Gout, cur.T, cur.H, cur.W, cout_total))) return rc;
    if ((rc = new_act(mid, cur.T, cur.H, cur.W, c1a + c2a)) || (rc = new_act(Gmid, cur.T, cur.H, cur.W, c1a + c2a))) return rc;
    if ((rc = new_act(Gpl, cur.T, cur.H, cur.W, cur_c)) || (rc = new_act(gxa, cur.T, cur.H, cur.W, cur_c))) return rc;
    PoolRec pr3;
    // forward: [b0 | b1 | b2 | b3] slices of `out` (tf.concat axis 4, i3d.py:219)
    {
      flk_conv_args a{};
      a.in = cur.p; a.in_ld = cur.ld; a.cin = cur_c; a.B = B; a.Ti = cur.T; a.Hi = cur.H; a.Wi = cur.W;
      a.kt = a.kh = a.kw = 1; a.st = a.sh = a.sw = 1;
      a.To = cur.T; a.Ho = cur.H; a.Wo = cur.W; a.OT = cur.T; a.OH = cur.H; a.OW = cur.W; a.ost = a.osh = a.osw = 1;
      a.out = out.p; a.out_ld = out.ld; a.out_coff = 0; a.cout = c0 + c1a + c2a;
      a.out2 = mid.p; a.out2_ld = mid.ld; a.out2_coff = 0; a.cout1 = c0;
      a.scale = Lf->d_scale; a.bias = Lf->d_bias; a.relu = 1;
      const double macs = (double)B * cur.T * cur.H * cur.W * cur_c * (c0 + c1a + c2a);
      flk_conv_weights* wf = Lf->wf;
      const int dt = dtype;
      fwd.push_back(Op{Lf->name, K_CONV, 2.0 * macs, conv_bytes(a), [a, wf, dt](hipStream_t s) { return flk_conv3d(&a, wf, dt, s); }});
    }
    // The Branch_3 pool reads the block input only: it starts beside the fused 1x1x1 GEMM (98-392 workgroups, which leave CUs
    // idle) on side stream 2 and Branch_3's 1x1x1 follows it there.  The two 3x3x3 branches fork after the GEMM (disjoint
    // channel slices of `out`).
    Op fused = std::move(fwd.back());
    fwd.pop_back();
    push_sync(fwd, K_FORK, 2);
    {
      const size_t m0 = fwd.size();
      if ((rc = emit_pool_fwd(bn + "/Branch_3/MaxPool3d_0a_3x3", cur, cur_c, 3, 3, 3, 1, 1, 1, pl, pr3))) return rc;
      set_lane(fwd, m0, 2);
    }
    const bool grp_f = gf.nfw > 0, grp_b = gb.nfw > 0;
    fwd.push_back(std::move(fused));
    if (!grp_f) push_sync(fwd, K_FORK, 1);
    if (grp_f) {
      // Branch_1 and Branch_2 in one launch on the caller's stream; only Branch_3's pool -> 1x1x1 chain runs beside it (side stream 2)
      emit_group(fwd, bn + "/Branch_1+2/Conv3d_0b_3x3", conv_fwd_args(L1b, mid, 0, out, c0), L1b->wf, conv_fwd_args(L2b, mid, c1a, out, c0 + c1b),
                 L2b->wf, gf.nfw, gf.ring);
    } else {
      emit_conv_fwd(L1b, mid, 0, out, c0);
      { const size_t m0 = fwd.size(); emit_conv_fwd(L2b, mid, c1a, out, c0 + c1b); set_lane(fwd, m0, 1); }
    }
    { const size_t m0 = fwd.size(); emit_conv_fwd(L3, pl, 0, out, c0 + c1b + c2b_); set_lane(fwd, m0, 2); }
    push_sync(fwd, K_JOIN, grp_f ? 2 : ~0);
    named[bn] = {out, cout_total};
    named["grad:" + bn] = {Gout, cout_total};
    named["mid:" + bn] = {mid, c1a + c2a};
    named["gradmid:" + bn] = {Gmid, c1a + c2a};
    // backward (emitted in reverse program order below): branch 3 first so the 1x1 dgrads can accumulate on it
    const Act in_act = cur, Gin = Gcur;
    const bool in_relu = cur_is_relu;
    const std::string pname = bn + "/Branch_3/MaxPool3d_0a_3x3";
    // Branch_3's backward is a CHAIN of two kernels (1x1x1 data-gradient -> pool scatter) beside the two single 3x3x3 data-gradients:
    // started together, its first link is starved by the big launches (19 -> 88 us in Mixed_4c) and the scatter then runs alone on
    // the device before the join (the 1x1x1 link ahead of the fork instead, alone: measured slower, DESIGN_LOG.md).
    // bf16: Branch_3's backward (1x1x1 data-gradient -> pool scatter) as ONE kernel; fp32, the parity mode: the two-launch chain
    const bool b3_fused = dtype == FLK_BF16 && c3 % 32 == 0 && c3 <= 128;
    void* wpg = nullptr;
    if (b3_fused) {
      std::vector<float> wt((size_t)c3 * cur_c);                     // Wt[k][c] = w[c][k] * bn_scale[k]
      for (int k = 0; k < c3; ++k)
        for (int c = 0; c < cur_c; ++c) wt[(size_t)k * cur_c + c] = L3->w[(size_t)c * c3 + k] * L3->scale[k];
      if ((rc = flk_pool_gemm_weights_create(wt.data(), c3, cur_c, &wpg))) return rc;
      pool_gemm_weights.push_back(wpg);
    }
    const int cur_c_blk = cur_c;
    bwd_emit.push_back([=]() {
      push_sync(bwd, K_FORK, grp_b ? 2 : ~0);
      {
        const size_t m0 = bwd.size();
        if (b3_fused) {
          // ONE kernel: the 1x1x1 data-gradient on MFMA inside the pool's scatter backward (pool.hip: maxpool_scatter_gemm_bwd)
          const flk_pool_args pa = pr3.a;
          const void* gp = Gout.p; void* gip = gxa.p;
          const int gld = Gout.ld, gco = c0 + c1b + c2b_, gild = gxa.ld, K = c3;
          const double macs = (double)B * Gout.T * Gout.H * Gout.W * cur_c_blk * c3;
          const double bytes = ((double)B * Gout.T * Gout.H * Gout.W * (c3 + cur_c_blk)) * esz() + (double)B * Gout.T * Gout.H * Gout.W * cur_c_blk;
          bwd.push_back(Op{pname + "/grad+Conv3d_0b_1x1/dgrad", K_POOL, 2.0 * macs, bytes, [pa, gp, gld, gco, K, wpg, gip, gild](hipStream_t s) {
                             return flk_maxpool3d_bwd_gemm(&pa, gp, gld, gco, K, wpg, gip, gild, 0, FLK_BF16, s);
                           }});
        } else {
          emit_conv_bwd(L3, Gout, c0 + c1b + c2b_, Gpl, 0, nullptr, 0, 0, nullptr, 0);
          emit_pool_bwd(pname, pr3, Gpl, gxa, nullptr);
        }
        set_lane(bwd, m0, 2);
      }
      if (grp_b) {
        emit_group(bwd, bn + "/Branch_1+2/Conv3d_0b_3x3/dgrad", conv_bwd_args(L1b, Gout, c0, Gmid, 0, &mid, 0), L1b->wb,
                   conv_bwd_args(L2b, Gout, c0 + c1b, Gmid, c1a, &mid, c1a), L2b->wb, gb.nfw, gb.ring);
      } else {
        { const size_t m0 = bwd.size(); emit_conv_bwd(L2b, Gout, c0 + c1b, Gmid, c1a, nullptr, 0, 0, &mid, c1a); set_lane(bwd, m0, 1); }
        emit_conv_bwd(L1b, Gout, c0, Gmid, 0, nullptr, 0, 0, &mid, 0);
      }
      push_sync(bwd, K_JOIN, grp_b ? 2 : ~0);
      {
        flk_conv_args a{};
        a.in = Gout.p; a.in_ld = Gout.ld; a.in_coff = 0; a.cin = c0 + c1a + c2a; a.cin1 = c0;
        a.in2 = Gmid.p; a.in2_ld = Gmid.ld; a.in2_coff = 0;
        a.B = B; a.Ti = Gout.T; a.Hi = Gout.H; a.Wi = Gout.W;
        a.kt = a.kh = a.kw = 1; a.st = a.sh = a.sw = 1;
        a.To = Gin.T; a.Ho = Gin.H; a.Wo = Gin.W; a.OT = Gin.T; a.OH = Gin.H; a.OW = Gin.W; a.ost = a.osh = a.osw = 1;
        a.out = Gin.p; a.out_ld = Gin.ld; a.out_coff = 0; a.cout = Lf->cin;
        a.add = gxa.p; a.add_ld = gxa.ld; a.add_coff = 0;
        if (in_relu) { a.mask = in_act.p; a.mask_ld = in_act.ld; a.mask_coff = 0; }
        const double macs = (double)B * Gin.T * Gin.H * Gin.W * Lf->cin * (c0 + c1a + c2a);
        flk_conv_weights* wb = Lf->wb;
        const int dt = dtype;
        bwd.push_back(Op{Lf->name + "/dgrad", K_CONV, 2.0 * macs, conv_bytes(a), [a, wb, dt](hipStream_t s) { return flk_conv3d(&a, wb, dt, s); }});
      }
    });
    cur = out; Gcur = Gout; cur_c = cout_total; cur_is_relu = true;
  }

  // ---- Logits head (i3d.py:459-474) ----
  FLK_REQUIRE(cur.H == 7 && cur.W == 7 && cur.T >= 2, "I3D head: expected 7x7 spatial, T>=2 (got %dx%dx%d)", cur.T, cur.H, cur.W);
  {
    auto* fw = find("RGB/inception_i3d/Logits/Conv3d_0c_1x1/conv_3d/w", (size_t)cur_c * num_classes);
    auto* fb = find("RGB/inception_i3d/Logits/Conv3d_0c_1x1/conv_3d/b", num_classes);
    if (!fw || !fb) return FLK_EINVAL;
    if ((rc = upload(&d_fcw, *fw)) || (rc = upload(&d_fcb, *fb))) return rc;
    const int Tn = cur.T, Tp = Tn - 1;       // avg-pool 2x7x7 VALID s1 -> T' = Tn-1 frames, then mean over T'
    std::vector<float> wt(Tn);
    for (int t = 0; t < Tn; ++t) {
      const int cover = (t >= 1 ? 1 : 0) + (t <= Tn - 2 ? 1 : 0);
      wt[t] = (float)cover / (2.0f * 49.0f * (float)Tp);
    }
    if ((rc = upload(&d_wt, wt))) return rc;
    if ((rc = dmalloc((void**)&d_feat, (size_t)B * cur_c * 4)) || (rc = dmalloc((void**)&d_dfeat, (size_t)B * cur_c * 4))) return rc;
    const Act y = cur, Gy = Gcur;
    const int C = cur_c, N = num_classes, dt = dtype;
    fwd.push_back(Op{"Logits", K_HEAD, 0.0, (double)y.numel(B) * esz(), [this, y, C, N, Tn, dt](hipStream_t s) {
                       return flk_head_forward(y.p, y.ld, 0, C, B, Tn, y.H * y.W, d_wt, d_fcw, d_fcb, N, d_feat, logits_out, dt, s);
                     }});
    bwd_emit.push_back([this, y, Gy, C, N, Tn, dt]() {
      bwd.push_back(Op{"Logits/grad", K_HEAD, 0.0, 2.0 * (double)y.numel(B) * esz(), [this, y, Gy, C, N, Tn, dt](hipStream_t s) {
                         return flk_head_backward(y.p, y.ld, 0, Gy.p, Gy.ld, 0, C, B, Tn, y.H * y.W, d_wt, d_fcw, N, dlogits_in,
                                                  d_dfeat, 1, dt, s);
                       }});
    });
  }
  for (auto it = bwd_emit.rbegin(); it != bwd_emit.rend(); ++it) (*it)();
  (void)xin;
  for (size_t i = 0; i < bwd.size(); ++i)
    if (bwd[i].name == "Conv3d_1a_7x7/dgrad") stem_dgrad_op = (int)i;
  return FLK_OK;
}


// ---------------------------------------------------------------------------------------------------
// torchvision-0.5.0 VideoResNet family (r2plus1d_18 / r3d_18 / mc3_18; reference call site model.py:421).
// Channel counts that are not multiples of 8 (45, 230, 460, 921) are zero-padded in storage and weights.
static inline int pad8(int c) { return (c + 7) / 8 * 8; }

int flk_net::make_conv_tv(const std::string& wname, const std::string& bnname, int cout, int cin, int kt, int kh, int kw,
                          int st_, int sh_, int sw_, int pt_, int ph_, int pw_, ConvLayer** out) {
  auto* w = find(wname + ".weight", (size_t)cout * cin * kt * kh * kw);      // torch layout [cout][cin][kt][kh][kw]
  auto* g = find(bnname + ".weight", cout);
  auto* b = find(bnname + ".bias", cout);
  auto* mean = find(bnname + ".running_mean", cout);
  auto* var = find(bnname + ".running_var", cout);
  if (!w || !g || !b || !mean || !var) return FLK_EINVAL;
  auto L = std::make_unique<ConvLayer>();
  const int cip = pad8(cin), cop = pad8(cout);
  L->name = wname; L->kt = kt; L->kh = kh; L->kw = kw; L->cin = cip; L->cout = cop;
  L->st = st_; L->sh = sh_; L->sw = sw_; L->pt = pt_; L->ph = ph_; L->pw = pw_;
  L->w.assign((size_t)kt * kh * kw * cip * cop, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < kt * kh * kw; ++t)
        L->w[((size_t)t * cip + ci) * cop + co] = (*w)[((size_t)co * cin + ci) * kt * kh * kw + t];
  L->scale.assign(cop, 0.f); L->bias.assign(cop, 0.f);
  for (int c = 0; c < cout; ++c) {
    const float a = (*g)[c] / sqrtf((*var)[c] + 1e-5f);                      // BatchNorm3d eval, eps 1e-5
    L->scale[c] = a; L->bias[c] = (*b)[c] - (*mean)[c] * a;
  }
  *out = L.get();
  convs.push_back(std::move(L));
  return FLK_OK;
}

// forward operator + one data-gradient operator per input parity class (i = s*j + p):
//   gx[s*j + p] = sum_{k = p + pad (mod s)} G[j + (p + pad - k)/s] . (a * W[k])^T
int flk_net::pack_generic(ConvLayer* L) {
  const int taps = L->kt * L->kh * L->kw;
  int rc = flk_conv_weights_create_impl(L->w.data(), L->kt, L->kh, L->kw, L->cin, L->cout, nullptr, 0, dtype,
                                        choose_nf(L->cout, taps), 0, &L->wf);
  if (rc) return rc;
  if ((rc = upload(&L->d_scale, L->scale)) || (rc = upload(&L->d_bias, L->bias))) return rc;
  auto dim_class = [](int k, int s, int pad, int p, std::vector<int>& ks, int& off_min) {
    ks.clear(); off_min = 0;
    // taps of this class ordered by ASCENDING offset (descending k)
    for (int kk = k - 1; kk >= 0; --kk)
      if ((((p + pad - kk) % s) + s) % s == 0) ks.push_back(kk);
    if (!ks.empty()) off_min = (p + pad - ks.front()) / s;   // may be negative
  };
  for (int ct = 0; ct < L->st; ++ct)
    for (int chh = 0; chh < L->sh; ++chh)
      for (int cw = 0; cw < L->sw; ++cw) {
        std::vector<int> kts, khs, kws;
        int ot, oh, ow;
        dim_class(L->kt, L->st, L->pt, ct, kts, ot);
        dim_class(L->kh, L->sh, L->ph, chh, khs, oh);
        dim_class(L->kw, L->sw, L->pw, cw, kws, ow);
        if (kts.empty() || khs.empty() || kws.empty()) continue;       // class receives no gradient from this layer
        const int nt = (int)kts.size(), nh = (int)khs.size(), nw = (int)kws.size();
        std::vector<float> wd((size_t)nt * nh * nw * L->cout * L->cin);
        for (int a = 0; a < nt; ++a) for (int b2 = 0; b2 < nh; ++b2) for (int c = 0; c < nw; ++c) {
          const int src = (kts[a] * L->kh + khs[b2]) * L->kw + kws[c];
          for (int co = 0; co < L->cout; ++co)
            for (int ci = 0; ci < L->cin; ++ci)
              wd[((((size_t)a * nh + b2) * nw + c) * L->cout + co) * L->cin + ci] = L->w[((size_t)src * L->cin + ci) * L->cout + co];
        }
        ConvLayer::BwdClass bc{};
        rc = flk_conv_weights_create_impl(wd.data(), nt, nh, nw, L->cout, L->cin, L->scale.data(), 0, dtype,
                                          choose_nf(L->cin, nt * nh * nw), 0, &bc.w);
        if (rc) return rc;
        bc.kt = nt; bc.kh = nh; bc.kw = nw; bc.pbt = -ot; bc.pbh = -oh; bc.pbw = -ow; bc.ot = ct; bc.oh = chh; bc.ow = cw;
        L->bcls.push_back(bc);
      }
  return FLK_OK;
}

void flk_net::emit_gen_fwd(ConvLayer* L, const Act& in, const Act& out, bool relu, const Act* add) {
  flk_conv_args a{};
  a.in = in.p; a.in_ld = in.ld; a.cin = L->cin; a.B = B; a.Ti = in.T; a.Hi = in.H; a.Wi = in.W;
  a.kt = L->kt; a.kh = L->kh; a.kw = L->kw; a.st = L->st; a.sh = L->sh; a.sw = L->sw; a.pt = L->pt; a.ph = L->ph; a.pw = L->pw;
  a.To = out.T; a.Ho = out.H; a.Wo = out.W; a.OT = out.T; a.OH = out.H; a.OW = out.W; a.ost = a.osh = a.osw = 1;
  a.out = out.p; a.out_ld = out.ld; a.cout = L->cout;
  a.scale = L->d_scale; a.bias = L->d_bias; a.relu = relu;
  if (add) { a.add = add->p; a.add_ld = add->ld; }
  const double macs = (double)B * out.T * out.H * out.W * L->kt * L->kh * L->kw * L->cin * L->cout;
  flk_conv_weights* wf = L->wf;
  const int dt = dtype;
  attach_splitk(a, wf);
  fwd.push_back(Op{L->name, K_CONV, 2.0 * macs, conv_bytes(a), [a, wf, dt](hipStream_t s) { return flk_conv3d(&a, wf, dt, s); }});
}

void flk_net::emit_gen_bwd(ConvLayer* L, const Act& G, const Act& gin, const void* add, int add_ld, const Act* mask) {
  // The parity classes of a strided layer's data-gradient write disjoint output cells and are small launches on their latency floor.  The
  // EIGHT classes of a 3x3x3 / 2 layer (r3d_18 layer2-4.0) run side by side on the caller's stream and the two side streams: r3d_18 bs 8 2.91 ->
  // 2.81 ms per iteration.  With two or four classes ((3,1,1) / (2,1,1), (1,3,3) / (1,2,2): r2plus1d_18, mc3_18) the fork / join pair costs
  // more than the overlap gives (r2plus1d_18 bs 8 4.24 -> 4.37, bs 1 2.20 -> 2.36; mc3_18 the same): those stay in line.
  const bool par = L->bcls.size() >= 8u;
  if (par) push_sync(bwd, K_FORK, ~0);
  int ci = 0;
  for (const auto& bc : L->bcls) {
    flk_conv_args a{};
    a.in = G.p; a.in_ld = G.ld; a.cin = L->cout; a.B = B; a.Ti = G.T; a.Hi = G.H; a.Wi = G.W;
    a.kt = bc.kt; a.kh = bc.kh; a.kw = bc.kw; a.st = a.sh = a.sw = 1; a.pt = bc.pbt; a.ph = bc.pbh; a.pw = bc.pbw;
    a.To = (gin.T - bc.ot + L->st - 1) / L->st; a.Ho = (gin.H - bc.oh + L->sh - 1) / L->sh; a.Wo = (gin.W - bc.ow + L->sw - 1) / L->sw;
    if (a.To <= 0 || a.Ho <= 0 || a.Wo <= 0) continue;
    a.OT = gin.T; a.OH = gin.H; a.OW = gin.W; a.ost = L->st; a.osh = L->sh; a.osw = L->sw; a.oot = bc.ot; a.ooh = bc.oh; a.oow = bc.ow;
    a.out = gin.p; a.out_ld = gin.ld; a.cout = L->cin;
    a.add = add; a.add_ld = add_ld;
    if (mask) { a.mask = mask->p; a.mask_ld = mask->ld; }
    const double macs = (double)B * a.To * a.Ho * a.Wo * bc.kt * bc.kh * bc.kw * L->cin * L->cout;
    flk_conv_weights* wb = bc.w;
    const int dt = dtype;
    attach_splitk(a, wb);
    const size_t m0 = bwd.size();
    bwd.push_back(Op{L->name + "/dgrad", K_CONV, 2.0 * macs, conv_bytes(a), [a, wb, dt](hipStream_t s) { return flk_conv3d(&a, wb, dt, s); }});
    if (par) set_lane(bwd, m0, ci++ % (kSideStreams + 1));
  }
  if (par) push_sync(bwd, K_JOIN, ~0);
}

int flk_net::build_videoresnet() {
  FLK_REQUIRE(H % 2 == 0 && W % 2 == 0 && T >= 1, "VideoResNet: H and W must be even");
  const bool r21 = arch == FLK_NET_R2PLUS1D_18;
  int rc;
  std::vector<std::function<void()>> bwd_emit;
  const int H2 = H / 2, W2 = W / 2;
  auto out_dim = [](int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; };

  // ---- stem conv (kt x 7 x 7, stride 1x2x2, pad (kt-1)/2 x 3 x 3) on the (h,w) space-to-depth clip [B,T,H/2,W/2,16]:
  //      kh = 2*j + q - 1 (j = 0..3 taps, q = parity), pad-before 2 in the folded space ----
  const int skt = r21 ? 1 : 3, sc_out = r21 ? 45 : 64;
  ConvLayer* stem7 = nullptr;
  if ((rc = make_conv_tv("stem.0", "stem.1", sc_out, 3, skt, 7, 7, 1, 2, 2, (skt - 1) / 2, 3, 3, &stem7))) return rc;
  ConvLayer* stem = nullptr;
  {
    auto L = std::make_unique<ConvLayer>();
    L->name = "stem.0"; L->kt = skt; L->kh = 4; L->kw = 4; L->cin = 16; L->cout = stem7->cout;
    L->st = L->sh = L->sw = 1; L->pt = (skt - 1) / 2; L->ph = 2; L->pw = 2;
    L->w.assign((size_t)skt * 16 * 16 * L->cout, 0.f);
    for (int kt = 0; kt < skt; ++kt) for (int jh = 0; jh < 4; ++jh) for (int jw = 0; jw < 4; ++jw)
      for (int qh = 0; qh < 2; ++qh) for (int qw = 0; qw < 2; ++qw) {
        const int kh = 2 * jh + qh - 1, kw = 2 * jw + qw - 1;
        if (kh < 0 || kh > 6 || kw < 0 || kw > 6) continue;
        for (int c = 0; c < 3; ++c) {
          const float* src = &stem7->w[((((size_t)kt * 7 + kh) * 7 + kw) * stem7->cin + c) * stem7->cout];
          float* dst = &L->w[((((size_t)kt * 4 + jh) * 4 + jw) * 16 + (qh * 2 + qw) * 3 + c) * L->cout];
          for (int co = 0; co < L->cout; ++co) dst[co] = src[co];
        }
      }
    L->scale = stem7->scale; L->bias = stem7->bias;
    stem = L.get();
    convs.push_back(std::move(L));
  }
  if ((rc = pack_generic(stem))) return rc;
  // bf16: the clip arrives as TWO bf16 numbers per value (flk_apply_args.fold_t = 4: channels [0,16) = bf16(x_adv), [16,32) = the
  // remainder), both halves against the same weights -- the stem's K step had 32 channels anyway (16 padded): same MFMA work, the
  // perturbed clip to ~16 bits.  The data-gradient keeps the 16-channel operator (its output is the gradient of x_adv).
  const bool hilo = dtype == FLK_BF16;
  flk_conv_weights* stem_wf = stem->wf;
  if (hilo) {
    std::vector<float> w2((size_t)skt * 16 * 32 * stem->cout);
    for (int t = 0; t < skt * 16; ++t)
      for (int half = 0; half < 2; ++half)
        memcpy(&w2[((size_t)t * 32 + half * 16) * stem->cout], &stem->w[(size_t)t * 16 * stem->cout], (size_t)16 * stem->cout * sizeof(float));
    if ((rc = flk_conv_weights_create_impl(w2.data(), skt, 4, 4, 32, stem->cout, nullptr, 0, dtype, choose_nf(stem->cout, skt * 16), 0, &stem_hilo_w))) return rc;
    stem_wf = stem_hilo_w;
  }
  in_ch = hilo ? 32 : 16;
  Act a_st, G_st;
  if ((rc = new_act(a_st, T, H2, W2, stem->cout)) || (rc = new_act(G_st, T, H2, W2, stem->cout))) return rc;
  named[r21 ? "stem.mid" : "stem"] = {a_st, sc_out};
  {
    flk_conv_args a{};
    a.in_ld = in_ch; a.cin = in_ch; a.B = B; a.Ti = T; a.Hi = H2; a.Wi = W2;
    a.kt = skt; a.kh = a.kw = 4; a.st = a.sh = a.sw = 1; a.pt = (skt - 1) / 2; a.ph = a.pw = 2;
    a.To = T; a.Ho = H2; a.Wo = W2; a.OT = T; a.OH = H2; a.OW = W2; a.ost = a.osh = a.osw = 1;
    a.out = a_st.p; a.out_ld = a_st.ld; a.cout = stem->cout; a.scale = stem->d_scale; a.bias = stem->d_bias; a.relu = 1;
    const double macs = (double)B * T * H2 * W2 * skt * 49.0 * 3 * sc_out;
    flk_conv_weights* wf = stem_wf;
    const int dt = dtype;
    fwd.push_back(Op{"stem.0", K_CONV, 2.0 * macs, conv_bytes(a), [this, a, wf, dt](hipStream_t s) mutable {
                       a.in = x_in;
                       if (int rc = apply_slice(0, a.B, (void*)x_in, s)) return rc;
                       return flk_conv3d(&a, wf, dt, s);
                     }});
    const ConvLayer::BwdClass bc = stem->bcls[0];
    flk_conv_args g{};
    g.in = G_st.p; g.in_ld = G_st.ld; g.cin = stem->cout; g.B = B; g.Ti = T; g.Hi = H2; g.Wi = W2;
    g.kt = bc.kt; g.kh = bc.kh; g.kw = bc.kw; g.st = g.sh = g.sw = 1; g.pt = bc.pbt; g.ph = bc.pbh; g.pw = bc.pbw;
    g.To = T; g.Ho = H2; g.Wo = W2; g.OT = T; g.OH = H2; g.OW = W2; g.ost = g.osh = g.osw = 1;
    g.out_ld = 16; g.cout = 16;
    flk_conv_weights* wb = bc.w;
    bwd_emit.push_back([this, g, wb, dt, macs]() {
      bwd.push_back(Op{"stem.0/dgrad", K_CONV, 2.0 * macs, conv_bytes(g), [this, g, wb, dt](hipStream_t s) mutable { g.out = gx_in; return flk_conv3d(&g, wb, dt, s); }});
    });
  }
  Act cur = a_st, Gcur = G_st;
  if (r21) {
    ConvLayer* s3 = nullptr;
    if ((rc = make_conv_tv("stem.3", "stem.4", 64, 45, 3, 1, 1, 1, 1, 1, 1, 0, 0, &s3)) || (rc = pack_generic(s3))) return rc;
    Act a2, G2;
    if ((rc = new_act(a2, T, H2, W2, 64)) || (rc = new_act(G2, T, H2, W2, 64))) return rc;
    emit_gen_fwd(s3, cur, a2, true, nullptr);
    named["stem"] = {a2, 64};
    const Act prev = cur, Gprev = Gcur;
    bwd_emit.push_back([this, s3, G2, Gprev, prev]() { emit_gen_bwd(s3, G2, Gprev, nullptr, 0, &prev); });
    cur = a2; Gcur = G2;
  }
  named["grad:stem"] = {Gcur, 64};

  // ---- residual stages ----
  const int planes_of[4] = {64, 128, 256, 512};
  int inpl = 64;
  for (int li = 1; li <= 4; ++li) {
    const int planes = planes_of[li - 1];
    const int kind = r21 ? 2 : ((arch == FLK_NET_R3D_18 || li == 1) ? 0 : 1);   // 0: 3x3x3, 1: 1x3x3 (no temporal), 2: (2+1)D
    for (int bi = 0; bi < 2; ++bi) {
      const int stride = (li > 1 && bi == 0) ? 2 : 1;
      const bool has_ds = stride != 1 || inpl != planes;
      const std::string pre = "layer" + std::to_string(li) + "." + std::to_string(bi);
      const int mid = (inpl * planes * 27) / (inpl * 9 + 3 * planes);
      const int dst = kind == 1 ? 1 : stride;        // temporal stride of the block (and of its downsample)
      Act out, Gout;
      if ((rc = new_act(out, out_dim(cur.T, 1, dst, 0), out_dim(cur.H, 1, stride, 0), out_dim(cur.W, 1, stride, 0), planes)) ||
          (rc = new_act(Gout, out.T, out.H, out.W, planes))) return rc;
      // builds conv_builder(cin -> cout, stride s) + BN; returns the layers (1 or 2) and the intermediate tensors
      struct Unit { ConvLayer* a = nullptr; ConvLayer* b = nullptr; Act midact, Gmid; };
      auto make_unit = [&](const std::string& up, const std::string& bnp, int ci, int co, int s, Unit& u) -> int {
        int r;
        if (kind == 2) {
          if ((r = make_conv_tv(up + ".0", up + ".1", mid, ci, 1, 3, 3, 1, s, s, 0, 1, 1, &u.a)) || (r = pack_generic(u.a))) return r;
          // second conv of Conv2Plus1D has no BN of its own inside the Sequential: the block's BN (bnp) follows it
          if ((r = make_conv_tv(up + ".3", bnp, co, mid, 3, 1, 1, s, 1, 1, 1, 0, 0, &u.b)) || (r = pack_generic(u.b))) return r;
        } else if (kind == 0) {
          if ((r = make_conv_tv(up, bnp, co, ci, 3, 3, 3, s, s, s, 1, 1, 1, &u.a)) || (r = pack_generic(u.a))) return r;
        } else {
          if ((r = make_conv_tv(up, bnp, co, ci, 1, 3, 3, 1, s, s, 0, 1, 1, &u.a)) || (r = pack_generic(u.a))) return r;
        }
        return FLK_OK;
      };
      Unit u1, u2;
      if ((rc = make_unit(pre + ".conv1.0", pre + ".conv1.1", inpl, planes, stride, u1))) return rc;
      if ((rc = make_unit(pre + ".conv2.0", pre + ".conv2.1", planes, planes, 1, u2))) return rc;
      Act h1, Gh1;
      if ((rc = new_act(h1, out.T, out.H, out.W, planes)) || (rc = new_act(Gh1, out.T, out.H, out.W, planes))) return rc;
      ConvLayer* ds = nullptr;
      Act dsout;
      if (has_ds) {
        if ((rc = make_conv_tv(pre + ".downsample.0", pre + ".downsample.1", planes, inpl, 1, 1, 1, dst, stride, stride, 0, 0, 0, &ds)) ||
            (rc = pack_generic(ds))) return rc;
        if ((rc = new_act(dsout, out.T, out.H, out.W, planes))) return rc;
      }
      // forward
      if (kind == 2) {
        if ((rc = new_act(u1.midact, cur.T, out.H, out.W, u1.a->cout)) || (rc = new_act(u1.Gmid, cur.T, out.H, out.W, u1.a->cout))) return rc;
        if ((rc = new_act(u2.midact, out.T, out.H, out.W, u2.a->cout)) || (rc = new_act(u2.Gmid, out.T, out.H, out.W, u2.a->cout))) return rc;
        emit_gen_fwd(u1.a, cur, u1.midact, true, nullptr);
        emit_gen_fwd(u1.b, u1.midact, h1, true, nullptr);
        if (has_ds) emit_gen_fwd(ds, cur, dsout, false, nullptr);
        emit_gen_fwd(u2.a, h1, u2.midact, true, nullptr);
        emit_gen_fwd(u2.b, u2.midact, out, true, has_ds ? &dsout : &cur);
      } else {
        emit_gen_fwd(u1.a, cur, h1, true, nullptr);
        if (has_ds) emit_gen_fwd(ds, cur, dsout, false, nullptr);
        emit_gen_fwd(u2.a, h1, out, true, has_ds ? &dsout : &cur);
      }
      named[pre] = {out, planes};
      named["grad:" + pre] = {Gout, planes};
      // backward: G_out (masked by out > 0) -> conv2 -> h1 -> conv1 (+ shortcut) -> block input, masked by input > 0
      const Act in_act = cur, Gin = Gcur;
      const bool k2 = kind == 2;
      bwd_emit.push_back([=]() {
        if (k2) {
          emit_gen_bwd(u2.b, Gout, u2.Gmid, nullptr, 0, &u2.midact);
          emit_gen_bwd(u2.a, u2.Gmid, Gh1, nullptr, 0, &h1);
        } else {
          emit_gen_bwd(u2.a, Gout, Gh1, nullptr, 0, &h1);
        }
        // shortcut gradient first (plain), conv1's data-gradient then accumulates onto it and applies the ReLU mask
        const void* addp; int addld;
        if (has_ds) {
          const size_t bytes = Gin.numel(B) * esz();
          void* gp = Gin.p;
          bwd.push_back(Op{pre + ".downsample/zero", K_OTHER, 0.0, (double)bytes, [gp, bytes](hipStream_t s) {
                             FLK_CHECK_HIP(hipMemsetAsync(gp, 0, bytes, s));
                             return FLK_OK;
                           }});
          emit_gen_bwd(ds, Gout, Gin, nullptr, 0, nullptr);
          addp = Gin.p; addld = Gin.ld;
        } else {
          addp = Gout.p; addld = Gout.ld;
        }
        if (k2) {
          emit_gen_bwd(u1.b, Gh1, u1.Gmid, nullptr, 0, &u1.midact);
          emit_gen_bwd(u1.a, u1.Gmid, Gin, addp, addld, &in_act);
        } else {
          emit_gen_bwd(u1.a, Gh1, Gin, addp, addld, &in_act);
        }
      });
      cur = out; Gcur = Gout; inpl = planes;
    }
  }

  // ---- head: AdaptiveAvgPool3d(1) + Linear(512, num_classes) ----
  {
    auto* fw = find("fc.weight", (size_t)num_classes * 512);     // torch layout [N][C]
    auto* fb = find("fc.bias", num_classes);
    if (!fw || !fb) return FLK_EINVAL;
    std::vector<float> wT((size_t)512 * num_classes);
    for (int n = 0; n < num_classes; ++n)
      for (int c = 0; c < 512; ++c) wT[(size_t)c * num_classes + n] = (*fw)[(size_t)n * 512 + c];
    if ((rc = upload(&d_fcw, wT)) || (rc = upload(&d_fcb, *fb))) return rc;
    const int Tn = cur.T;
    std::vector<float> wt(Tn, 1.0f / (float)(cur.T * cur.H * cur.W));
    if ((rc = upload(&d_wt, wt))) return rc;
    if ((rc = dmalloc((void**)&d_feat, (size_t)B * 512 * 4)) || (rc = dmalloc((void**)&d_dfeat, (size_t)B * 512 * 4))) return rc;
    const Act y = cur, Gy = Gcur;
    const int C = 512, N = num_classes, dt = dtype;
    fwd.push_back(Op{"fc", K_HEAD, 0.0, (double)y.numel(B) * esz(), [this, y, C, N, Tn, dt](hipStream_t s) {
                       return flk_head_forward(y.p, y.ld, 0, C, B, Tn, y.H * y.W, d_wt, d_fcw, d_fcb, N, d_feat, logits_out, dt, s);
                     }});
    bwd_emit.push_back([this, y, Gy, N, Tn, dt]() {
      const int C = 512;
      bwd.push_back(Op{"fc/grad", K_HEAD, 0.0, 2.0 * (double)y.numel(B) * esz(), [this, y, Gy, C, N, Tn, dt](hipStream_t s) {
                         return flk_head_backward(y.p, y.ld, 0, Gy.p, Gy.ld, 0, C, B, Tn, y.H * y.W, d_wt, d_fcw, N, dlogits_in, d_dfeat, 1, dt, s);
                       }});
    });
  }
  for (auto it = bwd_emit.rbegin(); it != bwd_emit.rend(); ++it) (*it)();
  return FLK_OK;
}

// ---------------------------------------------------------------------------------------------------
extern "C" int flk_net_create(int arch, int dtype, int B, int T, int H, int W, int device, flk_net** out) {
  FLK_REQUIRE(out, "flk_net_create: null out");
  FLK_REQUIRE(arch == FLK_NET_I3D || arch == FLK_NET_R2PLUS1D_18 || arch == FLK_NET_R3D_18 || arch == FLK_NET_MC3_18,
              "flk_net_create: unknown arch %d", arch);
  FLK_REQUIRE(dtype == FLK_F32 || dtype == FLK_BF16, "flk_net_create: bad dtype %d", dtype);
  FLK_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0, "flk_net_create: bad dims");
  flk_net* n = new (std::nothrow) flk_net();
  if (!n) { flk_set_error("out of host memory"); return FLK_ENOMEM; }
  n->arch = arch; n->dtype = dtype; n->B = B; n->T = T; n->H = H; n->W = W; n->device = device;
  *out = n;
  return FLK_OK;
}

extern "C" int flk_net_destroy(flk_net* n) {
  if (!n) return FLK_OK;
  (void)hipSetDevice(n->device);
  for (int l = 0; l < kSideStreams; ++l) {
    if (n->side[l]) (void)hipStreamDestroy(n->side[l]);
    if (n->ev_join[l]) (void)hipEventDestroy(n->ev_join[l]);
  }
  if (n->mask_stream) (void)hipStreamDestroy(n->mask_stream);
  for (auto e : n->ev_fork) if (e) (void)hipEventDestroy(e);
  for (auto e : n->ev_main) if (e) (void)hipEventDestroy(e);
  if (n->ev_mask_fork) (void)hipEventDestroy(n->ev_mask_fork);
  if (n->ev_mask_done) (void)hipEventDestroy(n->ev_mask_done);
  flk_stem_delta_grad_weights_destroy(n->d_stem_wf);
  flk_stem_delta_grad_weights_destroy(n->d_stem_sums);
  flk_conv_weights_destroy(n->stem_u8_w);
  flk_conv_weights_destroy(n->stem_hilo_w);
  for (void* p : n->allocs) (void)hipFree(p);
  for (void* p : n->pool_gemm_weights) flk_pool_gemm_weights_destroy(p);
  for (auto& L : n->convs) {
    flk_conv_weights_destroy(L->wf); flk_conv_weights_destroy(L->wb);
    for (auto& c : L->bcls) flk_conv_weights_destroy(c.w);
  }
  for (auto& e : n->ev_fwd) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : n->ev_bwd) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete n;
  return FLK_OK;
}

extern "C" int flk_net_set_weight(flk_net* n, const char* name, const float* data, int64_t numel) {
  FLK_REQUIRE(n && name && data && numel > 0, "flk_net_set_weight: bad argument");
  FLK_REQUIRE(!n->finalized, "flk_net_set_weight: net already finalized");
  n->weights[name].assign(data, data + numel);
  return FLK_OK;
}

extern "C" int flk_net_finalize(flk_net* n) {
  FLK_REQUIRE(n && !n->finalized, "flk_net_finalize: bad state");
  FLK_CHECK_HIP(hipSetDevice(n->device));
  int rc = n->arch == FLK_NET_I3D ? n->build_i3d() : n->build_videoresnet();
  if (rc) return rc;
  n->weights.clear();
  for (auto& L : n->convs) { std::vector<float>().swap(L->w); }
  // fork/join events order device work only (nobody on the host inspects them): without the system-scope fence a record no
  // longer writes back / invalidates the caches for the host's benefit -- every kernel still releases to agent scope at its end,
  // which is what the waiting queue needs.  Measured: the idle gap at each fork and join shrinks, 7.14 -> 6.88 ms per step.
  const unsigned evf = getenv("FLK_EVENT_FLAGS") ? (unsigned)strtoul(getenv("FLK_EVENT_FLAGS"), nullptr, 0)
                                                 : (hipEventDisableTiming | hipEventDisableSystemFence);
  for (int l = 0; l < kSideStreams; ++l) {
    FLK_CHECK_HIP(hipStreamCreateWithFlags(&n->side[l], hipStreamNonBlocking));      // (stream priorities for the side lanes: measured, no gain)
    FLK_CHECK_HIP(hipEventCreateWithFlags(&n->ev_join[l], evf));
  }
  FLK_CHECK_HIP(hipStreamCreateWithFlags(&n->mask_stream, hipStreamNonBlocking));
  for (auto& e : n->ev_fork) FLK_CHECK_HIP(hipEventCreateWithFlags(&e, evf));
  for (auto& e : n->ev_main) FLK_CHECK_HIP(hipEventCreateWithFlags(&e, evf));
  FLK_CHECK_HIP(hipEventCreateWithFlags(&n->ev_mask_fork, evf));
  FLK_CHECK_HIP(hipEventCreateWithFlags(&n->ev_mask_done, evf));
  n->multi_stream = !getenv("FLK_SINGLE_STREAM");
  n->ext_events = !(getenv("FLK_EXT_EVENTS") && atoi(getenv("FLK_EXT_EVENTS")) == 0);
  FLK_CHECK_HIP(hipDeviceSynchronize());
  n->finalized = true;
  return FLK_OK;
}

extern "C" int64_t flk_net_workspace_bytes(const flk_net* n) { return n ? (int64_t)n->alloc_bytes : 0; }
extern "C" int64_t flk_net_input_numel(const flk_net* n) {
  if (!n) return 0;
  return n->arch == FLK_NET_I3D ? (int64_t)n->B * (n->T / 2) * (n->H / 2) * (n->W / 2) * 32
                                : (int64_t)n->B * n->T * (n->H / 2) * (n->W / 2) * n->in_ch;
}
extern "C" int flk_net_num_classes(const flk_net* n) { return n ? n->num_classes : 0; }
extern "C" int flk_net_input_channels(const flk_net* n) { return !n ? 0 : n->arch == FLK_NET_I3D ? 32 : n->in_ch; }
extern "C" int flk_net_input_fold(const flk_net* n) { return !n ? 0 : n->arch == FLK_NET_I3D ? 3 : n->in_ch == 32 ? 4 : 1; }

static int run_ops(flk_net* n, std::vector<Op>& ops, std::vector<std::pair<hipEvent_t, hipEvent_t>>& ev, bool& ev_valid, hipStream_t s,
                   int replace_op = -1, const std::function<int(hipStream_t)>* replacement = nullptr) {
  std::vector<const char*>& tags = &ops == &n->fwd ? n->tag_fwd : n->tag_bwd;
  if (n->profile) tags.assign(ops.size(), "");
  if (n->profile && ev.size() != ops.size()) {
    for (size_t i = ev.size(); i < ops.size(); ++i) {
      // timing-only events (hipEventDisableSystemFence, the flag's documented use): a default event's system-scope fence writes the
      // caches back / invalidates them for the host at every record and that cost lands inside the measured interval of an
      // 8-70 us kernel (measured: 73.5 -> 71.7 us per conv launch; rocprofv3's kernel time is 69.8).  hipEventReleaseToDevice
      // alone changes nothing.
      const unsigned pf = hipEventDisableSystemFence;
      hipEvent_t a, b;
      FLK_CHECK_HIP(hipEventCreateWithFlags(&a, pf));
      FLK_CHECK_HIP(hipEventCreateWithFlags(&b, pf));
      ev.push_back({a, b});
    }
  }
  // per-layer profiling runs the plan serially on the caller's stream: durations of co-running kernels would overlap
  const bool ms = n->multi_stream && n->side[0] && !n->profile;
  // Fork / join events riding on kernels (FLK_EXT_EVENTS=0: every fork / join records its event with a marker packet of its own).
  // The last operator a stream runs before another stream waits for it is launched with the fork's / join's event as its STOP event
  // (FLK_LAUNCH_KERNEL): arm[i] = event for operator i, or null; armed operators are single-launch ones (nlaunch of their last run).
  const bool ext_ev = n->ext_events;
  std::vector<hipEvent_t> arm(ops.size(), nullptr);
  std::vector<char> rode(ops.size() * (kSideStreams + 1), 0);      // [sync op][0 = fork | 1 + side stream]: its event rode on a kernel
  // A fork that follows a join directly (forward pass: block k's join, block k + 1's pool fork) needs no event of its own: the forked
  // streams wait for what the caller's stream waited for at the join (the other side streams' join events) and for the caller's stream's
  // last kernel before the join (ev_main, riding on it).  derived[i] = index of that join, or -1; dmain[i] = its ev_main.
  std::vector<int> derived(ops.size(), -1);
  std::vector<hipEvent_t> dmain(ops.size(), nullptr);
  if (ms && ext_ev) {
    int last[kSideStreams + 1];
    for (int& l : last) l = -1;
    unsigned nf = 0, nm = 0;
    int last_join = -1, main_before_join = -1;
    for (size_t i = 0; i < ops.size(); ++i) {
      const Op& op = ops[i];
      if (op.kind == K_FORK) {
        hipEvent_t e = n->ev_fork[nf++ & 1];
        if (last[0] >= 0 && ops[last[0]].nlaunch == 1 && !arm[last[0]]) { arm[last[0]] = e; rode[i * (kSideStreams + 1)] = 1; }
        else if (last[0] < 0 && last_join >= 0 && main_before_join >= 0 && ops[main_before_join].nlaunch == 1 && !arm[main_before_join]) {
          hipEvent_t em = n->ev_main[nm++ & 1];
          arm[main_before_join] = em; derived[i] = last_join; dmain[i] = em;
        }
        last[0] = -1;                                   // (an event rides on a kernel once)
        last_join = -1;
      } else if (op.kind == K_JOIN) {
        last_join = (int)i; main_before_join = last[0];
        for (int l = 0; l < kSideStreams; ++l) {
          if (!(op.mask >> l & 1)) continue;
          const int j = last[l + 1];
          if (j >= 0 && ops[j].nlaunch == 1 && !arm[j]) { arm[j] = n->ev_join[l]; rode[i * (kSideStreams + 1) + 1 + l] = 1; }
          last[l + 1] = -1;
        }
        last[0] = -1;      // the caller's stream waits here: a later fork's event must fire behind these waits, not with an earlier kernel
      } else {
        last[op.lane] = (int)i == replace_op ? -1 : (int)i;     // (the replaced operator -- the fused stem kernel -- launches several kernels)
        if (op.lane == 0) last_join = -1;                       // work on the caller's stream between the join and a fork: not "directly"
      }
    }
  }
  bool in_fork = false;
  unsigned n_fork = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    Op& op = ops[i];
    if (op.kind == K_FORK) {
      in_fork = true;
      if (ms) {
        hipEvent_t e = n->ev_fork[n_fork++ & 1];          // two forks per block: alternate the event objects
        if (derived[i] >= 0) {
          const Op& jn = ops[derived[i]];
          for (int l = 0; l < kSideStreams; ++l) {
            if (!(op.mask >> l & 1)) continue;
            FLK_CHECK_HIP(hipStreamWaitEvent(n->side[l], dmain[i], 0));
            for (int l2 = 0; l2 < kSideStreams; ++l2)
              if (l2 != l && (jn.mask >> l2 & 1)) FLK_CHECK_HIP(hipStreamWaitEvent(n->side[l], n->ev_join[l2], 0));
          }
          continue;
        }
        if (!rode[i * (kSideStreams + 1)]) FLK_CHECK_HIP(hipEventRecord(e, s));
        for (int l = 0; l < kSideStreams; ++l)
          if (op.mask >> l & 1) FLK_CHECK_HIP(hipStreamWaitEvent(n->side[l], e, 0));
      }
      continue;
    }
    if (op.kind == K_JOIN) {
      in_fork = false;
      if (ms)
        for (int l = 0; l < kSideStreams; ++l) {
          if (!(op.mask >> l & 1)) continue;              // side streams that took no part in this fork
          if (!rode[i * (kSideStreams + 1) + 1 + l]) FLK_CHECK_HIP(hipEventRecord(n->ev_join[l], n->side[l]));
          FLK_CHECK_HIP(hipStreamWaitEvent(s, n->ev_join[l], 0));
        }
      continue;
    }
    hipStream_t st = (ms && op.lane > 0) ? n->side[op.lane - 1] : s;
    // tuning pass: only the launches that run alone on the device (outside the fork/join regions) are timed in the
    // conditions they will run in; branch kernels co-run with their siblings, where the isolated optimum is not the best
    if (n->tuning) flk_conv_set_autotune(!in_fork);
    if (n->profile) { FLK_CHECK_HIP(hipEventRecord(ev[i].first, st)); flk_last_kernel_tag = ""; }
    const unsigned lc0 = flk_launch_count;
    flk_stop_event = arm[i];
    int rc = ((int)i == replace_op && replacement) ? (*replacement)(st) : op.run(st);
    const bool taken = arm[i] && !flk_stop_event;
    flk_stop_event = nullptr;
    if (rc) return rc;
    op.nlaunch = (int)(flk_launch_count - lc0);
    // armed on the strength of the PREVIOUS run's launch count: if the operator launched nothing (event never taken) or more than one
    // kernel this time (the event rode on its FIRST kernel), record it the plain way behind the last one -- the waits on it are
    // enqueued later and see the re-record
    if (arm[i] && (!taken || op.nlaunch != 1)) FLK_CHECK_HIP(hipEventRecord(arm[i], st));
    if (n->profile) { FLK_CHECK_HIP(hipEventRecord(ev[i].second, st)); tags[i] = flk_last_kernel_tag; }
  }
  ev_valid = n->profile;
  return FLK_OK;
}

extern "C" int flk_net_has_forward_flicker(const flk_net* n) {
  static const bool off = getenv("FLK_STEM_CENTER") && atoi(getenv("FLK_STEM_CENTER")) == 0;
  return n && n->finalized && n->d_stem_sums && n->d_stem_tab && !off;
}

extern "C" int flk_net_forward_flicker(flk_net* n, const void* x_in, const flk_apply_args* a, float* logits, void* stream) {
  FLK_REQUIRE(n && n->finalized && x_in && logits && a, "flk_net_forward_flicker: bad argument / not finalized");
  FLK_REQUIRE(n->d_stem_sums && n->d_stem_tab, "flk_net_forward_flicker: only the I3D plan in bf16 has the exact perturbation path");
  FLK_REQUIRE(a->center == 1 && a->T == n->T && !a->delta_dense, "flk_net_forward_flicker: the clip must have been applied with center = 1 "
              "(flicker perturbation, T = %d)", n->T);
  int rc = flk_stem_delta_bias(a, n->d_stem_sums, n->d_stem_tab, stream);
  if (rc) return rc;
  FLK_REQUIRE(!a->delta_per_clip || a->B == n->B, "flk_net_forward_flicker: per-clip perturbations for %d clips, the net has %d", a->B, n->B);
  n->cur_pos_bias = n->d_stem_tab;
  n->cur_pos_bias_bstride = a->delta_per_clip ? (int64_t)(n->T / 2) * 16 * 64 : 0;
  rc = flk_net_forward(n, x_in, logits, 1, stream);
  n->cur_pos_bias = nullptr;
  n->cur_pos_bias_bstride = 0;
  return rc;
}

extern "C" int flk_net_forward_apply(flk_net* n, const flk_apply_args* a, void* x_s2d_out, float* logits, void* stream) {
  FLK_REQUIRE(n && n->finalized && a && x_s2d_out && logits, "flk_net_forward_apply: bad argument / not finalized");
  FLK_REQUIRE(a->B == n->B && a->T == n->T && a->H == n->H && a->W == n->W, "flk_net_forward_apply: apply args (%d,%d,%d,%d) do not match the "
              "net (%d,%d,%d,%d)", a->B, a->T, a->H, a->W, n->B, n->T, n->H, n->W);
  FLK_REQUIRE(a->fold_t == (n->arch == FLK_NET_I3D ? 3 : n->in_ch == 32 ? 4 : 1), "flk_net_forward_apply: fold_t %d is not this plan's input layout", a->fold_t);
  n->cur_apply = a;
  const int rc = a->center ? flk_net_forward_flicker(n, x_s2d_out, a, logits, stream) : flk_net_forward(n, x_s2d_out, logits, 1, stream);
  n->cur_apply = nullptr;
  return rc;
}

extern "C" int flk_net_forward(flk_net* n, const void* x_in, float* logits, int save_for_backward, void* stream) {
  FLK_REQUIRE(n && n->finalized && x_in && logits, "flk_net_forward: bad argument / not finalized");
  (void)save_for_backward;   // every activation lives in its own resident buffer
  n->x_in = x_in; n->logits_out = logits;
  int rc = run_ops(n, n->fwd, n->ev_fwd, n->ev_fwd_valid, (hipStream_t)stream);
  if (rc) return rc;
  n->fwd_done = true;
  return FLK_OK;
}

extern "C" int flk_net_backward(flk_net* n, const float* dlogits, void* gx_in, void* stream) {
  FLK_REQUIRE(n && n->finalized && dlogits && gx_in, "flk_net_backward: bad argument / not finalized");
  if (!n->fwd_done) { flk_set_error("flk_net_backward: no forward pass to differentiate"); return FLK_ESTATE; }
  n->dlogits_in = dlogits; n->gx_in = gx_in;
  return run_ops(n, n->bwd, n->ev_bwd, n->ev_bwd_valid, (hipStream_t)stream);
}

extern "C" int flk_net_has_backward_delta(const flk_net* n) {
  static const bool off = getenv("FLK_STEM_FUSED") && atoi(getenv("FLK_STEM_FUSED")) == 0;
  return n && n->finalized && n->d_stem_wf && n->stem_dgrad_op >= 0 && !off;
}

// do two argument sets define the same clip mask? (field by field: the struct has padding)
static bool same_mask_args(const flk_apply_args& p, const flk_apply_args& q) {
  return p.x == q.x && p.x_is_u8 == q.x_is_u8 && p.x_scale == q.x_scale && p.x_bias == q.x_bias && p.delta == q.delta &&
         p.delta_dense == q.delta_dense && p.dclip == q.dclip && p.inv_std[0] == q.inv_std[0] && p.inv_std[1] == q.inv_std[1] &&
         p.inv_std[2] == q.inv_std[2] && p.lo == q.lo && p.hi == q.hi && p.adv_flag == q.adv_flag && p.shift_x == q.shift_x &&
         p.shift_p == q.shift_p && p.B == q.B && p.T == q.T && p.H == q.H && p.W == q.W && p.delta_per_clip == q.delta_per_clip &&
         p.dclip_dev == q.dclip_dev;
}

// backward to the flickering perturbation: the stem's data-gradient op is replaced by the fused delta-gradient kernel
// (stem_grad.hip), which runs in its place in the plan (same stream, same profile slot "Conv3d_1a_7x7/dgrad")
extern "C" int flk_net_backward_delta(flk_net* n, const float* dlogits, const flk_apply_args* a, float* gdelta, float* partials, void* stream) {
  FLK_REQUIRE(n && n->finalized && dlogits && a && gdelta && partials, "flk_net_backward_delta: bad argument / not finalized");
  FLK_REQUIRE(n->d_stem_wf && n->stem_dgrad_op >= 0, "flk_net_backward_delta: only the I3D plan in bf16 has the fused stem delta-gradient");
  FLK_REQUIRE(a->B == n->B && a->T == n->T && a->H == n->H && a->W == n->W, "flk_net_backward_delta: apply args (%d,%d,%d,%d) do not match the "
              "net (%d,%d,%d,%d)", a->B, a->T, a->H, a->W, n->B, n->T, n->H, n->W);
  if (!n->fwd_done) { flk_set_error("flk_net_backward_delta: no forward pass to differentiate"); return FLK_ESTATE; }
  n->dlogits_in = dlogits; n->gx_in = nullptr;
  const flk_apply_args ac = *a;
  hipStream_t s = (hipStream_t)stream;
  // the clip mask depends on the clip and on delta only: it runs on a side stream beside the whole backward pass and is joined
  // right in front of the GEMM that consumes it (serial mode: inline)
  // (per-layer profiling is serial: the mask pre-pass then runs inline, inside the stem slot it belongs to, instead of stretching the
  // first backward operators it would otherwise run beside)
  // (its own stream: on side stream 0 -- an in-order queue -- the 0.19 ms pre-pass held back Branch_2's first data-gradients, and with
  // them the first joins of the backward pass, profiles/r3m_timeline.txt: 6.82 -> 6.79 ms per step)
  const bool beside = n->multi_stream && n->mask_stream && !n->profile;
  // (flk_net_prepare_backward_delta with the same arguments and scratch: the mask has been on its way since before the forward pass)
  const bool prepared = n->premask && beside && n->premask_scratch == partials && same_mask_args(n->premask_args, ac);
  n->premask = false;
  if (beside && !prepared) {
    FLK_CHECK_HIP(hipEventRecord(n->ev_mask_fork, s));
    FLK_CHECK_HIP(hipStreamWaitEvent(n->mask_stream, n->ev_mask_fork, 0));
    int rc = flk_stem_delta_grad_mask(&ac, partials, n->mask_stream);
    if (rc) return rc;
    FLK_CHECK_HIP(hipEventRecord(n->ev_mask_done, n->mask_stream));
  }
  // the GEMM per half of the batch on the halves' own streams (multi-stream runs with the mask on its side stream only; as one launch
  // behind the join: 5.567 against 5.536 ms per step)
  const bool halves = beside && n->stem_halves == 2;
  const int nb_half = n->B / 2;
  if (halves)
    n->delta_part = [n, ac, partials](int b0, int nb, hipStream_t st) {
      FLK_CHECK_HIP(hipStreamWaitEvent(st, n->ev_mask_done, 0));
      return flk_stem_delta_grad_part(&ac, b0, nb, n->stem_G.p, n->stem_G.ld, n->d_stem_wf, partials, st);
    };
  const std::function<int(hipStream_t)> fused = [n, ac, gdelta, partials, beside, halves, nb_half](hipStream_t st) {
    if (halves) return flk_stem_delta_grad_finish(&ac, nb_half, partials, gdelta, st);
    if (beside) FLK_CHECK_HIP(hipStreamWaitEvent(st, n->ev_mask_done, 0));
    return flk_stem_delta_grad(&ac, n->stem_G.p, n->stem_G.ld, n->d_stem_wf, gdelta, partials, beside ? 1 : 0, st);
  };
  const int rc = run_ops(n, n->bwd, n->ev_bwd, n->ev_bwd_valid, s, n->stem_dgrad_op, &fused);
  n->delta_part = nullptr;
  return rc;
}

// Optional: start the clip-mask pre-pass of the coming flk_net_backward_delta(a, scratch) NOW, on the net's own side stream -- called
// before the forward pass it runs beside the MFMA-bound stem instead of beside the first (small, latency-bound) kernels of the backward
// pass, which it slowed down (profiles/r3z_timeline.txt: the head's backward 37 us instead of 8).  The mask depends on the clip and on
// delta only.  A no-op when the plan runs serially; flk_net_backward_delta falls back to its own pre-pass when the arguments differ.
extern "C" int flk_net_prepare_backward_delta(flk_net* n, const flk_apply_args* a, float* scratch, void* stream) {
  FLK_REQUIRE(n && n->finalized && a && scratch, "flk_net_prepare_backward_delta: bad argument / not finalized");
  FLK_REQUIRE(n->d_stem_wf && n->stem_dgrad_op >= 0, "flk_net_prepare_backward_delta: only the I3D plan in bf16 has the fused stem delta-gradient");
  FLK_REQUIRE(a->B == n->B && a->T == n->T && a->H == n->H && a->W == n->W, "flk_net_prepare_backward_delta: apply args (%d,%d,%d,%d) do not match "
              "the net (%d,%d,%d,%d)", a->B, a->T, a->H, a->W, n->B, n->T, n->H, n->W);
  n->premask = false;
  if (!(n->multi_stream && n->mask_stream && !n->profile)) return FLK_OK;
  hipStream_t s = (hipStream_t)stream;
  FLK_CHECK_HIP(hipEventRecord(n->ev_mask_fork, s));          // behind everything queued so far: the previous update of delta, the previous GEMM's reads
  FLK_CHECK_HIP(hipStreamWaitEvent(n->mask_stream, n->ev_mask_fork, 0));
  if (int rc = flk_stem_delta_grad_mask(a, scratch, n->mask_stream)) return rc;
  FLK_CHECK_HIP(hipEventRecord(n->ev_mask_done, n->mask_stream));
  n->premask = true; n->premask_scratch = scratch; n->premask_args = *a;
  return FLK_OK;
}

// one serial forward + backward with conv autotuning switched on (conv_igemm.hip): every convolution of the plan times its
// candidate launch layouts on its real operands and keeps the fastest for all later calls
extern "C" int flk_net_autotune(flk_net* n, const void* x_in, float* logits, const float* dlogits, void* gx_in, void* stream) {
  FLK_REQUIRE(n && n->finalized && x_in && logits && dlogits && gx_in, "flk_net_autotune: bad argument / not finalized");
  const bool ms = n->multi_stream;
  n->multi_stream = false;
  n->tuning = true;
  int rc = flk_net_forward(n, x_in, logits, 1, stream);
  if (!rc) rc = flk_net_backward(n, dlogits, gx_in, stream);
  n->tuning = false;
  flk_conv_set_autotune(0);
  n->multi_stream = ms;
  if (rc) return rc;
  FLK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return FLK_OK;
}

extern "C" int flk_net_profile(flk_net* n, int enable) {
  FLK_REQUIRE(n, "flk_net_profile: null net");
  n->profile = enable != 0;
  if (!enable) { n->ev_fwd_valid = n->ev_bwd_valid = false; }
  return FLK_OK;
}

extern "C" int flk_net_profile_read(flk_net* n, char* json_out, int64_t cap) {
  FLK_REQUIRE(n && json_out && cap > 2, "flk_net_profile_read: bad argument");
  std::string js = "[";
  auto dump = [&](std::vector<Op>& ops, std::vector<std::pair<hipEvent_t, hipEvent_t>>& ev, bool valid, const char* pass) -> int {
    if (!valid) return FLK_OK;
    const std::vector<const char*>& tags = &ops == &n->fwd ? n->tag_fwd : n->tag_bwd;
    for (size_t i = 0; i < ops.size(); ++i) {
      if (ops[i].kind == K_FORK || ops[i].kind == K_JOIN) continue;
      if (ops[i].nlaunch == 0) continue;                  // (operators that had nothing to do in this run: the half-batch delta-gradient slots)
      FLK_CHECK_HIP(hipEventSynchronize(ev[i].second));
      float ms = 0.f;
      FLK_CHECK_HIP(hipEventElapsedTime(&ms, ev[i].first, ev[i].second));
      char buf[512];
      snprintf(buf, sizeof(buf), "%s{\"name\":\"%s\",\"pass\":\"%s\",\"kind\":\"%s\",\"kernel\":\"%s\",\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
               js.size() > 1 ? "," : "", ops[i].name.c_str(), pass, kKindName[ops[i].kind], i < tags.size() ? tags[i] : "", ms, ops[i].flops, ops[i].bytes);
      js += buf;
    }
    return FLK_OK;
  };
  int rc = dump(n->fwd, n->ev_fwd, n->ev_fwd_valid, "fwd");
  if (rc) return rc;
  rc = dump(n->bwd, n->ev_bwd, n->ev_bwd_valid, "bwd");
  if (rc) return rc;
  js += "]";
  FLK_REQUIRE((int64_t)js.size() + 1 <= cap, "flk_net_profile_read: buffer too small (%zu needed)", js.size() + 1);
  memcpy(json_out, js.c_str(), js.size() + 1);
  return FLK_OK;
}

extern "C" int flk_net_get_activation(flk_net* n, const char* name, float* host_out, int64_t cap_numel, int64_t* dims5) {
  FLK_REQUIRE(n && n->finalized && name && dims5, "flk_net_get_activation: bad argument");
  auto it = n->named.find(name);
  FLK_REQUIRE(it != n->named.end(), "flk_net_get_activation: unknown endpoint '%s'", name);
  const Act& a = it->second.first;
  const int C = it->second.second;
  dims5[0] = n->B; dims5[1] = a.T; dims5[2] = a.H; dims5[3] = a.W; dims5[4] = C;
  const size_t npos = (size_t)n->B * a.T * a.H * a.W;
  if (!host_out) return FLK_OK;
  FLK_REQUIRE((int64_t)(npos * C) <= cap_numel, "flk_net_get_activation: buffer too small");
  FLK_CHECK_HIP(hipDeviceSynchronize());
  const size_t esz = n->esz();
  std::vector<char> tmp(npos * a.ld * esz);
  FLK_CHECK_HIP(hipMemcpy(tmp.data(), a.p, tmp.size(), hipMemcpyDeviceToHost));
  for (size_t p = 0; p < npos; ++p)
    for (int c = 0; c < C; ++c) {
      float v;
      if (n->dtype == FLK_BF16) {
        const uint32_t u = (uint32_t)((const uint16_t*)tmp.data())[p * a.ld + c] << 16;
        memcpy(&v, &u, 4);
      } else {
        v = ((const float*)tmp.data())[p * a.ld + c];
      }
      host_out[p * C + c] = v;
    }
  return FLK_OK;
}

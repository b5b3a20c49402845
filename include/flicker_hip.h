/* flicker_hip.h -- C ABI of libflicker_hip.so (MI355X / gfx950 flickering-attack hot path).
 *
 * The reference (roiponytch/Flickering_Adversarial_Video) has no FFI / operator layer: its hot path is
 * framework ops (TF-1.15 Conv3D / MaxPool3D / ... and torch-1.4 aten::conv3d / ...) reached from
 * Python objects.  This header is the boundary a maintainer binds instead; every
 * entry point cites the reference code it replaces.  See INTEGRATION.md for the ctypes binding.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns 0 (FLK_OK) or a negative
 *     FLK_E* code; flk_last_error() returns a thread-local message for the last failure.
 *   - the CALLER owns every tensor: functions take raw device pointers, PODs and a hipStream_t
 *     (passed as void*); everything is asynchronous on that stream, there are no hidden syncs.
 *   - tensors are channels-last ("NDHWC") as the I3D maths is specified (i3d.py); `ld` is the
 *     channel stride between consecutive positions (>= C), `coff` a channel offset into the buffer
 *     (so Inception branches write slices of one concat buffer -- tf.concat, i3d.py:219, is elided).
 *   - dtype: FLK_F32 = fp32 storage + exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): the parity mode;
 *            FLK_BF16 = bf16 storage + bf16 MFMA with fp32 accumulation: the performance mode.
 *     BN scale/bias, losses, delta, Adam state and all reductions are fp32 in both modes.
 *   - a flk_net is bound to one device and is not thread-safe; distinct nets are independent.
 */
#ifndef FLICKER_HIP_H
#define FLICKER_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FLK_OK 0
#define FLK_EINVAL (-1)   /* bad argument / unsupported shape */
#define FLK_EHIP (-2)     /* a HIP runtime call failed */
#define FLK_ENOMEM (-3)
#define FLK_ESTATE (-4)   /* call order violated (e.g. backward before forward) */

#define FLK_F32 0
#define FLK_BF16 1

int flk_version(void);
const char* flk_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Per-op entry points (kernel-level parity tests call these).
 * ------------------------------------------------------------------------------------------- */

/* Packed weights for one convolution: built once on the host, resident on the device.
 * w_dhwio: fp32 host array [kt][kh][kw][cin][cout] (TF checkpoint layout, i3d.py:61-65).
 * row_scale (optional, len cout for transpose=1): folded into the weights (BN scale for dgrad).
 * transpose=1 builds the data-gradient operator (taps flipped, cin<->cout swapped), i.e. what
 * Conv3DBackpropInputV2 / aten::conv3d backward-input applies.  nf = output-channel fragments per
 * workgroup tile (2, 4 or 8 -> 32/64/128 channels). */
typedef struct flk_conv_weights flk_conv_weights;
int flk_conv_weights_create(const float* w_dhwio, int kt, int kh, int kw, int cin, int cout,
                            const float* row_scale, int transpose, int dtype, int nf,
                            flk_conv_weights** out);
/* Same, for an operator whose INPUT channels come from two tensors (flk_conv_args.in / .in2): channels [0,cin_split)
 * and [cin_split,cin) are each padded to a whole 64-byte slab in the packed K order.  w_dhwio is the plain
 * [kt][kh][kw][cin][cout] array (no transpose option: build the data-gradient operator explicitly). */
int flk_conv_weights_create_split(const float* w_dhwio, int kt, int kh, int kw, int cin, int cout,
                                  const float* row_scale, int cin_split, int dtype, int nf,
                                  flk_conv_weights** out);
/* Launch-layout autotuning (speed only: the arithmetic per output does not depend on the layout).  While switched on
 * (per host thread), every flk_conv3d call with a geometry not seen before times its candidate layouts on its own
 * operands, SYNCHRONISING the stream, and remembers the fastest in the weights object; later calls reuse it.
 * flk_net_autotune runs one forward + backward of a finalized network in that mode. */
int flk_conv_set_autotune(int on);

/* Weights of the folded 7x7x7 / stride-2 I3D stem (Conv3d_1a_7x7, i3d.py:168-170) as a 4x4x4 convolution over the
 * fold_t = 3 space-to-depth clip (flk_perturb_apply_s2d): w_folded is [4,4,4,32,cout] with channel
 * (qt*2+qh)*8 + qw*3 + c.  Tap index 3 of an axis exists for parity 0 only, so whole 8-channel chunks are structurally
 * zero (checked); in bf16 the K loop is assembled from the non-zero chunks only (49 steps instead of 64).  Use with
 * flk_conv3d(kt = kh = kw = 4, cin = 32, stride 1). */
int flk_conv_weights_create_s2d_stem(const float* w_folded, int cout, int dtype, int nf, flk_conv_weights** out);
int flk_conv_weights_destroy(flk_conv_weights* w);

/* Generic 3-D convolution as implicit GEMM on MFMA, LDS-staged T x H x W halo tiles.
 * Replaces snt.Conv3D + snt.BatchNorm(inference) + relu (Unit3D, i3d.py:51-71) and, with a
 * transposed weight set, Conv3DBackpropInputV2 + ReluGrad; also aten::conv3d(+BatchNorm3d+ReLU
 * +residual) of torchvision VideoResNet (model.py:421).
 *   out[pos, coff_out + n] = epilogue( sum_{tap,c} in[pos*stride - pad + tap, coff_in + c] * W )
 *   epilogue: v = acc*scale[n] + bias[n];  v += add[pos,n];  relu;  v = mask[pos,n] > 0 ? v : 0
 * (each stage optional).  Logical output grid (To,Ho,Wo) maps to physical output positions
 * o*ostride + ooffset inside (OT,OH,OW): stride-1 everywhere except the parity-decomposed
 * data-gradient of strided convolutions. */
typedef struct {
  const void* in; int in_ld, in_coff, cin;
  int B, Ti, Hi, Wi;
  int kt, kh, kw;          /* tap box */
  int st, sh, sw;          /* input step per logical output step */
  int pt, ph, pw;          /* padding BEFORE (TF SAME puts the extra pad after, i3d.py:64) */
  int To, Ho, Wo;          /* logical output grid */
  void* out; int out_ld, out_coff, cout;
  int OT, OH, OW;          /* physical output dims */
  int ost, osh, osw, oot, ooh, oow;
  const float* scale; const float* bias;       /* per output channel or NULL */
  const void* add; int add_ld, add_coff;       /* NULL or tensor with the physical output geometry */
  const void* mask; int mask_ld, mask_coff;    /* NULL or tensor with the physical output geometry */
  int relu;
  /* optional second segments (fused Inception 1x1x1 convolutions: i3d.py:197-207 share one input):
   * in2 != NULL : input channels [cin1, cin) are read from in2[:, in2_coff + (c - cin1)] (weights packed with
   *               flk_conv_weights_create_split(cin_split = cin1));
   * out2 != NULL: output channels [cout1, cout) are written to out2[:, out2_coff + (n - cout1)] (cout1 % 8 == 0). */
  const void* in2; int in2_ld, in2_coff, cin1;
  void* out2; int out2_ld, out2_coff, cout1;
  /* optional position-class bias (the exact perturbation path of the I3D stem, flk_stem_delta_bias): fp32 [To][4][4][cout], added
   * after scale / bias; row = (ot, class(oh), class(ow)) with class(o) = 0 for o == 0, 2 for o == n-2, 3 for o == n-1, else 1 */
  const float* pos_bias;
  /* optional workspace for deterministic split-K (launches with too few output tiles to fill the chip: the input-channel slabs are
   * divided over up to 8 slices, each writes fp32 partial sums here, a second launch adds them in slice order and runs the
   * epilogue -- bitwise reproducible).  NULL: never split.  Size: flk_conv_splitk_bytes (0 = this convolution is never split).
   * Two convolutions that may run concurrently need separate workspaces. */
  void* splitk_ws; int64_t splitk_ws_bytes;
  int64_t pos_bias_bstride;  /* elements between the position-class tables of consecutive clips (per-clip perturbations); 0: one table */
} flk_conv_args;
int flk_conv3d(const flk_conv_args* a, const flk_conv_weights* w, int dtype, void* stream);
int64_t flk_conv_splitk_bytes(const flk_conv_args* a, const flk_conv_weights* w);
/* n <= 3 multi-tap convolutions in ONE grid (bf16): an Inception block's Branch_1 and Branch_2 3x3x3 units (i3d.py:201-209), forward or
 * data-gradient, which the reference's graph runs as independent ops.  Member i is computed exactly as flk_conv3d(a[i], w[i]) would with
 * direct-A weights and (w[i]'s nf) / nfw waves along the channels -- bitwise the same outputs -- but its workgroups share the launch: no
 * second stream, no fork / join, and the short member fills the tail of the long one.  nfw (2 or 4) = channel fragments per wave, the
 * one template argument the members share; every w[i] must have been packed with nf in {nfw, 2 nfw, 4 nfw}.  ring != 0: the members
 * take the LDS weight ring instead (the large launches' path): all packed with nf == nfw (4 or 8), 256-row tiles.  Longest K loops first. */
int flk_conv3d_group(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int nfw, int ring, int dtype, void* stream);
/* n <= 3 convolutions of 3x3x3 taps, stride 1, bf16, weights packed with nf = 4 (i3d.py:183-186 Conv3d_2c_3x3; 200-209 / 229-238 Branch_1 and
 * Branch_2 of the Mixed_3* blocks; forward and data-gradient) in ONE persistent launch with wave-specialised producers (csrc/conv_pc.hip: one
 * 512-thread workgroup per CU; four consumer waves that only read fragments and issue MFMAs, two waves streaming the weights by LDS-DMA, two
 * staging the next halo).  Bitwise the outputs of flk_conv3d.  Members in the order given, longest K loops first.
 * flk_conv3d_pc_why_not: NULL when flk_conv3d_pc takes the convolution, else the reason (a static string). */
int flk_conv3d_pc(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int dtype, void* stream);
const char* flk_conv3d_pc_why_not(const flk_conv_args* a, const flk_conv_weights* w, int dtype);
/* 1 when a plan should send these convolutions to flk_conv3d_pc: eligible, at least two rounds of items per workgroup and a modelled
 * efficiency >= 0.8 (the large layers at the benchmark batch); flk_conv3d and flk_conv3d_group route by it themselves */
int flk_conv3d_pc_worthwhile(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int dtype);
/* the same decision from the geometries alone (no weights, no device), with the launch behind it: member 0's tile (tile3[3] = Tt, Ht, Wt), the
 * position fragments per consumer wave, the modelled efficiency and the K steps of the busiest workgroup; returns 1 / 0 or a negative FLK_E* */
int flk_conv3d_pc_query(const flk_conv_args* const* a, int n, int dtype, int* tile3, int* ni, double* eff, double* steps);
/* the validation and planning of flk_conv3d_group without the launch (FLK_OK / the error the launch would return): plan builders call it
 * once per group when the plan is built */
int flk_conv3d_group_check(const flk_conv_args* const* a, const flk_conv_weights* const* w, int n, int nfw, int ring, int dtype);
/* the (waves along the channels, weight path: 0 LDS ring / 5 LDS ring with the weights a row of taps ahead / 1 direct A / ...) flk_conv3d's heuristics choose for this geometry when the
 * weights are packed at `nf` (force_da: -1 heuristic, 0 ring, 1 direct A): lets a plan builder pick the packing of a grouped launch's
 * members before any weights exist.  No device work. */
int flk_conv_layout_query(const flk_conv_args* a, int nf, int dtype, int force_da, int* wn_out, int* mode_out);

/* tf.nn.max_pool3d SAME (i3d.py:174,189,212,252,398): padded cells never win; argmax = FIRST
 * maximum in (t,h,w) scan order, stored as a uint8 window index for the backward pass.
 * MaxPool3DGrad: gather form (bitwise reproducible) for strided windows; stride-1 odd windows (the Inception branch-3 pool)
 * use an LDS scatter with float atomics (<= kt*kh*kw fp32 additions per cell in arbitrary order; FLK_POOL_GATHER=1 selects
 * the reproducible gather form).  Optional relu-mask like flk_conv3d. */
typedef struct {
  const void* in; int in_ld, in_coff; int C;
  int B, Ti, Hi, Wi;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int To, Ho, Wo;
  void* out; int out_ld, out_coff;
  uint8_t* idx;            /* [B,To,Ho,Wo,C]; 255 = "no cell" */
  int relu_input;          /* forward only.  1: the input is a ReLU output whose gradient is masked by (input > 0)
                              anyway -> windows whose maximum is <= 0 record "no cell", so the backward pass needs no
                              mask tensor (pass mask = NULL): identical result, ~40 % less traffic */
} flk_pool_args;
int flk_maxpool3d_fwd(const flk_pool_args* a, int dtype, void* stream);
/* gin[pos,c] = (add?add:0) + sum_{windows containing pos with argmax == pos} gout[window,c];
 * then masked by mask[pos,c] > 0 if mask != NULL.  a->in/out describe the FORWARD tensors'
 * geometry: gout has the `out` geometry (ld/coff given here), gin the `in` geometry. */
int flk_maxpool3d_bwd(const flk_pool_args* a, const void* gout, int gout_ld, int gout_coff,
                      void* gin, int gin_ld, int gin_coff,
                      const void* mask, int mask_ld, int mask_coff, int dtype, void* stream);

/* Branch_3 backward of an Inception block in ONE kernel (i3d.py:211-216 backward; bf16 only): the data-gradient of the 1x1x1 unit that
 * follows the stride-1 3x3x3 max-pool is computed on MFMA inside the pool's scatter backward,
 *   gin[cell, c] = sum_{windows w whose saved argmax for channel c is cell} ( sum_k g[w, k] * Wt[k, c] ),
 * instead of flk_conv3d (transposed 1x1x1) -> HBM -> flk_maxpool3d_bwd.  `a` describes the FORWARD pool (a->C = C channels, a->idx its
 * argmax bytes); g: [B,To,Ho,Wo,g_ld] gradient of the unit's pre-ReLU output (K = 32, 64, 96 or 128 channels at g_coff); wpack: from
 * flk_pool_gemm_weights_create(Wt [K][C] = unit weight transposed x batch-norm scale).  Products stay in fp32 (no bf16 rounding of the
 * intermediate), the scatter sums in 32-bit fixed point like flk_maxpool3d_bwd: bitwise reproducible. */
int flk_pool_gemm_weights_create(const float* wt_kc, int K, int C, void** out_dev);
int flk_pool_gemm_weights_destroy(void* dev);
int flk_maxpool3d_bwd_gemm(const flk_pool_args* a, const void* g, int g_ld, int g_coff, int K, const void* wpack,
                           void* gin, int gin_ld, int gin_coff, int dtype, void* stream);

/* Perturbation apply fused with the stem's space-to-depth staging.
 * kinetics_i3d_utils.py:100-142:  x_adv = clip(x + a * clip(delta[t,c], +-dclip), lo, hi)
 * model.py:80-101 (torch dialect): same with delta/std[c] and scalar clamp bounds.
 * x: uint8 (x = u8*x_scale + x_bias, the TFRecord path pre_process_rgb_flow.py:226-234) or fp32,
 *    [B,T,H,W,3]; delta: fp32 [T,3] (flicker) or [T,H,W,3] (dense, "L12" baseline).
 * out: [B,T/2,H/2,W/2,32] of dtype: channel (qt*4+qh*2+qw)*3+c, channels 24..31 zero -- the
 * 7x7x7/2 stem (i3d.py:169) then runs as a 4x4x4/1 convolution on MFMA.  T,H,W must be even. */
typedef struct {
  const void* x; int x_is_u8; float x_scale, x_bias;
  const float* delta; int delta_dense;
  float dclip;               /* 0.4 (TF) or dynamic_max_norm (torch); <=0 disables */
  float inv_std[3];          /* 1 (TF) or 1/DEFAULT_STD (torch) */
  float lo, hi;              /* -1,1 (TF) or -1.73488, 2.49020 (torch) */
  float adv_flag;            /* 0 -> clean input */
  int shift_x, shift_p;      /* cyclic temporal rolls (kinetics_i3d_utils.py:115-137), 0 = off */
  int B, T, H, W;
  int fold_t;                /* 0/2: fold (t,h,w) parities -> [B,T/2,H/2,W/2,32], channel (qt*4+qh*2+qw)*3+c, 24..31 zero
                                (I3D stem, stride 2x2x2);
                                3: the same fold with chunk-aligned channels (qt*2+qh)*8 + qw*3+c, 6 and 7 of every 8
                                zero (one (qt,qh) parity per 16-byte chunk: flk_conv_weights_create_s2d_stem);
                                1: fold (h,w) only -> [B,T,H/2,W/2,16], channel (qh*2+qw)*3+c, 12..15 zero
                                (VideoResNet stems, stride 1x2x2);
                                4 (bf16 output only): the fold of 1 with every value as TWO bf16 numbers -> [B,T,H/2,W/2,32]: channel k =
                                bf16(x_adv), channel 16 + k = bf16(x_adv - channel k) -- the input of the VideoResNet stems in bf16
                                plans (both halves against the same weights: the perturbed clip to ~16 bits at no extra MFMA work;
                                one bf16 per value swallows |delta| < 1e-3 / std).  Gradients come back in the layout of 1. */
  int center;                /* 1 (flicker delta only): write x' = x_adv - a*p' instead of x_adv, i.e. the CLEAN value wherever the
                                clip is inactive (exactly representable in bf16 for uint8 clips) -- the perturbation then reaches the stem
                                through flk_conv_args.pos_bias in fp32 (flk_stem_delta_bias) instead of being rounded away with the
                                bf16 input: |delta| < 2^-9 is below half a bf16 ulp of a pixel value near +-1 */
  int delta_per_clip;        /* 1 (flicker delta only): delta is [B,T,3] and clip b is perturbed by ITS OWN delta[b] -- B independent
                                single-video attacks (i3d_adversarial_main_single_video_npy.py:103-337, model.py:791-982) advancing in
                                one batch; the delta-gradient then comes back per clip, [B,T,3].  0: one delta [T,3] shared by the batch */
  const float* dclip_dev;    /* delta_per_clip only, or NULL: per-clip clamp bounds [B] on the device replacing `dclip` -- the torch loop grows a
                                video's bound by 1.3 when it restarts (model.py:1061-1066), independently per video */
} flk_apply_args;
int flk_perturb_apply_s2d(const flk_apply_args* a, void* out, int dtype, void* stream);

/* d(loss)/d(delta): reduce the stem's input gradient (space-to-depth layout, from flk_conv3d) over
 * (B,H,W) with the clip masks of flk_perturb_apply_s2d.  gdelta: fp32 [T,3]
 * (flicker; deterministic two-stage reduction) or [T,H,W,3] (dense).  `partials` is caller scratch
 * of flk_perturb_grad_scratch_bytes(). */
int64_t flk_perturb_grad_scratch_bytes(int B, int T, int H, int W);
int flk_perturb_grad_reduce(const flk_apply_args* a, const void* gx_s2d, int dtype,
                            float* gdelta, float* partials, void* stream);

/* Fused form of (stem data-gradient + flk_perturb_grad_reduce) for the flickering perturbation of I3D: d(loss)/d(delta[t,c]) straight
 * from G = d(loss)/d(pre-ReLU output of Conv3d_1a_7x7) (bf16 [B,T/2,H/2,W/2,g_ld], 64 channels) -- replaces Conv3DBackpropInputV2 of
 * i3d.py:169 and the clip_by_value / reduce_sum gradients of kinetics_i3d_utils.py:100-142 with ONE MFMA kernel in the weight-gradient
 * form (csrc/stem_grad.hip): the per-pixel gradient is never materialised.  `a` are the arguments the clip was applied with (its
 * clip / roll / 1/std semantics define the mask); wf_dev comes from flk_stem_delta_grad_weights_create (the canonical
 * [7,7,7,3,64] stem weights x the folded batch-norm scale, fp32); scratch: flk_stem_delta_grad_scratch_bytes() of caller scratch
 * (stage-1 partials + the de-interleaved clip mask, 768 bytes per clip row).  Two launches: the mask pre-pass (HBM-bound;
 * mask_done != 0 skips it when flk_stem_delta_grad_mask already ran for the same arguments) and the GEMM.
 * Deterministic (fixed summation order); bf16 gradients only -- the fp32 parity mode keeps the two-kernel path. */
int64_t flk_stem_delta_grad_scratch_bytes(int B, int T, int H);
int flk_stem_delta_grad_weights_create(const float* w7_dhwio, const float* bn_scale, float** out_dev);
int flk_stem_delta_grad_weights_destroy(float* dev);
int flk_stem_delta_grad_mask(const flk_apply_args* a, float* scratch, void* stream);   /* step 1 alone (clip mask; needs x and delta only) */
int flk_stem_delta_grad(const flk_apply_args* a, const void* G, int g_ld, const float* wf_dev, float* gdelta,
                        float* scratch, int mask_done, void* stream);

/* Exact perturbation path of the folded I3D stem in bf16 mode.  conv(x') + sum over the taps inside the clip of W * a*p'[t,c] equals
 * conv(x_adv) wherever the clip is inactive (x' from flk_perturb_apply_s2d with center = 1); the second term depends on the output
 * frame and on which taps fall outside the frame only: a [T/2][4][4][64] table, added in the stem's epilogue (flk_conv_args.pos_bias).
 * weights: [7][4][4][3][64] = bn_scale * (sums of the canonical [7,7,7,3,64] stem weights over the in-frame (kh, kw) of each class). */
int flk_stem_delta_bias_weights_create(const float* w7_dhwio, const float* bn_scale, float** out_dev);
int flk_stem_delta_bias(const flk_apply_args* a, const float* sums_dev, float* table_out, void* stream);

/* Conv3d_1a_7x7 (i3d.py:168-170: 7x7x7 / 2 SAME, 3 -> 64, batch norm + ReLU) computed straight from the uint8 clip with the flickering
 * perturbation applied on the way in (kinetics_i3d_utils.py:100-142) -- ONE kernel (csrc/stem_fwd.hip) in place of
 * flk_perturb_apply_s2d + flk_conv3d over the space-to-depth tensor: K packed along the pixel row (37 MFMA K steps instead of 49),
 * the applied clip never written to HBM.  `a`: uint8 clip [B,T,224,224,3], flicker perturbation, center = 1 (the value fed to the
 * convolution is clamp(x, lo - p, hi - p) = clip(x + p, lo, hi) - p; the perturbation's own contribution comes from pos_bias =
 * flk_stem_delta_bias's table, per clip when pos_bias_bstride != 0, or NULL).  weights: flk_stem_fwd_u8_weights_create from the
 * canonical [7,7,7,3,64] array (destroy with flk_conv_weights_destroy); bn_scale / bn_bias: fp32 [64];
 * out: bf16 [B,T/2,112,112,out_ld], channels [0,64).  bf16 only (the fp32 parity mode keeps the two-kernel path). */
int flk_stem_fwd_u8_weights_create(const float* w7_dhwio, flk_conv_weights** out);
int flk_stem_fwd_u8(const flk_apply_args* a, const flk_conv_weights* w, const float* bn_scale, const float* bn_bias,
                    const float* pos_bias, int64_t pos_bias_bstride, void* out, int out_ld, void* stream);

/* Tail of the data-parallel payload (flickering_adversarial_video_amd/parallel.py; replaces the per-iteration
 * reduce_sum / reduce_mean fetches of i3d_adversarial_main_single_video_npy.py:213-217): from the per-clip
 * outputs of flk_softmax_adv_loss ([B,4] = loss, p_label, p_max_other, argmax)
 *   out3 = { sum_b loss_b, prob_scale * sum_b p_label, prob_scale * sum_b p_max_other }   (fixed summation order). */
int flk_pack_batch_sums(const float* per_clip, int B, float prob_scale, float* out3, void* stream);

/* Regulariser gradient + Adam on delta (flicker, [T,3] time-major; the torch dialect's [3,T] is
 * transposed by the host wrapper).  TF dialect: kinetics_i3d_utils.py:177-186 +
 * i3d_adversarial_main_single_video_npy.py:56-59,79-84 (regulariser on the RAW delta, TF Adam).
 * torch dialect: model.py:198-209 (regulariser on the CLAMPED delta, torch Adam, model.py:868).
 * g_adv is the (all-reduced) adversarial-loss gradient.  Writes scalars[8] =
 * {reg_total, norm, diff, lap, thickness, roughness, max, min} of the PRE-update delta. */
typedef struct {
  int T;
  int torch_dialect;
  float beta0;               /* TF: LAMBDA; torch: lambda_ */
  float beta1, beta2, beta3; /* TF: b1*norm + b2*diff + b3*lap; torch: b1*norm + (1-b1)(diff+lap) */
  float dyn_max_norm;        /* torch clamp bound for the regulariser */
  float g_scale;             /* multiplies g_adv (1/(global batch) for mean losses) */
  float lr, adam_b1, adam_b2, adam_eps;
  int step;                  /* 1-based Adam step of THIS update */
} flk_adam_args;
int flk_perturb_reg_adam(const flk_adam_args* a, const float* g_adv, float* delta, float* m, float* v,
                         float* scalars, void* stream);
/* The same update for nclip INDEPENDENT perturbations (delta, m, v, g_adv: [nclip,T,3]; scalars: [nclip,8]) -- single-video attacks
 * batched (i3d_adversarial_main_single_video_npy.py:103-337; model.py:791-982 fit_many_videos): clip b takes Adam step
 * steps_dev[b] + 1 (a->step is ignored) and the counter is advanced on the device; clips with active_dev[b] == 0 (already
 * adversarial: retired) keep delta / m / v / counter, their scalars are still written.  active_dev may be NULL (all active);
 * dyn_max_norm_dev (torch dialect) may be NULL (a->dyn_max_norm for every clip) or hold one clamp bound per clip.
 * Per clip the arithmetic is exactly flk_perturb_reg_adam's. */
int flk_perturb_reg_adam_batched(const flk_adam_args* a, int nclip, const float* g_adv, float* delta, float* m, float* v,
                                 int* steps_dev, const int* active_dev, const float* dyn_max_norm_dev, float* scalars, void* stream);

/* Dense-delta ("sparse adversarial perturbations" baseline, kinetics_i3d_L12, kinetics_i3d_utils.py:308-521): delta is
 * [T,H,W,3] (init 1e-8, no +-0.4 clip), regulariser L12 = sum_t sqrt(mean_{hwc} delta_t^2) + 1e-12 (:409; torch dialect
 * model.py:211-214), loss = adv + beta1 * L12 (i3d_adversarial_main_universal.py:129-133), TF or torch Adam.
 * Two HBM-bound passes: per-frame sum of squares (deterministic two-stage reduction into frame_sq[T], fp32), then the
 * fused gradient + Adam update streaming delta, m, v, g_adv once (5 fp32 streams = 193 MB at T=64).
 * scalars[4] = {L12, thickness, roughness, max|delta|} of the PRE-update delta.  scratch: flk_dense_adam_scratch_bytes(). */
typedef struct {
  int T, H, W;
  int torch_dialect;
  float beta;                /* weight of L12 in the loss */
  float g_scale;
  float lr, adam_b1, adam_b2, adam_eps;
  int step;
  float dyn_max_norm;        /* > 0 (torch dialect): L12 is taken on clamp(delta, +-dyn_max_norm), as Losses receives the CLAMPED
                              * perturbation (model.py:1078,211-214); its gradient vanishes where |delta| > dyn_max_norm.  0: raw delta */
} flk_dense_adam_args;
int64_t flk_dense_adam_scratch_bytes(int T, int H, int W);
int flk_perturb_dense_l12_adam(const flk_dense_adam_args* a, const float* g_adv, float* delta, float* m, float* v,
                               float* scalars, float* scratch, void* stream);

/* Loss head: softmax + adversarial loss + d(loss)/d(logits) (kinetics_i3d_utils.py:152-169,253-307;
 * model.py:177-250).  per_clip[b*4..] = {loss_b, label_prob, max_non_label_prob, argmax}. */
typedef struct {
  int B, C;
  int torch_dialect, improve_loss, use_logits, targeted;
  float margin;
  float mean_scale;          /* CE variants are means: 1/(global batch) */
} flk_loss_args;
int flk_softmax_adv_loss(const flk_loss_args* a, const float* logits, const int64_t* labels,
                         float* softmax, float* dlogits, float* per_clip, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-network plan: the victim classifier forward + backward-to-input as one call
 * (replaces sess.run([train_op, ...]) of i3d_adversarial_main_single_video_npy.py:213-217 and
 * model([x,True]) + loss.backward() of model.py:1073-1101).  Weights are handed over once,
 * packed on the host and kept resident (never re-broadcast, cf. nn.DataParallel model.py:576).
 * ------------------------------------------------------------------------------------------- */
typedef struct flk_net flk_net;
#define FLK_NET_I3D 0
#define FLK_NET_R2PLUS1D_18 1
#define FLK_NET_R3D_18 2
#define FLK_NET_MC3_18 3

int flk_net_create(int arch, int dtype, int B, int T, int H, int W, int device, flk_net** out);
int flk_net_destroy(flk_net* n);
/* name = checkpoint variable name (kinetics_i3d_utils.py:41-62 for I3D; torchvision state_dict keys
 * for VideoResNet); data = fp32 host array in the checkpoint's own layout. */
int flk_net_set_weight(flk_net* n, const char* name, const float* data, int64_t numel);
int flk_net_finalize(flk_net* n);                 /* pack + upload; plan buffers */
int64_t flk_net_workspace_bytes(const flk_net* n);
/* x_s2d: output of flk_perturb_apply_s2d (I3D) ; logits: fp32 [B,num_classes] */
int flk_net_forward(flk_net* n, const void* x_in, float* logits, int save_for_backward, void* stream);
/* forward of a clip applied with flk_apply_args.center = 1 (I3D plan in bf16: flk_net_has_forward_flicker): computes the stem's
 * position-class bias table of `a` (flk_stem_delta_bias) and runs the plan with it */
int flk_net_has_forward_flicker(const flk_net* n);
int flk_net_forward_flicker(flk_net* n, const void* x_in, const flk_apply_args* a, float* logits, void* stream);
/* apply + forward in one call: the plan launches flk_perturb_apply_s2d itself, per batch slice on the stream that slice's stem
 * convolution runs on (I3D at bs >= 4: the two half-batches -- the second half's apply overlaps the first half's stem), writing the
 * space-to-depth clip into x_s2d_out; with a->center = 1 (flk_net_has_forward_flicker) the position-class bias path is taken.
 * a->fold_t must be the plan's layout, flk_net_input_fold(): 3 for I3D; VideoResNet plans 1 in fp32 (16 channels) and 4 in bf16 (32 channels:
 * every value as two bf16 numbers, [hi(16) | lo(16)]).  Same logits as apply followed by flk_net_forward[_flicker].
 * x_s2d_out is SCRATCH: when the stem reads the uint8 clip itself (I3D plan in bf16, a->x_is_u8 with a->center = 1 -- the default
 * engine path) it is left untouched, so a caller that needs the space-to-depth tensor calls flk_perturb_apply_s2d itself. */
int flk_net_forward_apply(flk_net* n, const flk_apply_args* a, void* x_s2d_out, float* logits, void* stream);
/* dlogits fp32 [B,C] -> gradient w.r.t. the network input: dtype of x_in; layout of x_in for I3D ([B,T/2,H/2,W/2,32]); VideoResNet plans ALWAYS
 * [B,T,H/2,W/2,16] -- the gradient of x_adv -- also where the bf16 input has 32 channels (hi | lo) */
int flk_net_backward(flk_net* n, const float* dlogits, void* gx_in, void* stream);
/* backward straight to the flickering perturbation: the plan without the stem's data-gradient, then flk_stem_delta_grad on the
 * stem's output gradient.  gdelta fp32 [T,3]; scratch: flk_stem_delta_grad_scratch_bytes(B, T, H).  I3D in bf16 only
 * (flk_net_has_backward_delta() tells); `a` = the flk_apply_args the clip of this forward pass was applied with. */
int flk_net_has_backward_delta(const flk_net* n);
int flk_net_backward_delta(flk_net* n, const float* dlogits, const flk_apply_args* a, float* gdelta, float* scratch, void* stream);
/* optional, before the forward pass of the same iteration: start the clip-mask pre-pass of the coming flk_net_backward_delta(a, scratch)
 * on the plan's own side stream (the mask depends on the clip and on delta only: kinetics_i3d_utils.py:100-142's clip_by_value
 * gradient), so that it runs beside the stem instead of beside the first kernels of the backward pass.  Same `a` contents and same
 * `scratch` as the backward call, which otherwise computes the mask itself. */
int flk_net_prepare_backward_delta(flk_net* n, const flk_apply_args* a, float* scratch, void* stream);
/* per-layer HIP-event timing of the next forward/backward (bench.py roofline leg).  While enabled the plan runs serially on
 * the caller's stream (normally independent Inception branches run on parallel streams; FLK_SINGLE_STREAM=1 disables that). */
int flk_net_profile(flk_net* n, int enable);
/* one serial forward + backward with flk_conv_set_autotune(1): tunes the launch layout of every convolution of the plan on
 * its real operands (x_in / dlogits as for forward / backward; synchronises the stream) */
int flk_net_autotune(flk_net* n, const void* x_in, float* logits, const float* dlogits, void* gx_in, void* stream);
int flk_net_profile_read(flk_net* n, char* json_out, int64_t cap);
/* the input tensor flk_net_forward reads (it cannot check the buffer it is given): elements, channels per folded position
 * (I3D 32; VideoResNet 16 in fp32, 32 in bf16) and the flk_apply_args.fold_t that produces it (3 / 1 / 4) */
int64_t flk_net_input_numel(const flk_net* n);
int flk_net_input_channels(const flk_net* n);
int flk_net_input_fold(const flk_net* n);
int flk_net_num_classes(const flk_net* n);
/* debugging / parity: copy a named activation (fp32, NDHWC) to the host */
int flk_net_get_activation(flk_net* n, const char* name, float* host_out, int64_t cap_numel, int64_t* dims5);

/* ---- data-parallel exchange: RCCL over xGMI behind the C ABI -------------------------------------------------
 * Replaces nn.DataParallel's gather of the perturbation gradient (model.py:576-578) / the dead tf.distribute.MirroredStrategy of
 * i3d_adversarial_main_universal.py:309-312: one process per GPU, ONE in-place sum all-reduce per attack iteration of the
 * (T*3 + 3)-float payload [d(sum adv)/d(delta) | sum adv | sum p_min | sum p_max] (or of the dense gradient), issued on the
 * caller's stream.  Rank 0 obtains a 128-byte id with flk_comm_unique_id and hands it to the other ranks by any side channel
 * (the Python host uses its torch.distributed process group); every rank then calls flk_comm_create.  RCCL is loaded at run time. */
typedef struct flk_comm flk_comm;
int flk_comm_unique_id(void* id128_out);
int flk_comm_create(const void* id128, int rank, int world, int device, flk_comm** out);
int flk_allreduce_sum_f32(flk_comm* c, float* buf, int64_t n, void* stream);
int flk_comm_destroy(flk_comm* c);

#ifdef __cplusplus
}
#endif
#endif

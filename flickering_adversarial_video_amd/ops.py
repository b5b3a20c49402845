"""Thin host-side wrappers over the C ABI (include/flicker_hip.h): torch tensors are used for device
memory and streams only; every computation happens in libflicker_hip.so."""
import ctypes as C
import json
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import (FLK_BF16, FLK_F32, FLK_NET_I3D, AdamArgs, ApplyArgs, ConvArgs, DenseAdamArgs, LossArgs, PoolArgs, check, dtype_code, load, ptr,
                   stream_ptr, torch_dtype)


def same_pad(n, k, s):
    """TF SAME: (out, pad_before); the extra pad goes after."""
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2


class ConvWeights:
    """Packed weights of one convolution operator (flk_conv_weights)."""

    def __init__(self, w_dhwio, dtype, nf, row_scale=None, transpose=False, cin_split=0):
        w = np.ascontiguousarray(w_dhwio, dtype=np.float32)
        assert w.ndim == 5
        self.kt, self.kh, self.kw, cin, cout = w.shape
        self.cin, self.cout = (cout, cin) if transpose else (cin, cout)
        self.dtype, self.nf = dtype_code(dtype), nf
        rs = None if row_scale is None else np.ascontiguousarray(row_scale, dtype=np.float32)
        h = C.c_void_p()
        if cin_split:
            assert not transpose
            check(load().flk_conv_weights_create_split(ptr(w), self.kt, self.kh, self.kw, cin, cout, ptr(rs), cin_split,
                                                       self.dtype, nf, C.byref(h)))
        else:
            check(load().flk_conv_weights_create(ptr(w), self.kt, self.kh, self.kw, cin, cout, ptr(rs), int(transpose),
                                                 self.dtype, nf, C.byref(h)))
        self.cin_split = cin_split
        self.handle = h

    @classmethod
    def s2d_stem(cls, w_folded, dtype, nf):
        """folded 7x7x7/2 stem: [4,4,4,32,cout] in the fold_t = 3 channel order (flk_conv_weights_create_s2d_stem)"""
        w = np.ascontiguousarray(w_folded, dtype=np.float32)
        assert w.shape[:4] == (4, 4, 4, 32)
        self = cls.__new__(cls)
        self.kt = self.kh = self.kw = 4
        self.cin, self.cout = 32, w.shape[4]
        self.dtype, self.nf, self.cin_split = dtype_code(dtype), nf, 0
        h = C.c_void_p()
        check(load().flk_conv_weights_create_s2d_stem(ptr(w), self.cout, self.dtype, nf, C.byref(h)))
        self.handle = h
        return self

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().flk_conv_weights_destroy(self.handle)
                self.handle = None
        except Exception:      # interpreter shutdown
            pass


def conv3d(x, w, **kw):
    """x: [B,T,H,W,ld] channels-last; returns / fills out [B,OT,OH,OW,ld_out].  pad = pad-before per dim
    (default: TF SAME).  out_grid = logical output grid (default: SAME output size).  Keywords: conv3d_args."""
    a, out = conv3d_args(x, w, **kw)
    check(load().flk_conv3d(C.byref(a), w.handle, dtype_code(x.dtype), stream_ptr()))
    return out


def conv3d_group(members, nfw, ring=False):
    """members: [(x, w, kwargs)] -- up to three multi-tap bf16 convolutions in ONE launch (flk_conv3d_group); returns their outputs.
    ring: the members take the LDS weight ring (all packed with nf == nfw) instead of direct-A weights"""
    built = [conv3d_args(x, w, **kw) for x, w, kw in members]
    n = len(built)
    ap = (C.POINTER(ConvArgs) * n)(*[C.pointer(a) for a, _ in built])
    wp = (C.c_void_p * n)(*[w.handle for _, w, _ in members])
    check(load().flk_conv3d_group(ap, wp, n, nfw, int(ring), dtype_code(members[0][0].dtype), stream_ptr()))
    return [o for _, o in built]


def conv3d_pc(members):
    """members: [(x, w, kwargs)] -- up to three 3x3x3 stride-1 bf16 convolutions (weights packed with nf = 4) in ONE persistent launch of
    the producer / consumer kernel (flk_conv3d_pc); returns their outputs.  Bitwise the outputs of conv3d."""
    built = [conv3d_args(x, w, **kw) for x, w, kw in members]
    n = len(built)
    ap = (C.POINTER(ConvArgs) * n)(*[C.pointer(a) for a, _ in built])
    wp = (C.c_void_p * n)(*[w.handle for _, w, _ in members])
    check(load().flk_conv3d_pc(ap, wp, n, dtype_code(members[0][0].dtype), stream_ptr()))
    return [o for _, o in built]


def conv3d_args(x, w, *, in_coff=0, cin=None, stride=(1, 1, 1), pad=None, out=None, out_coff=0, out_grid=None,
                out_stride=(1, 1, 1), out_offset=(0, 0, 0), scale=None, bias=None, add=None, add_coff=0, mask=None,
                mask_coff=0, relu=False, in2=None, in2_coff=0, out2=None, out2_coff=0, cout1=0, splitk=False, pos_bias=None):
    """the flk_conv_args of conv3d(x, w, ...) and the output tensor it will fill"""
    B, Ti, Hi, Wi, in_ld = x.shape
    cin = w.cin if cin is None else cin
    k = (w.kt, w.kh, w.kw)
    if pad is None:
        pad = tuple(same_pad(n, kk, s)[1] for n, kk, s in zip((Ti, Hi, Wi), k, stride))
    if out_grid is None:
        out_grid = tuple(same_pad(n, kk, s)[0] for n, kk, s in zip((Ti, Hi, Wi), k, stride))
    if out is None:
        phys = tuple((g - 1) * os_ + oo + 1 for g, os_, oo in zip(out_grid, out_stride, out_offset))
        out = torch.zeros((B, *phys, w.cout + out_coff), dtype=x.dtype, device=x.device)
    a = ConvArgs()
    a.in_, a.in_ld, a.in_coff, a.cin = ptr(x), in_ld, in_coff, cin
    a.B, a.Ti, a.Hi, a.Wi = B, Ti, Hi, Wi
    a.kt, a.kh, a.kw = k
    a.st, a.sh, a.sw = stride
    a.pt, a.ph, a.pw = pad
    a.To, a.Ho, a.Wo = out_grid
    a.out, a.out_ld, a.out_coff, a.cout = ptr(out), out.shape[4], out_coff, w.cout
    a.OT, a.OH, a.OW = out.shape[1:4]
    a.ost, a.osh, a.osw = out_stride
    a.oot, a.ooh, a.oow = out_offset
    a.scale, a.bias = ptr(scale), ptr(bias)
    if add is not None:
        a.add, a.add_ld, a.add_coff = ptr(add), add.shape[4], add_coff
    if mask is not None:
        a.mask, a.mask_ld, a.mask_coff = ptr(mask), mask.shape[4], mask_coff
    a.relu = int(relu)
    if pos_bias is not None:     # fp32 [1 or B, To, 4, 4, cout] position-class bias (flk_stem_delta_bias)
        a.pos_bias = ptr(pos_bias)
        a.pos_bias_bstride = pos_bias[0].numel() if pos_bias.shape[0] > 1 else 0
    if in2 is not None:
        a.in2, a.in2_ld, a.in2_coff, a.cin1 = ptr(in2), in2.shape[4], in2_coff, w.cin_split
    if out2 is not None:
        a.out2, a.out2_ld, a.out2_coff, a.cout1 = ptr(out2), out2.shape[4], out2_coff, cout1
    if splitk:           # workspace for deterministic split-K (used only where the launch would leave most CUs idle)
        nbytes = load().flk_conv_splitk_bytes(C.byref(a), w.handle)
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            a.splitk_ws, a.splitk_ws_bytes = ptr(ws), nbytes
            a._keepalive = ws
    return a, out


def _pool_args(x, C_, k, s, pad, out, idx, in_coff=0, out_coff=0):
    B, Ti, Hi, Wi, ld = x.shape
    a = PoolArgs()
    a.in_, a.in_ld, a.in_coff, a.C = ptr(x), ld, in_coff, C_
    a.B, a.Ti, a.Hi, a.Wi = B, Ti, Hi, Wi
    a.kt, a.kh, a.kw = k
    a.st, a.sh, a.sw = s
    a.pt, a.ph, a.pw = pad
    a.To, a.Ho, a.Wo = out.shape[1:4]
    a.out, a.out_ld, a.out_coff = ptr(out), out.shape[4], out_coff
    a.idx = ptr(idx)
    return a


def maxpool3d(x, k, s, C_=None, relu_input=False):
    """tf.nn.max_pool3d SAME.  Returns (out, idx uint8, ctx) with ctx for maxpool3d_bwd."""
    B, Ti, Hi, Wi, ld = x.shape
    C_ = ld if C_ is None else C_
    og, pad = zip(*(same_pad(n, kk, ss) for n, kk, ss in zip((Ti, Hi, Wi), k, s)))
    out = torch.empty((B, *og, C_), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, *og, C_), dtype=torch.uint8, device=x.device)
    a = _pool_args(x, C_, k, s, pad, out, idx)
    a.relu_input = int(relu_input)
    check(load().flk_maxpool3d_fwd(C.byref(a), dtype_code(x.dtype), stream_ptr()))
    return out, idx, (x, C_, k, s, pad, out, idx)


def maxpool3d_bwd(ctx, gout, mask=None):
    x, C_, k, s, pad, out, idx = ctx
    gin = torch.empty((*x.shape[:4], C_), dtype=x.dtype, device=x.device)
    a = _pool_args(x, C_, k, s, pad, out, idx)
    check(load().flk_maxpool3d_bwd(C.byref(a), ptr(gout), gout.shape[4], 0, ptr(gin), C_, 0, ptr(mask),
                                   0 if mask is None else mask.shape[4], 0, dtype_code(x.dtype), stream_ptr()))
    return gin


class PoolGemmWeights:
    """MFMA-packed Wt [K][C] (a 1x1x1 unit's weight transposed x batch-norm scale) for maxpool3d_bwd_gemm"""

    def __init__(self, wt_kc):
        w = np.ascontiguousarray(wt_kc, dtype=np.float32)
        self.K, self.C = w.shape
        self.handle = C.c_void_p()
        check(load().flk_pool_gemm_weights_create(w.ctypes.data_as(C.c_void_p), self.K, self.C, C.byref(self.handle)))

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            load().flk_pool_gemm_weights_destroy(self.handle)
            self.handle = C.c_void_p()


def maxpool3d_bwd_gemm(ctx, g, weights, g_coff=0):
    """Branch_3 backward in one kernel (bf16): gin = MaxPool3DGrad(idx, g[..., g_coff:g_coff+K] @ Wt); ctx from maxpool3d"""
    x, C_, k, s, pad, out, idx = ctx
    assert x.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and weights.C == C_
    gin = torch.empty((*x.shape[:4], C_), dtype=x.dtype, device=x.device)
    a = _pool_args(x, C_, k, s, pad, out, idx)
    check(load().flk_maxpool3d_bwd_gemm(C.byref(a), ptr(g), g.shape[4], g_coff, weights.K, weights.handle, ptr(gin), C_, 0,
                                        dtype_code(x.dtype), stream_ptr()))
    return gin


I3D_FOLD = 3   # space-to-depth layout the I3D plan (flk_net, FLK_NET_I3D) expects: chunk-aligned (t,h,w) fold


def make_apply_args(x, delta, *, dialect="tf", dclip=0.4, adv_flag=1.0, shift_x=0, shift_p=0, inv_std=(1.0, 1.0, 1.0),
                    lo=-1.0, hi=1.0, fold_t=2, center=False, dclip_dev=None):
    """x: uint8 or fp32 [B,T,H,W,3] on the GPU; delta fp32 [T,3] (flicker, shared by the batch), [B,T,3] (one flicker perturbation PER
    CLIP: independent single-video attacks advancing in one batch) or [T,H,W,3] (dense)."""
    B, T, H, W, c3 = x.shape
    assert c3 == 3 and x.is_contiguous() and delta.is_contiguous() and delta.dtype == torch.float32
    a = ApplyArgs()
    a.x = ptr(x)
    a.x_is_u8 = int(x.dtype == torch.uint8)
    assert x.dtype in (torch.uint8, torch.float32)
    # TFRecord path: x = u8/128 - 1 (pre_process_rgb_flow.py:226-234)
    a.x_scale, a.x_bias = (1.0 / 128.0, -1.0) if dialect == "tf" else (1.0, 0.0)
    a.delta = ptr(delta)
    a.delta_dense = int(delta.dim() == 4)
    a.delta_per_clip = int(delta.dim() == 3)
    assert tuple(delta.shape) in ((T, 3), (B, T, 3), (T, H, W, 3)), delta.shape
    assert not (a.delta_per_clip and (shift_x or shift_p)), "per-clip perturbations: cyclic rolls are drawn per run in the reference; not batched"
    a.dclip = float(dclip)
    a.inv_std = (C.c_float * 3)(*inv_std)
    a.lo, a.hi, a.adv_flag = float(lo), float(hi), float(adv_flag)
    a.shift_x, a.shift_p = int(shift_x), int(shift_p)
    a.B, a.T, a.H, a.W = B, T, H, W
    a.fold_t = fold_t
    a.center = int(center)      # write x_adv - a*p' (the clean value where the clip is inactive); see Net.forward_flicker
    if dclip_dev is not None:   # per-clip clamp bounds (fp32 [B] on the device), per-clip perturbations only
        assert a.delta_per_clip and dclip_dev.dtype == torch.float32 and dclip_dev.shape == (B,) and dclip_dev.is_cuda
    a.dclip_dev = ptr(dclip_dev)
    a._keepalive = (x, delta, dclip_dev)   # the struct holds raw pointers only
    return a


def perturb_apply_s2d(args, dtype, out=None):
    if out is None:
        ft = 1 if args.fold_t in (1, 4) else 2
        out = torch.empty((args.B, args.T // ft, args.H // 2, args.W // 2, 32 if args.fold_t == 4 else 16 * ft), dtype=torch_dtype(dtype_code(dtype)),
                          device="cuda")
    check(load().flk_perturb_apply_s2d(C.byref(args), ptr(out), dtype_code(dtype), stream_ptr()))
    return out


def perturb_grad_reduce(args, gx_s2d, gdelta=None, scratch=None):
    if gdelta is None:
        shape = (args.T, args.H, args.W, 3) if args.delta_dense else (args.B, args.T, 3) if args.delta_per_clip else (args.T, 3)
        gdelta = torch.empty(shape, dtype=torch.float32, device="cuda")
    if scratch is None and not args.delta_dense:
        n = load().flk_perturb_grad_scratch_bytes(args.B, args.T, args.H, args.W)
        scratch = torch.empty(n // 4, dtype=torch.float32, device="cuda")
    check(load().flk_perturb_grad_reduce(C.byref(args), ptr(gx_s2d), dtype_code(gx_s2d.dtype), ptr(gdelta), ptr(scratch), stream_ptr()))
    return gdelta


class StemDeltaGradWeights:
    """fp32 weights of flk_stem_delta_grad: canonical stem weights [7,7,7,3,64] x folded batch-norm scale [64]"""

    def __init__(self, w7_dhwio, bn_scale):
        w = np.ascontiguousarray(w7_dhwio, dtype=np.float32)
        sc = np.ascontiguousarray(bn_scale, dtype=np.float32)
        assert w.shape == (7, 7, 7, 3, 64) and sc.shape == (64,)
        h = C.c_void_p()
        check(load().flk_stem_delta_grad_weights_create(ptr(w), ptr(sc), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().flk_stem_delta_grad_weights_destroy(self.handle)
                self.handle = None
        except Exception:      # interpreter shutdown
            pass


def stem_delta_grad(args, G, weights, gdelta=None, scratch=None):
    """G: bf16 [B,T/2,H/2,W/2,ld] gradient of the stem's pre-ReLU output; returns d(loss)/d(delta) [T,3]"""
    assert G.dtype == torch.bfloat16 and G.is_contiguous() and G.dim() == 5
    if gdelta is None:
        gdelta = torch.empty((args.T, 3), dtype=torch.float32, device="cuda")
    if scratch is None:
        scratch = torch.empty(max(1, load().flk_stem_delta_grad_scratch_bytes(args.B, args.T, args.H) // 4), dtype=torch.float32, device="cuda")
    check(load().flk_stem_delta_grad(C.byref(args), ptr(G), G.shape[4], weights.handle, ptr(gdelta), ptr(scratch), 0, stream_ptr()))
    return gdelta


class StemFwdU8Weights:
    """MFMA fragments of flk_stem_fwd_u8 from the canonical stem weights [7,7,7,3,64] (37 K steps x 2 column parities)"""

    def __init__(self, w7_dhwio):
        w = np.ascontiguousarray(w7_dhwio, dtype=np.float32)
        assert w.shape == (7, 7, 7, 3, 64)
        h = C.c_void_p()
        check(load().flk_stem_fwd_u8_weights_create(ptr(w), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().flk_conv_weights_destroy(self.handle)
                self.handle = None
        except Exception:      # interpreter shutdown
            pass


def stem_delta_bias_table(args, w7_dhwio, bn_scale):
    """position-class bias table of the stem for the perturbation of `args` (flk_stem_delta_bias): fp32 [1 or B, T/2, 4, 4, 64]"""
    w = np.ascontiguousarray(w7_dhwio, dtype=np.float32)
    sc = np.ascontiguousarray(bn_scale, dtype=np.float32)
    h = C.c_void_p()
    check(load().flk_stem_delta_bias_weights_create(ptr(w), ptr(sc), C.byref(h)))
    try:
        tab = torch.zeros((args.B if args.delta_per_clip else 1, args.T // 2, 4, 4, 64), dtype=torch.float32, device="cuda")
        check(load().flk_stem_delta_bias(C.byref(args), h, ptr(tab), stream_ptr()))
        torch.cuda.synchronize()
    finally:
        load().flk_stem_delta_grad_weights_destroy(h)
    return tab


def stem_fwd_u8(args, weights, bn_scale, bn_bias, pos_bias=None, out=None):
    """Conv3d_1a_7x7 + batch norm + ReLU straight from the uint8 clip of `args` (center = 1): bf16 [B,T/2,112,112,64]"""
    assert bn_scale.dtype == torch.float32 and bn_bias.dtype == torch.float32 and bn_scale.is_cuda and bn_bias.is_cuda
    if out is None:
        out = torch.empty((args.B, args.T // 2, 112, 112, 64), dtype=torch.bfloat16, device="cuda")
    bstride = 0
    if pos_bias is not None:
        assert pos_bias.dtype == torch.float32 and pos_bias.is_contiguous() and pos_bias.shape[1:] == (args.T // 2, 4, 4, 64)
        bstride = pos_bias[0].numel() if pos_bias.shape[0] > 1 else 0
    check(load().flk_stem_fwd_u8(C.byref(args), weights.handle, ptr(bn_scale), ptr(bn_bias), ptr(pos_bias), bstride, ptr(out),
                                 out.shape[4], stream_ptr()))
    return out


def perturb_reg_adam(g_adv, delta, m, v, step, *, dialect="tf", beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5,
                     dyn_max_norm=0.0, g_scale=1.0, lr=1e-3, adam=(0.9, 0.999, 1e-8), scalars=None):
    a = AdamArgs()
    a.T = delta.shape[0]
    a.torch_dialect = int(dialect == "torch")
    a.beta0, a.beta1, a.beta2, a.beta3 = beta0, beta1, beta2, beta3
    a.dyn_max_norm, a.g_scale, a.lr = dyn_max_norm, g_scale, lr
    a.adam_b1, a.adam_b2, a.adam_eps = adam
    a.step = int(step)
    if scalars is None:
        scalars = torch.empty(8, dtype=torch.float32, device="cuda")
    check(load().flk_perturb_reg_adam(C.byref(a), ptr(g_adv), ptr(delta), ptr(m), ptr(v), ptr(scalars), stream_ptr()))
    return scalars


def perturb_reg_adam_batched(g_adv, delta, m, v, steps, active=None, *, dialect="tf", beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5,
                             dyn_max_norm=0.0, g_scale=1.0, lr=1e-3, adam=(0.9, 0.999, 1e-8), scalars=None, dyn_max_norm_dev=None):
    """B independent perturbations [B,T,3], each with its own Adam state and DEVICE step counter ``steps`` (int32 [B], advanced by the
    kernel); clips with ``active[b] == 0`` are frozen.  Returns scalars [B,8] of the pre-update perturbations."""
    B, T, _ = delta.shape
    assert steps.dtype == torch.int32 and steps.shape == (B,) and steps.is_cuda and (active is None or (active.dtype == torch.int32 and active.shape == (B,)))
    for t in (g_adv, delta, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == B * T * 3
    a = AdamArgs()
    a.T = T
    a.torch_dialect = int(dialect == "torch")
    a.beta0, a.beta1, a.beta2, a.beta3 = beta0, beta1, beta2, beta3
    a.dyn_max_norm, a.g_scale, a.lr = dyn_max_norm, g_scale, lr
    a.adam_b1, a.adam_b2, a.adam_eps = adam
    a.step = 0
    if scalars is None:
        scalars = torch.empty((B, 8), dtype=torch.float32, device="cuda")
    assert dyn_max_norm_dev is None or (dyn_max_norm_dev.dtype == torch.float32 and dyn_max_norm_dev.shape == (B,) and dyn_max_norm_dev.is_cuda)
    check(load().flk_perturb_reg_adam_batched(C.byref(a), B, ptr(g_adv), ptr(delta), ptr(m), ptr(v), ptr(steps), ptr(active),
                                              ptr(dyn_max_norm_dev), ptr(scalars), stream_ptr()))
    return scalars


def perturb_dense_l12_adam(g_adv, delta, m, v, step, *, dialect="tf", beta=1.0, g_scale=1.0, lr=1e-3, adam=(0.9, 0.999, 1e-8),
                           scalars=None, scratch=None, dyn_max_norm=0.0):
    """dense delta [T,H,W,3]: L12 regulariser gradient + Adam (kinetics_i3d_L12); returns scalars {L12, thickness, roughness, max}.
    dyn_max_norm > 0 (torch dialect): L12 of the clamped perturbation, as Losses receives it (model.py:1078)"""
    T, H, W, _ = delta.shape
    a = DenseAdamArgs()
    a.T, a.H, a.W, a.torch_dialect = T, H, W, int(dialect == "torch")
    a.beta, a.g_scale, a.lr = beta, g_scale, lr
    a.adam_b1, a.adam_b2, a.adam_eps = adam
    a.step = int(step)
    a.dyn_max_norm = float(dyn_max_norm)
    if scalars is None:
        scalars = torch.empty(4, dtype=torch.float32, device="cuda")
    if scratch is None:
        scratch = torch.empty(load().flk_dense_adam_scratch_bytes(T, H, W) // 4, dtype=torch.float32, device="cuda")
    check(load().flk_perturb_dense_l12_adam(C.byref(a), ptr(g_adv), ptr(delta), ptr(m), ptr(v), ptr(scalars), ptr(scratch), stream_ptr()))
    return scalars


def pack_batch_sums(per_clip, prob_scale, out3):
    """out3 = [sum loss_b, prob_scale * sum p_label, prob_scale * sum p_max_other] from softmax_adv_loss's per-clip table"""
    assert per_clip.dtype == torch.float32 and per_clip.is_contiguous() and per_clip.shape[1] == 4 and out3.dtype == torch.float32
    check(load().flk_pack_batch_sums(ptr(per_clip), per_clip.shape[0], float(prob_scale), ptr(out3), stream_ptr()))
    return out3


_labels_ok = {}      # id(LIVE label tensor object) -> (weakref to it, _version, num_classes) it was range-checked at


def _labels_memo_get(labels):
    ent = _labels_ok.get(id(labels))
    return (ent[1], ent[2]) if ent is not None and ent[0]() is labels else None


def mark_labels_validated(labels, num_classes):
    """Record that ``labels`` (a CUDA tensor) was range-checked where it originated, on the host (the dataset driver checks the
    numpy labels of every record batch before the copy): check_labels then skips its device read-back for this tensor object.
    The entry is removed when the tensor object dies (weakref callback), so a recycled id / address can never inherit it."""
    key = id(labels)
    _labels_ok[key] = (weakref.ref(labels, lambda _r, k=key: _labels_ok.pop(k, None)), labels._version, num_classes)
    return labels


def check_labels(labels, batch, num_classes):
    """labels must be a contiguous CUDA int64 tensor of shape (batch,) with 0 <= label < num_classes: anything else would hand the
    loss kernel a host pointer (GPU memory fault), a wrong stride or an out-of-range class index.  Device / dtype / shape are checked
    on every call (free).  The range check reads min / max back from the device (two syncs) ONCE per live tensor object and
    version: the memo is keyed on the identity of the LIVE tensor object (weak reference) -- not on its address, which the caching allocator recycles for
    the next batch's labels -- so an entry dies with its tensor and an in-place write invalidates it.  A loop that reuses its label
    tensor pays the read-back once; a loop that builds labels per batch validates them on the host and says so
    (mark_labels_validated).  Independently, the kernel clamps a bad index and poisons that clip's outputs with NaN (head.hip)."""
    if not torch.is_tensor(labels) or not labels.is_cuda or labels.dtype != torch.int64 or tuple(labels.shape) != (batch,):
        desc = f"{tuple(labels.shape)} {labels.dtype} {labels.device}" if torch.is_tensor(labels) else type(labels).__name__
        raise ValueError(f"labels must be a CUDA int64 tensor of shape ({batch},), got {desc}")
    if not labels.is_contiguous():
        raise ValueError("labels must be contiguous")
    if _labels_memo_get(labels) != (labels._version, num_classes):
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= num_classes:
            raise ValueError(f"labels must lie in [0, {num_classes}), got [{lo}, {hi}]")
        mark_labels_validated(labels, num_classes)
    return labels


def softmax_adv_loss(logits, labels, *, dialect="tf", improve_loss=True, use_logits=False, targeted=False, margin=0.05,
                     mean_scale=1.0, out=None):
    """out: optional (softmax, dlogits, per_clip) buffers to write into"""
    B, Cn = logits.shape
    a = LossArgs()
    a.B, a.C = B, Cn
    a.torch_dialect, a.improve_loss, a.use_logits, a.targeted = int(dialect == "torch"), int(improve_loss), int(use_logits), int(targeted)
    a.margin, a.mean_scale = margin, mean_scale
    if out is None:
        sm = torch.empty_like(logits)
        dl = torch.empty_like(logits)
        pc = torch.empty((B, 4), dtype=torch.float32, device=logits.device)
    else:
        sm, dl, pc = out
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    check_labels(labels, B, Cn)
    check(load().flk_softmax_adv_loss(C.byref(a), ptr(logits), ptr(labels), ptr(sm), ptr(dl), ptr(pc), stream_ptr()))
    return sm, dl, pc


class Net:
    """flk_net: whole-network forward + backward-to-input plan with resident packed weights."""

    def __init__(self, arch, dtype, B, T, H, W, weights, device=0):
        self.dtype = dtype_code(dtype)
        self.B, self.T, self.H, self.W = B, T, H, W
        h = C.c_void_p()
        check(load().flk_net_create(arch, self.dtype, B, T, H, W, device, C.byref(h)))
        self.handle = h
        for name, arr in weights.items():
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            check(load().flk_net_set_weight(h, name.encode(), ptr(arr), arr.size))
        check(load().flk_net_finalize(h))
        self.num_classes = load().flk_net_num_classes(h)
        self.input_numel = load().flk_net_input_numel(h)
        # VideoResNet plans: channels of the (h,w)-folded input tensor -- 16, or 32 in bf16 (two bf16 numbers per value, fold_t = 4);
        # the input GRADIENT always has the 16-channel layout
        self.input_channels = load().flk_net_input_channels(h)
        self.input_fold = load().flk_net_input_fold(h)
        self.grad_numel = self.input_numel if arch == FLK_NET_I3D else self.input_numel // self.input_channels * 16
        self.workspace_bytes = load().flk_net_workspace_bytes(h)

    def forward(self, x_in, logits=None):
        if logits is None:
            logits = torch.empty((self.B, self.num_classes), dtype=torch.float32, device="cuda")
        assert x_in.numel() == self.input_numel and x_in.is_contiguous()
        check(load().flk_net_forward(self.handle, ptr(x_in), ptr(logits), 1, stream_ptr()))
        return logits

    @property
    def has_forward_flicker(self):
        """the exact perturbation path of the stem is available: I3D plan, bf16 (FLK_STEM_CENTER=0 switches it off)"""
        return bool(load().flk_net_has_forward_flicker(self.handle))

    def forward_flicker(self, x_in, apply_args, logits=None):
        """forward of a clip applied with ``center=True``: the perturbation enters the stem in fp32 through its epilogue"""
        if logits is None:
            logits = torch.empty((self.B, self.num_classes), dtype=torch.float32, device="cuda")
        assert x_in.numel() == self.input_numel and x_in.is_contiguous() and apply_args.center == 1
        check(load().flk_net_forward_flicker(self.handle, ptr(x_in), C.byref(apply_args), ptr(logits), stream_ptr()))
        return logits

    def forward_apply(self, apply_args, x_s2d, logits=None):
        """perturbation apply + forward in one call: the plan applies each batch slice on the stream its stem convolution runs on
        (x_s2d is scratch: it receives the space-to-depth clip unless the stem reads the uint8 clip itself -- the bf16 I3D plan with a
        centred uint8 clip -- in which case it is left untouched; call perturb_apply_s2d when the tensor itself is needed)"""
        if logits is None:
            logits = torch.empty((self.B, self.num_classes), dtype=torch.float32, device=x_s2d.device)
        assert x_s2d.is_contiguous() and x_s2d.numel() == self.input_numel and dtype_code(x_s2d.dtype) == self.dtype
        check(load().flk_net_forward_apply(self.handle, C.byref(apply_args), ptr(x_s2d), ptr(logits), stream_ptr()))
        return logits

    def backward(self, dlogits, gx=None):
        if gx is None:
            gx = torch.empty(self.grad_numel, dtype=torch_dtype(self.dtype), device="cuda")
        check(load().flk_net_backward(self.handle, ptr(dlogits), ptr(gx), stream_ptr()))
        return gx

    @property
    def has_backward_delta(self):
        """the fused stem delta-gradient (csrc/stem_grad.hip) is available: I3D plan, bf16 (FLK_STEM_FUSED=0 switches it off)"""
        return bool(load().flk_net_has_backward_delta(self.handle))

    def prepare_backward_delta(self, apply_args, scratch):
        """start the clip-mask pre-pass of the coming backward_delta(…, apply_args, …, scratch) now (beside the forward pass)"""
        check(load().flk_net_prepare_backward_delta(self.handle, C.byref(apply_args), ptr(scratch), stream_ptr()))

    def backward_delta(self, dlogits, apply_args, gdelta, scratch):
        """backward straight to the flickering perturbation [T,3]: no per-pixel input gradient is materialised"""
        assert gdelta.dtype == torch.float32 and gdelta.is_contiguous() and gdelta.numel() == 3 * self.T * (self.B if apply_args.delta_per_clip else 1)
        assert scratch.numel() * 4 >= load().flk_stem_delta_grad_scratch_bytes(self.B, self.T, self.H)
        check(load().flk_net_backward_delta(self.handle, ptr(dlogits), C.byref(apply_args), ptr(gdelta), ptr(scratch), stream_ptr()))
        return gdelta

    def autotune(self, x, logits, dlogits, gx):
        """tune the launch layout of every convolution of the plan on these operands (one serial forward + backward)"""
        check(load().flk_net_autotune(self.handle, ptr(x), ptr(logits), ptr(dlogits), ptr(gx), stream_ptr()))

    def profile(self, enable):
        check(load().flk_net_profile(self.handle, int(enable)))

    def profile_read(self):
        buf = C.create_string_buffer(1 << 20)
        check(load().flk_net_profile_read(self.handle, buf, len(buf)))
        return json.loads(buf.value.decode())

    def activation(self, name):
        dims = (C.c_int64 * 5)()
        check(load().flk_net_get_activation(self.handle, name.encode(), None, 0, dims))
        out = np.empty(tuple(dims), dtype=np.float32)
        check(load().flk_net_get_activation(self.handle, name.encode(), ptr(out), out.size, dims))
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().flk_net_destroy(self.handle)
                self.handle = None
        except Exception:      # interpreter shutdown
            pass

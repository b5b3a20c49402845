"""The stem computed straight from the uint8 clip (csrc/stem_fwd.hip: perturbation apply + Conv3d_1a_7x7 + batch norm + ReLU in one
kernel, K packed along the pixel row) against the CPU oracle -- torch-CPU conv3d over the applied clip -- and inside the I3D plan.

Reference lines: i3d.py:168-170 (7x7x7 / 2 SAME convolution, 3 -> 64), kinetics_i3d_utils.py:100-142 (apply + clip).

Tolerance: the kernel multiplies bf16(clamp(x, lo - p, hi - p)) by bf16 weights, accumulates in fp32 and adds the perturbation's own
contribution in fp32 (position-class table); the oracle below evaluates exactly that model in fp32 on the CPU, so what is left is the
summation order and ONE bf16 rounding of the output: 1e-2 of the output scale (measured values are printed)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import ops as o
    return o


def oracle_stem(xu, delta, w7, scale, bias, *, dclip=0.4, adv_flag=1.0, shift_x=0, shift_p=0, lo=-1.0, hi=1.0):
    """relu(bn(conv7(x_adv))) with x_adv = clip(roll(x) + a * roll(p), lo, hi), evaluated the way the kernel splits it:
    conv(bf16(x_adv - a p')) with bf16 weights + conv(a p' inside the frame) with the fp32 weights.  xu uint8 [B,T,H,W,3];
    delta [T,3] or [B,T,3]."""
    B, T = xu.shape[:2]
    x = xu.float() / 128 - 1
    x = torch.roll(x, shift_x, 1)
    d = delta if delta.dim() == 3 else delta[None].expand(B, T, 3)
    p = adv_flag * torch.roll(d.clamp(-dclip, dclip), shift_p, 1)                      # [B,T,3]
    pf = p[:, :, None, None, :].expand(B, T, 224, 224, 3)
    xc = (torch.clamp(x + pf, lo, hi) - pf).bfloat16().float()

    def conv(inp, w):
        y = F.conv3d(F.pad(inp.permute(0, 4, 1, 2, 3), (2, 3, 2, 3, 2, 3)), w.permute(4, 3, 0, 1, 2).contiguous(), stride=2)
        return y.permute(0, 2, 3, 4, 1)
    y = conv(xc, w7.bfloat16().float()) + conv(pf.contiguous(), w7)
    return torch.relu(y * scale + bias)


CASES = [
    # name, B, T, per-clip delta, shift_x, shift_p, adv_flag
    ("shared", 2, 16, False, 0, 0, 1.0),
    ("per_clip_T10", 2, 10, True, 0, 0, 1.0),              # To = 5: a partial tile along t
    ("rolled", 1, 12, False, 5, 3, 1.0),
    ("clean", 1, 8, False, 0, 0, 0.0),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_stem_fwd_u8_vs_oracle(ops, case):
    _, B, T, per_clip, sx, sp, adv = case
    rng = np.random.default_rng(11)
    xu = torch.from_numpy(rng.integers(0, 256, (B, T, 224, 224, 3), dtype=np.uint8))
    xu[:, :, :8] = 0                      # saturated bands: the clip to [-1, 1] is active there
    xu[:, :, -8:] = 255
    delta = torch.from_numpy(rng.uniform(-0.5, 0.5, (B, T, 3) if per_clip else (T, 3)).astype(np.float32))
    w7 = torch.from_numpy((rng.standard_normal((7, 7, 7, 3, 64)) * (2.0 / 1029) ** 0.5).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32))
    bias = torch.from_numpy(rng.uniform(-0.3, 0.3, 64).astype(np.float32))
    ref = oracle_stem(xu, delta, w7, scale, bias, adv_flag=adv, shift_x=sx, shift_p=sp)

    xg, dg = xu.cuda(), delta.cuda()
    a = ops.make_apply_args(xg, dg, adv_flag=adv, shift_x=sx, shift_p=sp, fold_t=ops.I3D_FOLD, center=True)
    tab = ops.stem_delta_bias_table(a, w7.numpy(), scale.numpy())
    w = ops.StemFwdU8Weights(w7.numpy())
    out = ops.stem_fwd_u8(a, w, scale.cuda(), bias.cuda(), tab)
    torch.cuda.synchronize()
    got = out.float().cpu()
    s = float(ref.abs().max())
    err = float((got - ref).abs().max()) / s
    print(f"[{case[0]}] stem from uint8 vs oracle: max abs err / output scale = {err:.2e} (scale {s:.3f})")
    assert err < 1e-2
    # a second run gives the same bits (no atomics, fixed summation order)
    out2 = ops.stem_fwd_u8(a, w, scale.cuda(), bias.cuda(), tab)
    assert torch.equal(out, out2)


def test_stem_fwd_u8_matches_the_two_kernel_path(ops):
    """same operands through flk_perturb_apply_s2d + the folded 4x4x4 convolution (conv_igemm mode 4) with the same position-class table:
    both multiply the same bf16 values, so they agree to fp32 summation order + one bf16 rounding"""
    B, T = 2, 8
    rng = np.random.default_rng(5)
    xu = torch.from_numpy(rng.integers(0, 256, (B, T, 224, 224, 3), dtype=np.uint8)).cuda()
    delta = torch.from_numpy(rng.uniform(-0.3, 0.3, (T, 3)).astype(np.float32)).cuda()
    w7 = (rng.standard_normal((7, 7, 7, 3, 64)) * (2.0 / 1029) ** 0.5).astype(np.float32)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.uniform(-0.3, 0.3, 64).astype(np.float32)).cuda()
    a = ops.make_apply_args(xu, delta, fold_t=ops.I3D_FOLD, center=True)
    tab = ops.stem_delta_bias_table(a, w7, scale.cpu().numpy())
    new = ops.stem_fwd_u8(a, ops.StemFwdU8Weights(w7), scale, bias, tab)
    wf = np.zeros((4, 4, 4, 32, 64), np.float32)
    for kt in range(7):
        for kh in range(7):
            for kw in range(7):
                ch = ((kt & 1) * 2 + (kh & 1)) * 8 + (kw & 1) * 3
                wf[kt >> 1, kh >> 1, kw >> 1, ch:ch + 3] = w7[kt, kh, kw]
    xs = ops.perturb_apply_s2d(a, "bf16")
    old = ops.conv3d(xs, ops.ConvWeights.s2d_stem(wf, torch.bfloat16, 4), pad=(1, 1, 1), out_grid=(T // 2, 112, 112), scale=scale, bias=bias,
                     relu=True, pos_bias=tab)
    s = float(old.float().abs().max())
    err = float((new.float() - old.float()).abs().max()) / s
    print(f"stem from uint8 vs apply + folded convolution: {err:.2e} of the output scale")
    assert err < 8e-3


@pytest.mark.parametrize("batch", [1, 4])
def test_plan_uses_the_uint8_stem(ops, batch, monkeypatch):
    """flk_net_forward_apply in bf16 on a uint8 clip: the stem reads the clip itself (the space-to-depth buffer stays untouched) and
    endpoints / logits agree with the two-kernel path (FLK_STEM_U8=0) at bf16 accuracy; the backward pass is unaffected"""
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd._lib import FLK_NET_I3D
    T = 16
    W = i3d_spec.synthetic_i3d_weights(42)
    net = ops.Net(FLK_NET_I3D, "bf16", batch, T, 224, 224, W)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(batch, T, seed=8)).cuda()
    rng = np.random.default_rng(1)
    for shape in ((T, 3), (batch, T, 3)):
        d = torch.from_numpy(rng.uniform(-0.3, 0.3, shape).astype(np.float32)).cuda()
        a = ops.make_apply_args(xu, d, fold_t=ops.I3D_FOLD, center=True)
        x2 = torch.full((batch, T // 2, 112, 112, 32), 7.0, dtype=torch.bfloat16, device="cuda")
        l_new = net.forward_apply(a, x2).clone()
        a1_new = torch.from_numpy(net.activation("Conv3d_1a_7x7"))
        assert bool((x2 == 7.0).all()), "the space-to-depth tensor was written: the uint8 stem did not run"
        monkeypatch.setenv("FLK_STEM_U8", "0")
        l_old = net.forward_apply(a, x2).clone()
        a1_old = torch.from_numpy(net.activation("Conv3d_1a_7x7"))
        monkeypatch.delenv("FLK_STEM_U8")
        assert not bool((x2 == 7.0).all())
        e1 = float((a1_new - a1_old).abs().max() / a1_old.abs().max())
        el = float((l_new - l_old).abs().max() / l_old.abs().max())
        print(f"[bs {batch}, delta {shape}] uint8 stem vs two-kernel path: stem output {e1:.2e}, logits {el:.2e}")
        assert e1 < 8e-3 and el < 2e-2


def test_prepared_clip_mask_gives_the_same_delta_gradient(ops):
    """flk_net_prepare_backward_delta (the stem's clip-mask pre-pass started before the forward pass, on the plan's own side stream) changes
    WHEN the mask is computed, not what it is: the delta-gradient is bitwise the one of the unprepared call; a preparation for other
    arguments is ignored (the backward call computes its own mask)"""
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd._lib import FLK_NET_I3D, load
    B, T = 2, 16
    W = i3d_spec.synthetic_i3d_weights(42)
    net = ops.Net(FLK_NET_I3D, "bf16", B, T, 224, 224, W)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T, seed=3)).cuda()
    rng = np.random.default_rng(2)
    d1 = torch.from_numpy(rng.uniform(-0.3, 0.3, (T, 3)).astype(np.float32)).cuda()
    d2 = torch.from_numpy(rng.uniform(-0.3, 0.3, (T, 3)).astype(np.float32)).cuda()
    dl = torch.from_numpy(rng.standard_normal((B, 400)).astype(np.float32) * 1e-3).cuda()
    scratch = torch.empty(max(1, load().flk_stem_delta_grad_scratch_bytes(B, T, 224) // 4), dtype=torch.float32, device="cuda")
    xs = torch.empty((B, T // 2, 112, 112, 32), dtype=torch.bfloat16, device="cuda")

    def grad(delta, prepare_with=None):
        a = ops.make_apply_args(xu, delta, fold_t=ops.I3D_FOLD, center=True)
        if prepare_with is not None:
            net.prepare_backward_delta(ops.make_apply_args(xu, prepare_with, fold_t=ops.I3D_FOLD, center=True), scratch)
        net.forward_apply(a, xs)
        g = torch.empty((T, 3), dtype=torch.float32, device="cuda")
        net.backward_delta(dl, a, g, scratch)
        torch.cuda.synchronize()
        return g.clone()

    plain = grad(d1)
    assert torch.isfinite(plain).all() and float(plain.abs().max()) > 0
    assert torch.equal(grad(d1, prepare_with=d1), plain)            # prepared: same bits
    assert torch.equal(grad(d1, prepare_with=d2), plain)            # prepared for another perturbation: ignored
    assert not torch.equal(grad(d2), plain)

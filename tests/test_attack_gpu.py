"""Parity of the fused perturbation kernels and the loss head (GPU, through the C ABI) against the
CPU oracle (oracle/attack_math.py, pinned by the reference's golden vectors) and against the golden
vectors themselves."""
import numpy as np
import pytest
import torch

from oracle import attack_math as am

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import ops as o
    return o


def s2d(x):
    """[B,T,H,W,3] -> [B,T/2,H/2,W/2,32], channel (qt*4+qh*2+qw)*3+c, 24..31 zero (include/flicker_hip.h)"""
    B, T, H, W, _ = x.shape
    y = x.reshape(B, T // 2, 2, H // 2, 2, W // 2, 2, 3).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B, T // 2, H // 2, W // 2, 24)
    return torch.cat([y, torch.zeros(*y.shape[:4], 8)], -1)


def s2d_aligned(x):
    """fold_t = 3: channel (qt*2+qh)*8 + qw*3 + c, 6 and 7 of every 8 zero (one (qt,qh) parity per 16-byte chunk)"""
    B, T, H, W, _ = x.shape
    y = x.reshape(B, T // 2, 2, H // 2, 2, W // 2, 2, 3).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B, T // 2, H // 2, W // 2, 4, 6)
    return torch.cat([y, torch.zeros(*y.shape[:5], 2)], -1).reshape(B, T // 2, H // 2, W // 2, 32)


def un_s2d(g):
    B, T2, H2, W2, _ = g.shape
    return g[..., :24].reshape(B, T2, H2, W2, 2, 2, 2, 3).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, T2 * 2, H2 * 2, W2 * 2, 3)


@pytest.mark.parametrize("u8", [True, False], ids=["u8", "f32in"])
@pytest.mark.parametrize("dense", [False, True], ids=["flicker", "dense"])
@pytest.mark.parametrize("shifts", [(0, 0), (3, 5)], ids=["plain", "cyclic"])
def test_apply_and_grad_tf(ops, u8, dense, shifts):
    """kinetics_i3d_utils.py:100-142: clip(x' + a*p', -1, 1) and its gradient w.r.t. delta (both clips inclusive)"""
    B, T, H, W = 2, 8, 12, 16       # W % 8 == 0: the uint8 flicker case takes the 8-byte-load fast path
    rng = np.random.default_rng(5)
    xu = torch.from_numpy(rng.integers(0, 256, (B, T, H, W, 3), dtype=np.uint8))
    x = xu.float() / 128 - 1 if u8 else torch.from_numpy(rng.uniform(-1, 1, (B, T, H, W, 3)).astype(np.float32))
    dshape = (T, H, W, 3) if dense else (T, 1, 1, 3)
    d = torch.from_numpy(rng.uniform(-0.6, 0.6, dshape).astype(np.float32)).requires_grad_(True)  # some beyond +-0.4
    sx, sp = shifts
    xa = am.tf_apply(x, d, 1.0, sx, float(sx != 0), sp, float(sp != 0), clip_delta=not dense)
    gw = torch.from_numpy(rng.standard_normal(xa.shape).astype(np.float32))
    (gref,) = torch.autograd.grad(xa, d, gw)
    dd = d.detach().reshape(T, H, W, 3) if dense else d.detach().reshape(T, 3)
    args = ops.make_apply_args((xu if u8 else x).cuda(), dd.contiguous().cuda(), dclip=0.0 if dense else 0.4, shift_x=sx, shift_p=sp)
    out = ops.perturb_apply_s2d(args, torch.float32)
    torch.testing.assert_close(out.cpu(), s2d(xa.detach()), rtol=0, atol=1e-7)
    out16 = ops.perturb_apply_s2d(args, torch.bfloat16)
    torch.testing.assert_close(out16.float().cpu(), s2d(xa.detach()).bfloat16().float(), rtol=0, atol=0)
    g = ops.perturb_grad_reduce(args, s2d(gw).cuda())
    torch.testing.assert_close(g.cpu().reshape(gref.shape), gref, rtol=1e-4, atol=1e-4)
    # adv_flag = 0 -> clean clip
    args0 = ops.make_apply_args((xu if u8 else x).cuda(), dd.contiguous().cuda(), adv_flag=0.0)
    torch.testing.assert_close(ops.perturb_apply_s2d(args0, torch.float32).cpu(), s2d(x.clamp(-1, 1)), rtol=0, atol=1e-7)
    # fold_t = 3 (chunk-aligned channels, the I3D plan's layout): same values, other channel order -- apply and gradient
    args3 = ops.make_apply_args((xu if u8 else x).cuda(), dd.contiguous().cuda(), dclip=0.0 if dense else 0.4, shift_x=sx, shift_p=sp, fold_t=3)
    torch.testing.assert_close(ops.perturb_apply_s2d(args3, torch.float32).cpu(), s2d_aligned(xa.detach()), rtol=0, atol=1e-7)
    torch.testing.assert_close(ops.perturb_apply_s2d(args3, torch.bfloat16).float().cpu(), s2d_aligned(xa.detach()).bfloat16().float(), rtol=0, atol=0)
    g3 = ops.perturb_grad_reduce(args3, s2d_aligned(gw).cuda())
    torch.testing.assert_close(g3.cpu().reshape(gref.shape), gref, rtol=1e-4, atol=1e-4)
    g3b = ops.perturb_grad_reduce(args3, s2d_aligned(gw).bfloat16().cuda())
    torch.testing.assert_close(g3b.cpu().reshape(gref.shape), g.cpu().reshape(gref.shape), rtol=2e-2, atol=2e-2 * float(gref.abs().max()))


def test_apply_and_grad_torch_dialect_golden(ops, golden):
    """Perturbation.forward (model.py:80-101) via the same kernel: delta/std, scalar clamp bounds.
    Checked against the golden vectors produced by the reference class itself."""
    g = golden
    x = torch.from_numpy(g["pert_x"])                   # [2,3,16,8,8] NCDHW normalised
    w = torch.from_numpy(g["pert_w"])
    xcl = x.permute(0, 2, 3, 4, 1).contiguous()
    for tag in ("flk01", "flk02", "dense02"):
        d = torch.from_numpy(g[f"pert_{tag}_delta"])     # [3,16,1,1] | [3,16,8,8]
        dense = d.shape[2] > 1
        dcl = d.permute(1, 2, 3, 0).contiguous().reshape((16, 8, 8, 3) if dense else (16, 3))
        args = ops.make_apply_args(xcl.cuda(), dcl.cuda(), dialect="torch", dclip=float(g[f"pert_{tag}_max_norm"]),
                                   inv_std=tuple(1.0 / s for s in am.DEFAULT_STD), lo=am.TORCH_MIN_VALUE, hi=am.TORCH_MAX_VALUE)
        out = un_s2d(ops.perturb_apply_s2d(args, torch.float32).cpu()).permute(0, 4, 1, 2, 3)
        torch.testing.assert_close(out, torch.from_numpy(g[f"pert_{tag}_xadv"]), rtol=1e-6, atol=1e-6)
        gd = ops.perturb_grad_reduce(args, s2d(w.permute(0, 2, 3, 4, 1).contiguous()).cuda()).cpu()
        gd = gd.reshape(16, 8, 8, 3).permute(3, 0, 1, 2) if dense else gd.reshape(16, 1, 1, 3).permute(3, 0, 1, 2)
        torch.testing.assert_close(gd, torch.from_numpy(g[f"pert_{tag}_grad"]), rtol=1e-4, atol=1e-4)


LOSS_MODES = [("tf", True, False, False), ("tf", True, True, False), ("tf", True, False, True), ("tf", True, True, True),
              ("tf", False, False, False), ("tf", False, False, True),
              ("torch", True, False, False), ("torch", True, True, False), ("torch", False, False, False),
              ("torch", False, False, True)]


@pytest.mark.parametrize("dialect,improve,use_logits,targeted", LOSS_MODES)
def test_loss_head_vs_oracle(ops, golden, dialect, improve, use_logits, targeted):
    rng = np.random.default_rng(9)
    lg = torch.from_numpy(np.concatenate([golden["loss_logits"], (rng.standard_normal((4, 400)) * 2).astype(np.float32)]))
    labels = torch.from_numpy(np.concatenate([golden["loss_labels"], rng.integers(0, 400, 4)]))
    labels[5] = int(lg[5].argmax())              # margin region u > m
    labels[6] = int(lg[6].argsort()[-2])
    if targeted:
        labels[:] = 17 if dialect == "torch" else labels
    B = lg.shape[0]
    z = lg.clone().requires_grad_(True)
    if dialect == "tf":
        if improve:
            loss, to_min, to_max = am.tf_improve_adversarial_loss(z, labels, 0.05, targeted, use_logits)
        else:
            loss, to_min, to_max = am.tf_ce_adversarial_loss(z, labels, targeted)
    else:
        p = torch.softmax(z, 1)
        loss = am.torch_improve_loss(z, p, labels, 0.05, use_logits) if improve else am.torch_ce_loss(p, labels, targeted, 17)
    (gref,) = torch.autograd.grad(loss, z)
    sm, dl, pc = ops.softmax_adv_loss(lg.cuda(), labels.cuda(), dialect=dialect, improve_loss=improve, use_logits=use_logits,
                                      targeted=targeted, margin=0.05, mean_scale=1.0 / B)
    torch.testing.assert_close(sm.cpu(), torch.softmax(lg, 1), rtol=1e-5, atol=1e-8)
    assert pc[:, 0].sum().item() == pytest.approx(loss.item(), rel=1e-4, abs=1e-7)
    torch.testing.assert_close(dl.cpu(), gref, rtol=2e-4, atol=1e-7)
    np.testing.assert_array_equal(pc[:, 3].cpu().numpy().astype(int), lg.argmax(1).numpy())
    np.testing.assert_allclose(pc[:, 1].cpu().numpy(), torch.softmax(lg, 1).gather(1, labels.view(-1, 1))[:, 0].numpy(), rtol=1e-5)


def test_loss_head_vs_golden(ops, golden):
    """torch dialect against the reference's own Losses outputs"""
    g = golden
    lg, labels = torch.from_numpy(g["loss_logits"]).cuda(), torch.from_numpy(g["loss_labels"]).cuda()
    for mode, improve, use_logits in (("improve_prob", True, False), ("improve_logits", True, True), ("ce", False, False)):
        sm, dl, pc = ops.softmax_adv_loss(lg, labels, dialect="torch", improve_loss=improve, use_logits=use_logits, margin=0.05,
                                          mean_scale=1.0 / 4)
        assert pc[:, 0].sum().item() == pytest.approx(float(g[f"loss_flk_{mode}_out"][1]), rel=1e-4)
        torch.testing.assert_close(dl.cpu(), torch.from_numpy(g[f"loss_flk_{mode}_dlogits"]), rtol=2e-4, atol=1e-7)


def test_loss_head_nan_logits_do_not_index_out_of_bounds(ops):
    """NaN logits (corrupt weights upstream) never win an arg-max comparison: the kernel must still read z[] inside the row -- the clip's
    loss and gradient come out NaN, the other clip is untouched (this was a GPU memory fault)"""
    lg = torch.randn(2, 400)
    lg[0] = float("nan")
    labels = torch.tensor([3, 7])
    for dialect, improve, use_logits in (("tf", True, False), ("tf", True, True), ("torch", True, False), ("tf", False, False)):
        sm, dl, pc = ops.softmax_adv_loss(lg.cuda(), labels.cuda(), dialect=dialect, improve_loss=improve, use_logits=use_logits, margin=0.05)
        torch.cuda.synchronize()
        assert torch.isnan(dl[0]).all() and torch.isfinite(dl[1]).all() and torch.isfinite(pc[1]).all()


def test_torch_targeted_improve_loss_refused(ops):
    from flickering_adversarial_video_amd._lib import FlickerHipError
    with pytest.raises(FlickerHipError):
        ops.softmax_adv_loss(torch.zeros(1, 400).cuda(), torch.zeros(1, dtype=torch.int64).cuda(), dialect="torch",
                             improve_loss=True, targeted=True)


@pytest.mark.parametrize("T", [16, 64, 90])
def test_reg_adam_tf(ops, T):
    """TF dialect: loss = g.delta + b0*(b1*norm + b2*diff + b3*lap) on the raw delta; TF-1.15 Adam; 5 steps"""
    rng = np.random.default_rng(T)
    d = torch.from_numpy(rng.uniform(-0.5, 0.5, (T, 1, 1, 3)).astype(np.float32))
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    dg, mg, vg = d.reshape(T, 3).clone().cuda(), torch.zeros(T, 3).cuda(), torch.zeros(T, 3).cuda()
    b0, b1, b2, b3 = 1.3, 0.5, 0.4, 0.7
    for step in range(1, 6):
        gadv = torch.from_numpy(rng.standard_normal((T, 1, 1, 3)).astype(np.float32) * (1e-9 if step == 2 else 1e-2))
        dv = d.clone().requires_grad_(True)
        total, reg = am.tf_total_loss((gadv * dv).sum(), dv, b0, b1, b2, b3)
        (g,) = torch.autograd.grad(total, dv)
        r = am.tf_regularizers(d)
        sc = ops.perturb_reg_adam(gadv.reshape(T, 3).cuda(), dg, mg, vg, step, beta0=b0, beta1=b1, beta2=b2, beta3=b3).cpu()
        np.testing.assert_allclose(sc.numpy()[:6], [reg.item(), r["norm"].item(), r["diff"].item(), r["lap"].item(),
                                                    r["thickness"].item(), r["roughness"].item()], rtol=2e-5)
        assert sc[6].item() == pytest.approx(d.max().item()) and sc[7].item() == pytest.approx(d.min().item())
        d, m, v = am.tf_adam_step(d, g, m, v, step)
        torch.testing.assert_close(dg.cpu(), d.reshape(T, 3), rtol=1e-4, atol=2e-7)


def test_reg_adam_torch_dialect_golden_trajectory(ops, golden):
    """Replays the reference's own 20-step mini attack (tests/golden: Perturbation + Losses + torch.optim.Adam
    on a tiny conv net) with the HIP apply / loss / reduce / Adam kernels; the tiny victim net itself runs on the
    CPU oracle side.  Pins dialect, clamp handling and loop ordering of the fused kernels."""
    import torch.nn.functional as F
    g = golden
    x = torch.from_numpy(g["mini_x"])                            # [1,3,16,6,6]
    xcl = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    tgt = torch.from_numpy(g["mini_target"]).cuda()
    d = torch.from_numpy(g["mini_delta0"]).reshape(3, 16).t().contiguous().cuda()   # [T,3]
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    W1, b1, W2, b2 = (torch.from_numpy(g[k]) for k in ("mini_w1", "mini_b1", "mini_w2", "mini_b2"))
    fw, fb = torch.from_numpy(g["mini_fw"]), torch.from_numpy(g["mini_fb"])
    for step in range(1, 21):
        args = ops.make_apply_args(xcl, d, dialect="torch", dclip=0.2, inv_std=tuple(1.0 / s for s in am.DEFAULT_STD),
                                   lo=am.TORCH_MIN_VALUE, hi=am.TORCH_MAX_VALUE)
        xa = un_s2d(ops.perturb_apply_s2d(args, torch.float32).cpu()).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
        h = F.relu(F.conv3d(xa, W1, b1, padding=1))
        h = F.relu(F.conv3d(h, W2, b2, padding=1))
        out = F.linear(h.mean(dim=(2, 3, 4)), fw, fb)
        sm, dl, pc = ops.softmax_adv_loss(out.detach().cuda(), tgt, dialect="torch", improve_loss=True, use_logits=True, margin=0.05)
        (gx,) = torch.autograd.grad(out, xa, dl.cpu())
        gd = ops.perturb_grad_reduce(args, s2d(gx.permute(0, 2, 3, 4, 1).contiguous()).cuda())
        sc = ops.perturb_reg_adam(gd, d, m, v, step, dialect="torch", beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, dyn_max_norm=0.2)
        loss = pc[:, 0].sum().item() + sc[0].item()
        np.testing.assert_allclose([loss, pc[:, 0].sum().item(), sc[0].item()], g["mini_losses"][step - 1], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(d.cpu().t().reshape(3, 16, 1, 1).numpy(), g["mini_traj"][step - 1], rtol=2e-4, atol=3e-7)


@pytest.mark.parametrize("dialect", ["tf", "torch"])
def test_dense_l12_adam(ops, dialect):
    """dense delta (kinetics_i3d_L12 / model.py:211-214): loss = g.delta + beta * (sum_t sqrt(mean_hwc delta_t^2) + 1e-12); 3 Adam steps"""
    T, H, W = 6, 8, 10
    rng = np.random.default_rng(4)
    d = torch.from_numpy(rng.uniform(-0.1, 0.1, (T, H, W, 3)).astype(np.float32))
    d[2] = 1e-8                                              # a frame at the reference's initial value
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    dg, mg, vg = d.clone().cuda(), m.clone().cuda(), v.clone().cuda()
    beta = 0.7
    for step in range(1, 4):
        gadv = torch.from_numpy(rng.standard_normal((T, H, W, 3)).astype(np.float32) * 1e-3)
        dv = d.clone().requires_grad_(True)
        l12 = am.tf_l12(dv)
        (g,) = torch.autograd.grad((gadv * dv).sum() + beta * l12, dv)
        sc = ops.perturb_dense_l12_adam(gadv.cuda(), dg, mg, vg, step, dialect=dialect, beta=beta).cpu()
        r = am.tf_regularizers(d)
        np.testing.assert_allclose(sc.numpy(), [l12.item(), r["thickness"].item(), r["roughness"].item(), d.abs().max().item()], rtol=2e-5)
        d, m, v = (am.tf_adam_step if dialect == "tf" else am.torch_adam_step)(d, g, m, v, step)
        torch.testing.assert_close(dg.cpu(), d, rtol=2e-4, atol=1e-7)


def test_torch_l12_regulariser_vs_golden(ops, golden):
    """Losses(attack_type='L12') of the reference (model.py:211-214) through the dense Adam kernel (torch dialect): the regulariser
    VALUE against the golden [loss, adv, reg] triplets and its GRADIENT against the golden d(loss)/d(delta) -- recovered from the
    first Adam moment, m_1 = (1 - b1) * g.  Then the product host classes (Perturbation [3,T,H,W], Losses.L12_regularization_loss)."""
    from flickering_adversarial_video_amd.torch_attack import Adversarial_metrics, Losses, Perturbation
    g = golden
    d_ref = torch.from_numpy(g["loss_dense_delta"])                            # [3,16,8,8]
    d = d_ref.permute(1, 2, 3, 0).contiguous().cuda()                          # device layout [T,H,W,3]
    for mode in ("improve_prob", "improve_logits", "ce"):
        reg = float(g[f"loss_dense_{mode}_out"][2])
        dd, m, v = d.clone(), torch.zeros_like(d), torch.zeros_like(d)
        sc = ops.perturb_dense_l12_adam(torch.zeros_like(d), dd, m, v, 1, dialect="torch", beta=1.0, lr=1e-6, dyn_max_norm=0.2)
        assert float(sc[0]) == pytest.approx(reg, rel=2e-5)
        grad = (m / (1 - 0.9)).cpu().permute(3, 0, 1, 2)
        torch.testing.assert_close(grad, torch.from_numpy(g[f"loss_dense_{mode}_ddelta"]), rtol=2e-4, atol=1e-9)
    # clamp: entries beyond dyn_max_norm enter the regulariser clamped and get no regulariser gradient
    dd, m, v = d.clone(), torch.zeros_like(d), torch.zeros_like(d)
    dd[3, 2, 1, 0] = 0.9
    ref = dd.cpu().permute(3, 0, 1, 2).clone().requires_grad_(True)
    l12 = torch.sum(torch.sqrt(torch.mean(ref.clamp(-0.1, 0.1) ** 2, [0, 2, 3]))) + 1e-12
    l12.backward()
    sc = ops.perturb_dense_l12_adam(torch.zeros_like(d), dd, m, v, 1, dialect="torch", beta=1.0, lr=1e-6, dyn_max_norm=0.1)
    assert float(sc[0]) == pytest.approx(l12.item(), rel=2e-5)
    torch.testing.assert_close((m / 0.1).cpu().permute(3, 0, 1, 2), ref.grad, rtol=2e-4, atol=1e-9)
    assert float(m[3, 2, 1, 0]) == 0.0
    # host classes: dense Perturbation keeps the reference's [3,T,H,W] view; Losses('L12') value; metrics on CUDA tensors
    P = Perturbation((3, 16, 8, 8), max_norm=float(g["pert_dense02_max_norm"]))
    P.init_perturbation(g["pert_dense02_delta"])
    clamped, raw = P.get_perturbation()
    np.testing.assert_array_equal(raw.cpu().numpy(), g["pert_dense02_delta"])
    np.testing.assert_array_equal(clamped.cpu().numpy(), g["pert_dense02_clamped"])
    np.testing.assert_allclose([float(t) for t in P.metric_calc()], g["pert_dense02_metric"], rtol=1e-5)
    x = torch.from_numpy(g["pert_x"]).permute(0, 2, 3, 4, 1).contiguous().cuda()
    xs = ops.perturb_apply_s2d(P.apply_args(x, True), torch.float32).cpu()             # fold_t = 1: [B,T,H/2,W/2,16], channel (qh*2+qw)*3+c
    out = xs[..., :12].reshape(2, 16, 4, 4, 2, 2, 3).permute(0, 6, 1, 2, 4, 3, 5).reshape(2, 3, 16, 8, 8)
    torch.testing.assert_close(out, torch.from_numpy(g["pert_dense02_xadv"]), rtol=1e-6, atol=1e-6)
    L = Losses(attack_type="L12")
    assert float(L.L12_regularization_loss(torch.from_numpy(g["loss_dense_delta"]).cuda())) == pytest.approx(float(g["loss_dense_ce_out"][2]), rel=1e-5)
    with pytest.raises(ValueError):
        Losses(attack_type="dense")
    M = Adversarial_metrics()
    miss, valid = M.accuracy_for_eval(torch.from_numpy(g["met_adv"]).cuda(), torch.from_numpy(g["met_gt"]).cuda(), clean_pred=torch.from_numpy(g["met_clean"]).cuda())
    assert [float(miss), float(valid)] == list(g["met_miss_valid"])


def test_torch_surface_classes_vs_golden(golden):
    """the reference's class-level interface against the reference's own outputs: ``Perturbation.forward([x, adversarial])`` /
    ``apply_perturbation`` on NCDHW clips (model.py:80-105) and ``Losses.__call__(labels, logits, prob, delta) -> [loss, adv, reg]`` with
    its ``label_prob`` attribute (model.py:169-175,232), flicker and dense (L12) perturbations."""
    from flickering_adversarial_video_amd.torch_attack import Losses, Perturbation
    g = golden
    x = torch.from_numpy(g["pert_x"]).cuda()                      # [2,3,16,8,8] NCDHW
    for tag in ("flk01", "flk02", "dense02"):
        d = g[f"pert_{tag}_delta"]
        pm = Perturbation(d.shape, max_norm=float(g[f"pert_{tag}_max_norm"]))
        pm.init_perturbation(d)
        out = pm([x, True])
        assert out.shape == x.shape
        torch.testing.assert_close(out.cpu(), torch.from_numpy(g[f"pert_{tag}_xadv"]), rtol=1e-6, atol=1e-6)
        # apply_perturbation / convert_adversarial_video_zero_one (model.py:103-112): numpy [B,T,H,W,3] de-normalised to [0,1]
        z = pm.apply_perturbation(x)
        assert isinstance(z, np.ndarray) and z.shape == (2, 16, 8, 8, 3) and z.dtype == np.float64
        np.testing.assert_allclose(z, g[f"pert_{tag}_zero_one"], rtol=0, atol=5e-7)
        np.testing.assert_allclose(pm.convert_adversarial_video_zero_one(out.permute(0, 2, 3, 4, 1)), g[f"pert_{tag}_zero_one"], rtol=0, atol=5e-7)
        torch.testing.assert_close(pm.forward([x, False]).cpu(), torch.from_numpy(g[f"pert_{tag}_clean"]), rtol=0, atol=0)
        cl = pm([x.permute(0, 2, 3, 4, 1).contiguous(), True])     # channels-last in, channels-last out
        torch.testing.assert_close(cl.permute(0, 4, 1, 2, 3).cpu(), torch.from_numpy(g[f"pert_{tag}_xadv"]), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(pm.get_perturbation()[0].cpu(), torch.from_numpy(g[f"pert_{tag}_clamped"]), rtol=0, atol=0)
    lg, labels = torch.from_numpy(g["loss_logits"]).cuda(), torch.from_numpy(g["loss_labels"]).cuda()
    prob = torch.softmax(lg, 1)
    for kind, attack_type in (("flk", "flickering"), ("dense", "L12")):
        delta = torch.from_numpy(g[f"loss_{kind}_delta"]).cuda()
        for mode, improve, use_logits in (("improve_prob", True, False), ("improve_logits", True, True), ("ce", False, False)):
            crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=improve, logits=use_logits, attack_type=attack_type)
            loss, adv, reg = crit(labels, lg, prob, delta)
            ref = g[f"loss_{kind}_{mode}_out"]
            np.testing.assert_allclose([float(loss), float(adv), float(reg)], ref, rtol=1e-4, atol=1e-7, err_msg=f"{kind} {mode}")
            np.testing.assert_allclose(crit.label_prob.cpu().numpy().reshape(-1), g[f"loss_{kind}_{mode}_label_prob"].reshape(-1), rtol=1e-5)


def test_video_learner_adversarial_name_and_keywords():
    """VideoLearnerAdversarial(dataset, num_classes, base_model, sample_length, cyclic_pert, l_inf_pert_norm, attack_type,
    labaels_id_to_text) (model.py:337-347) constructs the HIP engine and runs an iteration; .pert_model / .model_name / .results exist"""
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import Losses, VideoLearnerAdversarial
    W = vs.synthetic_weights("r3d_18", 7)
    learner = VideoLearnerAdversarial(None, num_classes=400, base_model="r3d_18", sample_length=8, cyclic_pert=False, l_inf_pert_norm=0.1,
                                      attack_type="flickering", labaels_id_to_text={0: "a"}, weights=W, image_size=32, dtype="f32")
    assert learner.model_name == "r3d_18" and learner.results == {} and learner.pert_model.size == (3, 8, 1, 1)
    assert learner.pert_model.max_norm == 0.1
    x = torch.from_numpy(vs.synthetic_clip(1, 8, 32, 32, seed=3)).cuda()
    y = learner.logits(x).argmax(-1).clone()
    r = learner.step(x, y, Losses(improve_loss=True)).host()
    assert np.isfinite(r["adv_loss"]) and float(learner.pert_model.perturbation.abs().max()) > 0
    with pytest.raises(ValueError):
        VideoLearnerAdversarial(None, base_model="r3d_18")

"""Pin the CPU oracle (oracle/attack_math.py) against golden vectors produced by the reference's own
classes (tests/golden/make_golden.py): Perturbation / Losses / Adversarial_metrics / Adam loop."""
import numpy as np
import pytest
import torch

from oracle import attack_math as am

RTOL, ATOL = 1e-5, 1e-6


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["flk01", "flk02", "dense02"])
def test_perturbation_forward_and_grad(golden, tag):
    g = golden
    x, w = T(g["pert_x"]), T(g["pert_w"])
    d = T(g[f"pert_{tag}_delta"]).clone().requires_grad_(True)
    mn = float(g[f"pert_{tag}_max_norm"])
    out = am.torch_apply(x, d, mn)
    np.testing.assert_allclose(out.detach().numpy(), g[f"pert_{tag}_xadv"], rtol=RTOL, atol=ATOL)
    (out * w).sum().backward()
    np.testing.assert_allclose(d.grad.numpy(), g[f"pert_{tag}_grad"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(am.torch_apply(x, d, mn, adversarial=False).numpy(), g[f"pert_{tag}_clean"])
    th, ro = am.torch_metrics(d.detach())
    np.testing.assert_allclose([th.item(), ro.item()], g[f"pert_{tag}_metric"], rtol=1e-5)
    np.testing.assert_allclose(d.detach().clamp(-mn, mn).numpy(), g[f"pert_{tag}_clamped"])


def test_clamp_bounds(golden):
    assert am.TORCH_MIN_VALUE == pytest.approx(float(golden["pert_min_value"]), rel=1e-12)
    assert am.TORCH_MAX_VALUE == pytest.approx(float(golden["pert_max_value"]), rel=1e-12)
    assert am.TORCH_MIN_VALUE == pytest.approx(-1.73488, abs=1e-5)
    assert am.TORCH_MAX_VALUE == pytest.approx(2.49020, abs=1e-5)


@pytest.mark.parametrize("dtag,atype", [("flk", "flickering"), ("dense", "L12")])
@pytest.mark.parametrize("mode,improve,use_logits", [("improve_prob", True, False),
                                                     ("improve_logits", True, True), ("ce", False, False)])
def test_losses(golden, dtag, atype, mode, improve, use_logits):
    g = golden
    lg = T(g["loss_logits"]).clone().requires_grad_(True)
    d = T(g[f"loss_{dtag}_delta"]).clone().requires_grad_(True)
    labels = T(g["loss_labels"])
    loss, adv, reg = am.torch_losses(labels, lg, torch.softmax(lg, 1), d, 0.5, 1.0, 0.05, improve, use_logits, atype)
    key = f"loss_{dtag}_{mode}"
    np.testing.assert_allclose([loss.item(), adv.item(), reg.item()], g[key + "_out"], rtol=2e-5, atol=1e-7)
    loss.backward()
    np.testing.assert_allclose(lg.grad.numpy(), g[key + "_dlogits"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(d.grad.numpy(), g[key + "_ddelta"], rtol=1e-4, atol=1e-9)


def test_losses_alt_hyper(golden):
    g = golden
    lg = T(g["loss_logits"]).clone().requires_grad_(True)
    d = T(g["loss_flk_delta"]).clone().requires_grad_(True)
    loss, adv, reg = am.torch_losses(T(g["loss_labels"]), lg, torch.softmax(lg, 1), d, 0.3, 2.5, 0.1, True, False)
    np.testing.assert_allclose([loss.item(), adv.item(), reg.item()], g["loss_alt_out"], rtol=2e-5)
    loss.backward()
    np.testing.assert_allclose(lg.grad.numpy(), g["loss_alt_dlogits"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(d.grad.numpy(), g["loss_alt_ddelta"], rtol=1e-4, atol=1e-9)


def test_targeted_improve_loss_raises():
    lg = torch.zeros(1, 400)
    with pytest.raises(NotImplementedError):
        am.torch_losses(torch.zeros(1, dtype=torch.long), lg, torch.softmax(lg, 1), torch.zeros(3, 4, 1, 1), targeted=True)


def test_metrics(golden):
    g = golden
    miss, valid = am.fooling_counts(T(g["met_adv"]), T(g["met_clean"]), T(g["met_gt"]))
    assert [miss, valid] == list(g["met_miss_valid"].astype(int))
    th, ro = am.torch_metrics(T(g["met_delta"]))
    np.testing.assert_allclose([th.item(), ro.item()], g["met_thick_rough"], rtol=1e-5)


def _mini_net(g, x):
    import torch.nn.functional as F
    h = F.relu(F.conv3d(x, T(g["mini_w1"]), T(g["mini_b1"]), padding=1))
    h = F.relu(F.conv3d(h, T(g["mini_w2"]), T(g["mini_b2"]), padding=1))
    return F.linear(h.mean(dim=(2, 3, 4)), T(g["mini_fw"]), T(g["mini_fb"]))


def test_mini_attack_trajectory(golden):
    """20 steps of the single-video loop (model.py:1056-1101) with the oracle's torch-dialect apply,
    Losses and hand-written torch-1.4 Adam -> must reproduce the reference delta trajectory."""
    g = golden
    x, tgt = T(g["mini_x"]), T(g["mini_target"])
    d = T(g["mini_delta0"]).clone()
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    for t in range(1, 21):
        dv = d.clone().requires_grad_(True)
        out = _mini_net(g, am.torch_apply(x, dv, 0.2))
        loss, adv, reg = am.torch_losses(tgt, out, torch.softmax(out, 1), dv.clamp(-0.2, 0.2), 0.5, 1.0, 0.05,
                                         True, True, "flickering")
        loss.backward()
        np.testing.assert_allclose([loss.item(), adv.item(), reg.item()], g["mini_losses"][t - 1], rtol=1e-4, atol=1e-7)
        d, m, v = am.torch_adam_step(d, dv.grad, m, v, t)
        np.testing.assert_allclose(d.numpy(), g["mini_traj"][t - 1], rtol=1e-4, atol=2e-7)


# ---- TF dialect: cross-checks against the shared formulas + hand-derived known answers ----
def test_tf_regs_match_torch_formulas(golden):
    """norm/diff/lap are the same formulas in both stacks (kinetics_i3d_utils.py:177-186 vs
    model.py:198-209); the torch-side value is pinned by the reference, so pin the TF-side with it."""
    d_t = T(golden["loss_flk_delta"])                  # [3,T,1,1]
    d_tf = d_t.permute(1, 2, 3, 0).contiguous()        # [T,1,1,3]
    r = am.tf_regularizers(d_tf)
    ref = am.torch_flicker_reg(d_t, 0.5)
    assert (0.5 * r["norm"] + 0.5 * (r["diff"] + r["lap"])).item() == pytest.approx(ref.item(), rel=1e-6)
    th, ro = am.torch_metrics(d_t)
    assert r["thickness"].item() * 100 == pytest.approx(th.item(), rel=1e-6)
    assert r["roughness"].item() * 100 == pytest.approx(ro.item(), rel=1e-6)


def test_tf_regs_known_answer():
    d = torch.tensor([1.0, 0.0, 0.0, 0.0]).view(4, 1, 1, 1).repeat(1, 1, 1, 3)
    r = am.tf_regularizers(d)
    assert r["norm"].item() == pytest.approx(0.25 + 1e-12)
    assert r["diff"].item() == pytest.approx(0.5)       # d = [1,-1,0,0]
    assert r["lap"].item() == pytest.approx((4 + 1 + 1) / 4)  # l = [-2,1,0,1]
    assert r["thickness_pct"].item() == pytest.approx(12.5)
    assert r["roughness_pct"].item() == pytest.approx(25.0)


def test_reg_closed_form_grads():
    torch.manual_seed(0)
    d = torch.randn(9, 1, 1, 3, requires_grad=True)
    r = am.tf_regularizers(d)
    gn, gd, gl = am.reg_grads_closed_form(d.detach())
    for val, g in ((r["norm"], gn), (r["diff"], gd), (r["lap"], gl)):
        (ag,) = torch.autograd.grad(val, d, retain_graph=True)
        np.testing.assert_allclose(ag.numpy(), g.numpy(), rtol=1e-5, atol=1e-7)


def test_tf_improve_prob_matches_torch_dialect(golden):
    """prob-mode improve loss is the same formula in both stacks (max(p - onehot) == max over k != y
    whenever some non-label prob exceeds p_y - 1, i.e. always)."""
    lg, labels = T(golden["loss_logits"]), T(golden["loss_labels"])
    tf_loss, to_min, to_max = am.tf_improve_adversarial_loss(lg, labels, 0.05, False, False)
    assert tf_loss.item() == pytest.approx(float(golden["loss_flk_improve_prob_out"][1]), rel=2e-5)
    np.testing.assert_allclose(to_min.numpy(), golden["loss_flk_improve_prob_label_prob"], rtol=1e-5)
    tf_ce, _, _ = am.tf_ce_adversarial_loss(lg, labels, False)
    assert tf_ce.item() == pytest.approx(float(golden["loss_flk_ce_out"][1]), rel=2e-5)


def test_tf_logits_mode_quirk():
    """max_non_label_logits = max(logits - onehot) does not exclude the label."""
    lg = torch.tensor([[5.0, 1.0, 0.0]])
    s = am.tf_label_stats(lg, torch.tensor([0]))
    assert s["max_non_label_logits"].item() == pytest.approx(4.0)   # label logit minus one, not 1.0
    assert s["max_non_label_prob"].item() == pytest.approx(torch.softmax(lg, 1)[0, 1].item())


def test_tf_apply_clip_roll_and_grad():
    torch.manual_seed(1)
    x = torch.rand(2, 5, 2, 2, 3) * 2 - 1
    d = torch.tensor([0.5, -0.5, 0.1, 0.39, -0.41]).view(5, 1, 1, 1).repeat(1, 1, 1, 3).requires_grad_(True)
    out = am.tf_apply(x, d)
    exp = torch.clamp(x + d.detach().clamp(-0.4, 0.4), -1, 1)
    np.testing.assert_array_equal(out.detach().numpy(), exp.numpy())
    out.sum().backward()
    assert d.grad[0].abs().sum() == 0 and d.grad[1].abs().sum() == 0 and d.grad[4].abs().sum() == 0
    inside = ((x + d.detach().clamp(-0.4, 0.4)).abs() <= 1).float().sum(dim=(0, 2, 3))
    np.testing.assert_allclose(d.grad[2].reshape(-1).numpy(), inside[2].numpy())
    r = am.tf_apply(x, d.detach(), shift_x=2, cyclic_flag=1.0)
    np.testing.assert_array_equal(r.numpy(), torch.clamp(torch.roll(x, 2, 1) + d.detach().clamp(-0.4, 0.4), -1, 1).numpy())


def test_adam_dialects_differ_only_in_eps_placement():
    g = torch.tensor([1e-9, 1e-3, 1.0])
    z = torch.zeros(3)
    a, _, _ = am.tf_adam_step(z, g, z, z, 1)
    b, _, _ = am.torch_adam_step(z, g, z, z, 1)
    np.testing.assert_allclose(a[1:].numpy(), b[1:].numpy(), rtol=1e-3)  # g=1e-3: eps shifts by ~3e-4 rel
    assert abs(a[0].item() - b[0].item()) > 1e-5     # eps matters for tiny gradients
    # against torch.optim.Adam (same formula as torch 1.4 for amsgrad=False, weight_decay=0)
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([p], lr=1e-3)
    p.grad = g.clone()
    opt.step()
    np.testing.assert_allclose(p.detach().numpy(), b.numpy(), rtol=1e-5, atol=1e-12)

"""End-to-end parity of the VideoResNet (r2plus1d_18 / r3d_18 / mc3_18) attack iteration on the GPU against the CPU oracle
(oracle/videoresnet_ref.py + torch-dialect attack maths), same methodology as tests/test_i3d_gpu.py: smooth quantities
at 1e-3 against the fp32 oracle; gradients against the fp64 oracle, no worse than the fp32 CPU oracle itself."""
import numpy as np
import pytest
import torch

from oracle import attack_math as am
from oracle import videoresnet_ref as vr

pytestmark = pytest.mark.gpu
T, HW = 8, 112


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def oracle_pass(W, x_cl, delta_ct, arch, dt):
    Wd = {k: torch.from_numpy(v).to(dt) for k, v in W.items()}
    x = x_cl.to(dt).permute(0, 4, 1, 2, 3).contiguous()
    d = delta_ct.to(dt).clone().requires_grad_(True)            # [3,T,1,1]
    logits, ep = vr.videoresnet_logits(am.torch_apply(x, d, 0.2), Wd, arch, return_endpoints=True)
    label = logits.argmax(-1)
    prob = torch.softmax(logits, 1)
    loss, adv, reg = am.torch_losses(label, logits, prob, d.clamp(-0.2, 0.2), 0.5, 1.0, 0.05, True, True, "flickering")
    names = [n for n in ep if n.startswith("layer")]
    g, *ge = torch.autograd.grad(adv, [d] + [ep[n] for n in names], retain_graph=True)
    (gtot,) = torch.autograd.grad(loss, d)
    return dict(logits=logits.detach(), ep={k: v.detach() for k, v in ep.items()}, adv=adv.item(), reg=reg.item(), label=label,
                g=g, gtot=gtot, ge=dict(zip(names, ge)))


@pytest.mark.parametrize("arch", ["r2plus1d_18", "r3d_18", "mc3_18"])
def test_videoresnet_forward_backward(arch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses
    W = vs.synthetic_weights(arch, 42)
    x_cl = torch.from_numpy(vs.synthetic_clip(1, T, HW, HW, seed=5))
    rng = np.random.default_rng(2)
    delta = torch.from_numpy(rng.uniform(-0.05, 0.05, (3, T, 1, 1)).astype(np.float32))
    delta[:, 2] = 0.25                                      # beyond dynamic_max_norm = 0.2: no gradient through the clamp
    r32, r64 = (oracle_pass(W, x_cl, delta, arch, dt) for dt in (torch.float32, torch.float64))
    for dtype in ("f32", "bf16"):
        f32 = dtype == "f32"
        eng = FlickerVideoResNet(arch, W, batch_size=1, sample_length=T, image_size=HW, dtype=dtype, l_inf_pert_norm=0.2)
        eng.pert_model.init_perturbation(delta.numpy())
        crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True, logits=True)
        res = eng.step(x_cl.cuda(), r32["label"].cuda(), crit, update=False)
        for name, ref in r32["ep"].items():
            e = rel_err(torch.from_numpy(eng.net.activation(name)), ref.permute(0, 2, 3, 4, 1))
            print(f"[{arch} {dtype}] {name}: max rel err {e:.3e}")
            assert e < (1e-3 if f32 else 8e-2), name
        e = rel_err(eng._logits.cpu(), r32["logits"])
        print(f"[{arch} {dtype}] logits: max rel err {e:.3e}")
        assert e < (1e-3 if f32 else 5e-2)
        assert float(res["adv_loss"]) == pytest.approx(r32["adv"], rel=1e-3 if f32 else 1e-1, abs=1e-5)
        for name in reversed(list(r64["ge"])):
            def masked(r):
                g = r["ge"][name].permute(0, 2, 3, 4, 1)
                return torch.where(r["ep"][name].permute(0, 2, 3, 4, 1) > 0, g, torch.zeros_like(g))
            e_hip = rel_l2(torch.from_numpy(eng.net.activation("grad:" + name)), masked(r64))
            e_cpu = rel_l2(masked(r32), masked(r64))
            print(f"[{arch} {dtype}] grad:{name}: rel-L2 vs fp64: HIP {e_hip:.3e} (fp32 CPU oracle {e_cpu:.3e})")
            assert e_hip < (max(3 * e_cpu, 0.1) if f32 else 0.9), name
        g = eng._red[:3 * T].view(T, 3).cpu().t().reshape(3, T, 1, 1)
        e_hip, e_cpu = rel_err(g, r64["g"]), rel_err(r32["g"], r64["g"])
        cos = float(torch.nn.functional.cosine_similarity(g.double().flatten(), r64["g"].flatten(), 0))
        print(f"[{arch} {dtype}] d(adv)/d(delta) vs fp64: HIP {e_hip:.3e} (fp32 CPU {e_cpu:.3e}) cosine {cos:.6f}")
        assert g[:, 2].abs().max() == 0
        if f32:
            assert e_hip < 3 * e_cpu + 5e-3 and cos > 0.999
        else:
            assert cos > 0.8
        # one real update: torch-Adam on (adv + lambda*reg) -- first step is a pure sign step of the total gradient
        before = eng.pert_model.perturbation.clone()
        eng.step(x_cl.cuda(), r32["label"].cuda(), crit)
        moved = (eng.pert_model.perturbation - before).cpu().t().reshape(3, T, 1, 1)
        if f32:
            big = r64["gtot"].abs() > 0.05 * r64["gtot"].abs().max()
            assert (torch.sign(moved[big]) == -torch.sign(r64["gtot"][big]).float()).all()
        del eng


@pytest.mark.gpu
def test_videoresnet_drivers():
    """fit_single_video_attack (restart-with-1.3x-norm schedule, model.py:1056-1066) and fit / train_an_epoch (StepLR,
    model.py:496-497,571-573) around the step: loop control, result keys, schedule; the evaluation-pass regulariser equals
    the kernel's value."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import Adversarial_metrics, FlickerVideoResNet, Losses
    arch = "r3d_18"
    W = vs.synthetic_weights(arch, 42)
    x = torch.from_numpy(vs.synthetic_clip(1, T, HW, HW, seed=5)).cuda()
    eng = FlickerVideoResNet(arch, W, batch_size=1, sample_length=T, image_size=HW, dtype="f32", l_inf_pert_norm=0.2)
    label = eng.logits(x, False).argmax(1).clone()
    crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True, logits=True)
    # a wrong label -> None (model.py:1030-1032)
    assert eng.fit_single_video_attack(x, (label + 1) % 400, crit) is None
    # evaluation pass and update pass report the same regulariser for the same delta
    r_eval = eng.step(x, label, crit, update=False)
    r_upd = eng.step(x, label, crit, lr=0.0)
    assert float(r_eval["reg_loss"]) == pytest.approx(float(r_upd["reg_loss"]), rel=1e-4, abs=1e-14)
    # restart schedule: a tiny lr never fools the net -> steps 0..restart_after of each chance, norm grown by 1.3 per restart
    eng.pert_model.dynamic_max_norm = 0.2
    res = eng.fit_single_video_attack(x, label, crit, lr=1e-9, n_iter=2, restart_after=2, max_restarts=3)
    assert res["restarts"] == 3 and not any(res["is_adversarial"])
    assert eng.pert_model.dynamic_max_norm == pytest.approx(0.2 * 1.3 ** 3)
    assert len(res["loss/total"]) == len(res["perturbation"]) == len(res["is_adversarial"]) == 9     # 3 chances x steps 0,1,2
    for k in ("loss/total", "loss/adv_loss", "loss/reg_loss", "perturbation/thickness", "perturbation/roughness",
              "perturbation/inf_norm", "perturbation", "prob_clean_input", "label", "is_adversarial"):
        assert k in res
    assert res["perturbation"][0].shape == (3, T, 1, 1)
    # a real attack terminates once n_iter steps are done and the clip is fooled
    eng.pert_model.dynamic_max_norm = 0.2
    eng.pert_model.init_perturbation()
    res = eng.fit_single_video_attack(x, label, crit, lr=2e-2, n_iter=3, restart_after=200, max_restarts=2)
    assert len(res["loss/total"]) >= 3 and (res["is_adversarial"][-1] or res["restarts"] == 2)
    # epochs: StepLR with the default step = ceil(2/3 epochs)
    dl = {"train": [(x, label, None)] * 2, "valid": [(x, label, None)]}
    out = eng.fit(dl, crit, Adversarial_metrics(), lr=1e-3, epochs=3, lr_gamma=0.1)
    assert [r["lr"] for r in out] == pytest.approx([1e-3, 1e-3, 1e-4])
    for r in out:
        for ph in ("train", "valid"):
            assert 0.0 <= r[f"{ph}/fooling_ratio"] <= 1.0 and np.isfinite(r[f"{ph}/loss"]) and r[f"{ph}/perturbation"].shape == (3, T, 1, 1)

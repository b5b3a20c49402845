"""End-to-end parity of the VideoResNet (r2plus1d_18 / r3d_18 / mc3_18) attack iteration on the GPU against the CPU oracle
(oracle/videoresnet_ref.py + torch-dialect attack maths) at BASELINE config 3's size (16 x 112 x 112), same methodology as
tests/test_i3d_gpu.py: smooth quantities at 1e-3 against the fp32 oracle; the backward pass LINK BY LINK (each residual block's
backward fed with the HIP path's own output gradient and evaluated on its own input endpoint) at 1e-3 of the buffer maximum for
all but a counted few elements (neighbourhoods of ReLU decisions that differ between two fp32 implementations)."""
import numpy as np
import pytest
import torch

from oracle import attack_math as am
from oracle import videoresnet_ref as vr

pytestmark = pytest.mark.gpu
T, HW = 16, 112


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
        # (bf16: the improve loss on logits is a difference of two logits + margin -- 2-Lipschitz in the max norm of the logits -- and on
        #  this random-sign fixture the two are close: the bound is derived from the measured logit error, not from the loss's size)
        abs_l = float((eng._logits.cpu() - r32["logits"]).abs().max())
        assert float(res["adv_loss"]) == pytest.approx(r32["adv"], rel=1e-3, abs=1e-5) if f32 else abs(float(res["adv_loss"]) - r32["adv"]) <= 2 * abs_l + 1e-5
        # ---- backward, link by link (see tests/test_i3d_gpu.py): stem <- layer1.0 <- ... <- layer4.1 <- logits ----
        Wd = {k: (torch.from_numpy(v).to(torch.bfloat16).double() if (not f32 and v.ndim == 5) else torch.from_numpy(v).double()) for k, v in W.items()}
        # (bf16: this oracle rounds the block ENDPOINTS to bf16 only; the HIP path also stores the activations inside a block in
        # bf16 -- the (2+1)D mid tensors of r2plus1d_18 above all: measured rel-L2 4e-2 there, 3.5e-3 for r3d_18 / mc3_18;
        # fp32: measured 1e-6, with up to 7e-4 of the elements of two r2plus1d_18 links next to a flipped ReLU decision)
        FWD_TOL, BWD_TOL, BWD_FRAC, BWD_L2 = (1e-4, 1e-3, 3e-3, 1e-2) if f32 else (3e-2, 5e-2, 2e-2, 8e-2)
        hip_act = lambda n: torch.from_numpy(eng.net.activation(n)).permute(0, 4, 1, 2, 3).contiguous().double()
        links = [("stem", lambda x, fr: vr.stem(x, Wd, arch, fr))]
        links += [(n, lambda x, fr, a=(n, kind, st, ds): vr.basic_block(x, Wd, a[0], a[1], a[2], a[3], fr)) for n, kind, st, ds in vr.blocks(arch)]
        links.append(("logits", lambda x, fr: torch.nn.functional.linear(x.mean(dim=(2, 3, 4)), Wd["fc.weight"], Wd["fc.bias"])))
        d0 = delta.double().clone().requires_grad_(True)
        xa = am.torch_apply(x_cl.double().permute(0, 4, 1, 2, 3).contiguous(), d0, 0.2)
        # bf16: the stem reads the perturbed clip as TWO bf16 numbers per value (hi = bf16(x_adv), lo = bf16(x_adv - hi); fold_t = 4):
        # value hi + lo, gradient x_adv's (straight through)
        hi16 = xa.detach().to(torch.bfloat16).double()
        prev_name, prev = "delta", (hi16 + (xa.detach() - hi16).to(torch.bfloat16).double() + (xa - xa.detach())) if not f32 else xa
        for name, fn in links:
            with torch.no_grad():
                out = fn(prev, True)
            pre = fn(prev, False)
            got_f, got_g = (eng._logits.cpu().double(), eng._dl.cpu().double()) if name == "logits" else (hip_act(name), hip_act("grad:" + name))
            e_f = rel_err(out, got_f)
            assert e_f < FWD_TOL, f"forward link {prev_name} -> {name}: {e_f:.3e}"
            if prev_name == "delta":
                (g_ref,) = torch.autograd.grad(pre, d0, grad_outputs=got_g)
                g_hip = eng._red[:3 * T].view(T, 3).cpu().t().reshape(3, T, 1, 1).double()
                assert rel_err(g_hip, g_ref) < (1e-3 if f32 else 3e-2), f"d(adv)/d(delta) link: {rel_err(g_hip, g_ref):.3e}"
            else:
                (g_ref,) = torch.autograd.grad(pre, prev, grad_outputs=got_g)
                g_ref = torch.where(prev > 0, g_ref, torch.zeros_like(g_ref))
                g_hip = hip_act("grad:" + prev_name)
            e_l2, frac = rel_l2(g_hip, g_ref), float(((g_hip - g_ref).abs() > BWD_TOL * g_ref.abs().max()).double().mean())
            print(f"[{arch} {dtype}] link {prev_name:>9s} -> {name:<9s} fwd max-rel {e_f:.2e} | bwd rel-L2 {e_l2:.2e}, elements beyond {BWD_TOL:g} of max: {frac:.2e}")
            assert frac <= BWD_FRAC and e_l2 < BWD_L2, f"backward link {name} -> {prev_name}"
            prev_name = name
            if name != "logits":
                prev = hip_act(name).requires_grad_(True)
        g = eng._red[:3 * T].view(T, 3).cpu().t().reshape(3, T, 1, 1)
        e_hip, e_cpu = rel_err(g, r64["g"]), rel_err(r32["g"], r64["g"])
        cos = float(torch.nn.functional.cosine_similarity(g.double().flatten(), r64["g"].flatten(), 0))
        print(f"[{arch} {dtype}] d(adv)/d(delta) end to end vs fp64: HIP {e_hip:.3e} (fp32 CPU {e_cpu:.3e}) cosine {cos:.6f}")
        assert g[:, 2].abs().max() == 0
        # (bf16, measured: r2plus1d_18 0.946, r3d_18 0.992, mc3_18 0.988 -- accumulated bf16 rounding of the gradient through 20 / 40 layers on
        #  the random-sign fixture; the links above bound every layer's own error)
        assert cos > (0.999 if f32 else 0.92 if arch == "r2plus1d_18" else 0.975)
        # one real update: torch-Adam on (adv + lambda*reg) -- first step is a pure sign step of the total gradient
        before = eng.pert_model.perturbation.clone()
        eng.step(x_cl.cuda(), r32["label"].cuda(), crit)
        moved = (eng.pert_model.perturbation - before).cpu().t().reshape(3, T, 1, 1)
        if f32:
            big = r64["gtot"].abs() > 0.05 * r64["gtot"].abs().max()
            assert (torch.sign(moved[big]) == -torch.sign(r64["gtot"][big]).float()).all()
        del eng


@pytest.mark.parametrize("arch,dtype", [("r2plus1d_18", "f32"), ("r2plus1d_18", "bf16"), ("mc3_18", "bf16"), ("r3d_18", "bf16")])
def test_videoresnet_attack_trajectory_well_conditioned(arch, dtype):
    """BASELINE config 3 (r2plus1d_18, bs 1, 16 x 112 x 112, torch dialect) at the north-star bar: logits, adversarial loss and the
    LEARNED DELTA of the fp32 mode within 1e-3 of the reference maths (fp64 oracle) over 6 iterations of the single-video loop
    (model.py:1073-1101: Perturbation -> net -> Losses -> backward -> torch-Adam) on the well-conditioned fixture
    (oracle/fixtures.py::coherent_videoresnet_weights).  The fixture is first shown to be well-conditioned: the torch-CPU fp32
    oracle reproduces the fp64 trajectory to < 1e-4."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses
    from oracle import fixtures
    steps, LR = 6, 1e-3
    x_cl = torch.from_numpy(vs.synthetic_clip(1, T, HW, HW, seed=5))
    x = x_cl.permute(0, 4, 1, 2, 3).contiguous()
    W = fixtures.coherent_videoresnet_weights(vs.synthetic_weights(arch, 42), x, arch, label=233)

    def trajectory(dt):
        Wd = {k: torch.from_numpy(v).to(dt) for k, v in W.items()}
        xd = x.to(dt)
        label = vr.videoresnet_logits(xd, Wd, arch).argmax(-1)
        d = torch.zeros(3, T, 1, 1, dtype=dt)
        m, v = torch.zeros_like(d), torch.zeros_like(d)
        out = []
        for it in range(1, steps + 1):
            dv = d.clone().requires_grad_(True)
            logits = vr.videoresnet_logits(am.torch_apply(xd, dv, 0.2), Wd, arch)
            loss, adv, reg = am.torch_losses(label, logits, torch.softmax(logits, 1), dv.clamp(-0.2, 0.2), 0.5, 1.0, 0.05, True, True, "flickering")
            (g,) = torch.autograd.grad(loss, dv)
            d, m, v = am.torch_adam_step(d, g, m, v, it, lr=LR)
            out.append(dict(logits=logits.detach().double(), adv=adv.item(), delta=d.detach().double().clone()))
        return label, out

    label, t64 = trajectory(torch.float64)
    label32, t32 = trajectory(torch.float32)
    assert int(label) == int(label32) == 233
    for it in range(steps):
        assert rel_err(t32[it]["delta"], t64[it]["delta"]) < 1e-4, "fixture is not well-conditioned"
    # bf16 (the dtype config 3 is benchmarked in; mc3_18 is the reference's universal-attack model, r2plus1d_main_universal_attack.py:
    # 30-33): the same trajectory at stated bf16 tolerances.  Round 3 (the clip rounded to ONE bf16 per value) measured delta 2.5e-2, logits
    # 9e-2 -- at delta = 0 already: it was the rounding of the CLIP, on a fixture whose logits are cancelling sums -- against bars of 4e-2 /
    # 1.5e-1.  Round 4: the bf16 plans read the clip as two bf16 numbers per value (fold_t = 4, ~16 bits).  Measured on MI355X:
    #   r2plus1d_18  delta 6.9e-3   logits 1.6e-2   adversarial loss 1.0e-4      (the (2+1)D mid tensors: twice the bf16 layers)
    #   mc3_18       delta 1.1e-4   logits 3.4e-4   adversarial loss 3.2e-6
    #   r3d_18       delta 1.0e-3   logits 1.2e-3   adversarial loss 7.9e-7
    TOL_D, TOL_L, TOL_A = (1e-3, 1e-3, 1e-3) if dtype == "f32" else (1e-2, 2.5e-2, 5e-4) if arch == "r2plus1d_18" else (3e-3, 4e-3, 5e-5)
    eng = FlickerVideoResNet(arch, W, batch_size=1, sample_length=T, image_size=HW, dtype=dtype, l_inf_pert_norm=0.2)
    eng.pert_model.init_perturbation(np.zeros((3, T, 1, 1), np.float32))      # (the default start is U(-1,1)*1e-6, model.py:121-126)
    crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True, logits=True)
    worst = [0.0, 0.0, 0.0]
    for it in range(steps):
        res = eng.step(x_cl.cuda(), label.cuda(), crit, lr=LR).host()
        delta = eng.pert_model.perturbation.cpu().t().reshape(3, T, 1, 1)
        e_d, e_l = rel_err(delta, t64[it]["delta"]), rel_err(eng._logits.cpu(), t64[it]["logits"])
        e_a = abs(float(res["adv_loss"]) - t64[it]["adv"]) / max(abs(t64[it]["adv"]), 1e-7)
        worst = [max(a, b) for a, b in zip(worst, (e_d, e_l, e_a))]
        print(f"[{dtype}] iter {it + 1}: adv {res['adv_loss']:.7f} (fp64 oracle {t64[it]['adv']:.7f}); delta max-rel {e_d:.2e}; logits max-rel {e_l:.2e} "
              f"(torch-CPU fp32 oracle delta {rel_err(t32[it]['delta'], t64[it]['delta']):.2e})")
    print(f"[{arch} {dtype}] worst over {steps} iterations: delta {worst[0]:.2e}, logits {worst[1]:.2e}, adversarial loss {worst[2]:.2e}")
    assert worst[0] < TOL_D and worst[1] < TOL_L and worst[2] < TOL_A


def test_videoresnet_bf16_small_delta_reaches_the_logits():
    """How small a flicker perturbation still moves the logits through the bf16 stem input?  The normalised clip (u8/255 - mean)/std
    is NOT exactly representable in bf16 (unlike I3D's u8/128 - 1), so x + delta/std is rounded per pixel with the pixel's own offset
    inside its ulp: a delta far below half an ulp still flips the fraction delta/ulp of the pixels, and the change of the logits
    -- a sum over 2e5 pixels per frame -- follows the fp32 change.  Measured here for |delta| from 1e-4 up to the reference's start value
    (U(+-0.005), model.py:946-948) and beyond: relative error of the bf16 logit CHANGE against the fp32 engine's, on the
    well-conditioned fixture.
    Round 4: the bf16 plans read the clip as TWO bf16 numbers per value (flk_apply_args.fold_t = 4: hi + lo against the same stem weights),
    i.e. x + delta/std to ~16 bits, so the bf16 logit CHANGE follows the fp32 engine's from |delta| = 1e-4 (where one bf16 per value
    moved the logits with the WRONG SIGN) upwards.  Asserted at every amplitude: right direction (cosine > 0.9) and within 25 %."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet
    from oracle import fixtures
    arch = "r2plus1d_18"
    x_cl = torch.from_numpy(vs.synthetic_clip(1, T, HW, HW, seed=5))
    W = fixtures.coherent_videoresnet_weights(vs.synthetic_weights(arch, 42), x_cl.permute(0, 4, 1, 2, 3).contiguous(), arch, label=233)
    sign = np.sign(np.random.default_rng(8).standard_normal((3, T, 1, 1))).astype(np.float32)
    amps = (1e-4, 3e-4, 1e-3, 5e-3, 2e-2)
    change = {}
    for dtype in ("f32", "bf16"):
        eng = FlickerVideoResNet(arch, W, batch_size=1, sample_length=T, image_size=HW, dtype=dtype, l_inf_pert_norm=0.2)
        eng.pert_model.init_perturbation(np.zeros((3, T, 1, 1), np.float32))
        base = eng.logits(x_cl.cuda(), adversarial=True).clone()          # delta = 0, same clamp of the clip to [min_v, max_v]
        change[dtype] = []
        for a in amps:
            eng.pert_model.init_perturbation(a * sign)
            change[dtype].append((eng.logits(x_cl.cuda(), adversarial=True) - base).cpu().double())
        del eng
    errs = []
    for a, c32, c16 in zip(amps, change["f32"], change["bf16"]):
        e = float((c16 - c32).norm() / c32.norm())
        cos = float(torch.nn.functional.cosine_similarity(c16.flatten(), c32.flatten(), 0))
        errs.append(e)
        print(f"|delta| = {a:g}: logit change fp32 |.| {float(c32.norm()):.3e}, bf16 {float(c16.norm()):.3e}; rel-L2 error {e:.3f}, cosine {cos:.4f}")
        assert cos > 0.9 and e < 0.25, (a, e, cos)


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


@pytest.mark.gpu
def test_videoresnet_dense_l12_attack():
    """the dense "L12" attack of the torch learner (attack_type != 'flickering': perturbation [3,T,H,W], model.py:380-384;
    loss = adv + lambda * L12(clamped delta), model.py:169-175,211-214) through FlickerVideoResNet: losses against the oracle,
    the dense adversarial gradient via the link from the stem (like-for-like), and one torch-Adam update of every pixel."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses
    arch, Tn = "r3d_18", 8
    W = vs.synthetic_weights(arch, 42)
    Wd = {k: torch.from_numpy(v).double() for k, v in W.items()}
    x_cl = torch.from_numpy(vs.synthetic_clip(1, Tn, HW, HW, seed=5))
    rng = np.random.default_rng(12)
    delta = torch.from_numpy(rng.uniform(-0.05, 0.05, (3, Tn, HW, HW)).astype(np.float32))
    delta[:, 1, :4] = 0.3                                   # beyond dynamic_max_norm = 0.2
    eng = FlickerVideoResNet(arch, W, batch_size=1, sample_length=Tn, image_size=HW, dtype="f32", l_inf_pert_norm=0.2, attack_type="L12")
    assert eng.pert_model.size == (3, Tn, HW, HW) and eng.pert_model.perturbation.shape == (Tn, HW, HW, 3)
    eng.pert_model.init_perturbation(delta.numpy())
    crit = Losses(lambda_=0.7, margin=0.05, improve_loss=True, logits=True, attack_type="L12")
    with pytest.raises(ValueError):
        eng.step(x_cl.cuda(), torch.zeros(1, dtype=torch.int64).cuda(), Losses(attack_type="flickering"))
    # oracle
    d0 = delta.double().clone().requires_grad_(True)
    xa = am.torch_apply(x_cl.double().permute(0, 4, 1, 2, 3).contiguous(), d0, 0.2)
    logits = vr.videoresnet_logits(xa, Wd, arch)
    label = logits.argmax(-1)
    loss, adv, reg = am.torch_losses(label, logits, torch.softmax(logits, 1), d0.clamp(-0.2, 0.2), 0.5, 0.7, 0.05, True, True, "L12")
    (gtot,) = torch.autograd.grad(loss, d0, retain_graph=True)
    r = eng.step(x_cl.cuda(), label.cuda(), crit, lr=1e-3)
    assert float(r["adv_loss"]) == pytest.approx(adv.item(), rel=1e-3, abs=1e-6)
    assert float(r["reg_loss"]) == pytest.approx(reg.item(), rel=1e-4)
    assert float(r["loss"]) == pytest.approx(loss.item(), rel=1e-3)
    # dense d(adv)/d(delta): the stem link on the HIP path's own stem gradient (like-for-like, no ReLU decisions in between)
    hip_act = lambda n: torch.from_numpy(eng.net.activation(n)).permute(0, 4, 1, 2, 3).contiguous().double()
    pre = vr.stem(xa, Wd, arch, False)
    (g_ref,) = torch.autograd.grad(pre, d0, grad_outputs=hip_act("grad:stem"))
    g_hip = eng._gdense.cpu().permute(3, 0, 1, 2).double()
    print(f"dense d(adv)/d(delta): max-rel {rel_err(g_hip, g_ref):.2e}")
    assert rel_err(g_hip, g_ref) < 1e-3 and float(g_hip[:, 1, :4].abs().max()) == 0
    # the update: first torch-Adam step = lr * sign of the total gradient wherever it is not tiny
    moved = eng.pert_model.get_perturbation()[1].cpu().double() - delta.double()
    big = gtot.abs() > 0.05 * gtot.abs().max()
    assert (torch.sign(moved[big]) == -torch.sign(gtot[big])).double().mean() > 0.995
    assert float(moved.abs().max()) == pytest.approx(1e-3, rel=1e-2)
    # an evaluation pass reports the same regulariser for the same delta
    r_eval = eng.step(x_cl.cuda(), label.cuda(), crit, update=False)
    r_upd = eng.step(x_cl.cuda(), label.cuda(), crit, lr=0.0)
    assert float(r_eval["reg_loss"]) == pytest.approx(float(r_upd["reg_loss"]), rel=1e-4)

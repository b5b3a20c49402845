"""End-to-end parity of the I3D attack iteration (GPU, C ABI) against the CPU oracle
(oracle/i3d_ref.py + oracle/attack_math.py) on the same seeded synthetic weights / clip.

north-star bar: logits, adversarial loss and learned delta within 1e-3 relative (fp32 mode).
bf16 mode (performance mode, bf16 storage + bf16 MFMA, fp32 accumulate) is checked at the looser,
stated tolerances below and its measured error is printed."""
import numpy as np
import pytest
import torch

from oracle import attack_math as am
from oracle import i3d_ref

pytestmark = pytest.mark.gpu

T = 16  # smallest clip the topology admits (T/2 -> pool4a /2 -> pool5a /2 -> 2-frame avg-pool)


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    W = i3d_spec.synthetic_i3d_weights(42)
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=1234))
    rng = np.random.default_rng(3)
    delta = torch.from_numpy(rng.uniform(-0.08, 0.08, (T, 1, 1, 3)).astype(np.float32))
    delta[3] = 0.45      # beyond the +-0.4 clip: gradient must vanish there
    return W, Wt, xu, delta


def oracle_forward(Wt, xu, delta, endpoints=False):
    x = xu.float() / 128 - 1
    d = delta.clone().requires_grad_(True)
    xa = am.tf_apply(x, d)
    out = i3d_ref.i3d_logits(xa, Wt, return_endpoints=endpoints)
    return (out, d) if not endpoints else (out[0], out[1], d)


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("dtype,tol_logits,tol_grad", [("f32", 1e-3, 1e-3), ("bf16", 5e-2, 1.5e-1)])
def test_forward_backward_vs_oracle(setup, dtype, tol_logits, tol_grad):
    from flickering_adversarial_video_amd import ops
    from flickering_adversarial_video_amd._lib import FLK_NET_I3D
    W, Wt, xu, delta = setup
    logits_ref, ep, d = oracle_forward(Wt, xu, delta, endpoints=True)
    label = torch.tensor([int(logits_ref.argmax())])
    loss_ref, _, _ = am.tf_improve_adversarial_loss(logits_ref, label, 0.05, False, False)
    (g_ref,) = torch.autograd.grad(loss_ref, d)

    net = ops.Net(FLK_NET_I3D, dtype, 1, T, 224, 224, W)
    args = ops.make_apply_args(xu.cuda(), delta.reshape(T, 3).contiguous().cuda())
    xs = ops.perturb_apply_s2d(args, dtype)
    logits = net.forward(xs)
    # endpoints first: localises a failure
    for name in ("Conv3d_1a_7x7", "MaxPool3d_2a_3x3", "Conv3d_2c_3x3", "Mixed_3b", "Mixed_3c", "Mixed_4b", "Mixed_4f", "Mixed_5c"):
        act = torch.from_numpy(net.activation(name))
        ref = ep[name].detach().permute(0, 2, 3, 4, 1)
        e = rel_err(act, ref)
        print(f"[{dtype}] {name}: max rel err {e:.3e}")
        assert e < (1e-3 if dtype == "f32" else 6e-2), name
    e = rel_err(logits.cpu(), logits_ref.detach())
    print(f"[{dtype}] logits: max rel err {e:.3e}")
    assert e < tol_logits
    sm, dl, pc = ops.softmax_adv_loss(logits, label.cuda(), dialect="tf", improve_loss=True, margin=0.05)
    assert pc[0, 0].item() == pytest.approx(loss_ref.item(), rel=tol_logits * 5, abs=1e-5)
    gx = net.backward(dl)
    g = ops.perturb_grad_reduce(args, gx.view(1, T // 2, 112, 112, 32)).cpu().reshape(g_ref.shape)
    e = rel_err(g, g_ref)
    print(f"[{dtype}] d(adv)/d(delta): max rel err {e:.3e}; |g|max {g_ref.abs().max():.3e}")
    assert e < tol_grad
    assert g[3].abs().max() == 0          # clipped delta entries get no gradient


def test_attack_trajectory_vs_oracle(setup):
    """4 iterations of the single-video attack (i3d_adversarial_main_single_video_npy.py:211-217) from delta=0:
    logits, adversarial loss and the learned delta against the oracle loop, fp32 mode, 1e-3 relative."""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu, _ = setup
    x = xu.float() / 128 - 1
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32")
    clean = eng(xu.cuda(), adv_flag=0).cpu()
    logits0 = i3d_ref.i3d_logits(x, Wt)
    torch.testing.assert_close(clean, torch.softmax(logits0, -1), rtol=1e-3, atol=1e-6)
    label = logits0.argmax(-1)
    d = torch.zeros(T, 1, 1, 3)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    b0, b1, b2, b3 = 1.0, 0.5, 0.5, 0.5
    for it in range(1, 5):
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x, dv), Wt)
        adv, to_min, to_max = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
        total, reg = am.tf_total_loss(adv, dv, b0, b1, b2, b3)
        (g,) = torch.autograd.grad(total, dv)
        res = eng.step(xu.cuda(), label.cuda(), lr=1e-3, beta0=b0, beta1=b1, beta2=b2, beta3=b3, margin=0.05).host()
        assert res["adv_loss"] == pytest.approx(adv.item(), rel=1e-3, abs=1e-6)
        assert res["total_loss"] == pytest.approx(total.item(), rel=1e-3, abs=1e-6)
        assert res["prob_to_min"] == pytest.approx(to_min.item(), rel=1e-3)
        torch.testing.assert_close(torch.from_numpy(res["softmax"]), torch.softmax(lg.detach(), -1), rtol=1e-3, atol=1e-6)
        d, m, v = am.tf_adam_step(d, g, m, v, it)
        got = eng.perturbation.cpu()
        e = rel_err(got, d)
        print(f"iter {it}: adv {adv.item():.6f} delta rel err {e:.3e}")
        assert e < 1e-3


def test_bf16_step_runs_and_tracks_fp32(setup):
    """bf16 performance mode: same iteration, looser agreement with the fp32 engine (stated: 10% on the loss,
    sign agreement of the first Adam step on >= 90% of the delta entries)."""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu, _ = setup
    label = i3d_ref.i3d_logits(xu.float() / 128 - 1, Wt).argmax(-1).cuda()
    out = {}
    for dt in ("f32", "bf16"):
        eng = FlickerI3D(W, batch_size=1, frames=T, dtype=dt)
        r = eng.step(xu.cuda(), label).host()
        out[dt] = (r["adv_loss"], eng.perturbation.cpu().clone())
        del eng
    assert out["bf16"][0] == pytest.approx(out["f32"][0], rel=0.1, abs=1e-3)
    agree = (torch.sign(out["bf16"][1]) == torch.sign(out["f32"][1])).float().mean().item()
    print(f"bf16 vs f32: adv {out['bf16'][0]:.5f} vs {out['f32'][0]:.5f}; first-step sign agreement {agree:.3f}")
    assert agree >= 0.9

"""End-to-end parity of the I3D attack iteration (GPU, C ABI) against the CPU oracle
(oracle/i3d_ref.py + oracle/attack_math.py) on the same seeded synthetic weights / clip.

north-star bar: logits, adversarial loss and learned delta within 1e-3 relative (fp32 mode).

How the bar is applied.
* Smooth quantities (every forward endpoint, logits, softmax, adversarial loss) are asserted at 1e-3 against the fp32 CPU oracle
  on every fixture (measured ~1e-6).
* The learned delta is asserted at 1e-3 against the fp64 oracle trajectory on the WELL-CONDITIONED fixture
  (oracle/fixtures.py: same topology and kernels, non-negative weights -> no cancellation in the gradient sums; the torch-CPU
  fp32 oracle itself reproduces the fp64 delta to ~3e-6 there): test_attack_trajectory_well_conditioned.
* On the random-sign ("noise") fixture two fp32 implementations of the same forward pass disagree on ~10-30 of the ~3e7 ReLU /
  max-pool decisions of a 16-frame clip, and ONE flipped unit in a 4e5-unit layer moves d(loss)/d(delta) by ~1/sqrt(N) = 1e-3
  (measured: one flip in Mixed_4e between torch-CPU fp32 and fp64 leaves rel-L2 6e-3 in every gradient buffer below it).  The
  backward pass is therefore checked LINK BY LINK (test_forward_backward_vs_oracle): the HIP gradient of each endpoint is pushed
  through the oracle's backward of just the next block and compared with the HIP gradient of the endpoint below -- at 1e-3 of
  the buffer's maximum for all but a counted few elements (the neighbourhoods of flipped decisions).  A wiring / indexing /
  mask bug in any block shows up as an O(1) error of that link; accumulated flips do not.
* bf16 mode (bf16 storage + bf16 MFMA, fp32 accumulate) goes through the same links against a like-for-like oracle (weights,
  activations and gradients rounded to bf16 where the HIP path stores bf16) at the stated bf16 tolerances."""
import numpy as np
import pytest
import torch

from oracle import attack_math as am
from oracle import i3d_ref

pytestmark = pytest.mark.gpu

GRAD_ENDPOINTS = ["Conv3d_1a_7x7", "MaxPool3d_2a_3x3", "Conv3d_2b_1x1", "Conv3d_2c_3x3", "MaxPool3d_3a_3x3", "Mixed_3b",
                  "Mixed_3c", "MaxPool3d_4a_3x3", "Mixed_4b", "Mixed_4c", "Mixed_4d", "Mixed_4e", "Mixed_4f", "MaxPool3d_5a_2x2",
                  "Mixed_5b", "Mixed_5c"]
T = 16  # smallest clip the topology admits (T/2 -> pool4a /2 -> pool5a /2 -> 2-frame avg-pool)
BETAS = (1.0, 0.5, 0.5, 0.5)


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


class _RoundBF16(torch.autograd.Function):
    """bf16 storage of an activation (forward) and of its gradient buffer (backward), as the HIP bf16 mode does"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def oracle_links(W, bf16, dt=torch.float64):
    """[(endpoint name, fn(previous endpoint NCDHW fp64, final_relu) -> this endpoint)] following i3d.py:144-455.  With
    final_relu=False the block's OUTPUT units skip their ReLU: the HIP gradient buffers hold d(loss)/d(pre-ReLU output), so the
    backward of a link starts exactly there.  In bf16 mode weights are bf16-rounded and every stored activation / gradient is
    rounded to bf16 (like-for-like with the HIP bf16 path)."""
    rnd = _RoundBF16.apply if bf16 else (lambda t: t)
    Wd = {k: (torch.from_numpy(v).to(torch.bfloat16).to(dt) if (bf16 and k.endswith("/w") and "Logits" not in k) else torch.from_numpy(v).to(dt))
          for k, v in W.items()}

    def u(x, name, k, s=(1, 1, 1), relu=True):
        y = i3d_ref.unit3d(x, Wd, name, k, s, relu=relu)
        return rnd(y) if relu else y

    def mixed(x, n, fr):
        b0 = u(x, n + "/Branch_0/Conv3d_0a_1x1", (1, 1, 1), relu=fr)
        b1 = u(u(x, n + "/Branch_1/Conv3d_0a_1x1", (1, 1, 1)), n + "/Branch_1/Conv3d_0b_3x3", (3, 3, 3), relu=fr)
        b2 = u(u(x, n + "/Branch_2/Conv3d_0a_1x1", (1, 1, 1)), n + "/Branch_2/" + i3d_ref.b2_3x3_name(n), (3, 3, 3), relu=fr)
        b3 = u(rnd(i3d_ref.maxpool_same(x, (3, 3, 3), (1, 1, 1))), n + "/Branch_3/Conv3d_0b_1x1", (1, 1, 1), relu=fr)
        return torch.cat([b0, b1, b2, b3], 1)

    links = [("Conv3d_1a_7x7", lambda x, fr: u(rnd(x), "Conv3d_1a_7x7", (7, 7, 7), (2, 2, 2), relu=fr)),
             ("MaxPool3d_2a_3x3", lambda x, fr: i3d_ref.maxpool_same(x, (1, 3, 3), (1, 2, 2))),
             ("Conv3d_2b_1x1", lambda x, fr: u(x, "Conv3d_2b_1x1", (1, 1, 1), relu=fr)),
             ("Conv3d_2c_3x3", lambda x, fr: u(x, "Conv3d_2c_3x3", (3, 3, 3), relu=fr)),
             ("MaxPool3d_3a_3x3", lambda x, fr: i3d_ref.maxpool_same(x, (1, 3, 3), (1, 2, 2)))]
    for name, spec in i3d_ref.MIXED:
        links.append((name, (lambda x, fr, sp=spec: i3d_ref.maxpool_same(x, *sp)) if name.startswith("MaxPool") else (lambda x, fr, n=name: mixed(x, n, fr))))

    def head(x, fr):
        x = torch.nn.functional.avg_pool3d(x, (2, 7, 7), (1, 1, 1))
        return i3d_ref.unit3d(x, Wd, "Logits/Conv3d_0c_1x1", (1, 1, 1), bn=False, relu=False, bias=True).squeeze(4).squeeze(3).mean(2)
    links.append(("Logits", head))
    return links


def outlier_frac(got, ref, tol):
    """fraction of elements further than tol * max|ref| from ref"""
    return float(((got.double() - ref.double()).abs() > tol * ref.double().abs().max()).double().mean())


def oracle_pass(Wt, xu, delta, dt):
    """one forward + backward of the oracle in dtype dt; returns logits, endpoints, loss, d(loss)/d(delta), endpoint grads"""
    x = xu.to(dt) / 128 - 1
    d = delta.to(dt).clone().requires_grad_(True)
    logits, ep = i3d_ref.i3d_logits(am.tf_apply(x, d), Wt[dt], return_endpoints=True)
    label = logits.argmax(-1)
    loss, _, _ = am.tf_improve_adversarial_loss(logits, label, 0.05, False, False)
    g, *ge = torch.autograd.grad(loss, [d] + [ep[n] for n in GRAD_ENDPOINTS])
    return dict(logits=logits.detach(), ep={k: v.detach() for k, v in ep.items()}, loss=loss.item(), label=label, g=g,
                ge=dict(zip(GRAD_ENDPOINTS, ge)))


def oracle_trajectory(Wt, xu, dt, steps):
    """single-video attack loop (i3d_adversarial_main_single_video_npy.py:211-217) from delta = 0"""
    x = xu.to(dt) / 128 - 1
    label = i3d_ref.i3d_logits(x, Wt[dt]).argmax(-1)
    d = torch.zeros(T, 1, 1, 3, dtype=dt)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    out = []
    for it in range(1, steps + 1):
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x, dv), Wt[dt])
        adv, to_min, to_max = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
        total, reg = am.tf_total_loss(adv, dv, *BETAS)
        (g,) = torch.autograd.grad(total, dv)
        d, m, v = am.tf_adam_step(d, g, m, v, it)
        out.append(dict(adv=adv.item(), total=total.item(), to_min=to_min.item(), softmax=torch.softmax(lg.detach(), -1), logits=lg.detach(), delta=d.clone()))
    return label, out


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    W = i3d_spec.synthetic_i3d_weights(42)
    Wt = {dt: {k: torch.from_numpy(v).to(dt) for k, v in W.items()} for dt in (torch.float32, torch.float64)}
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=1234))
    rng = np.random.default_rng(3)
    delta = torch.from_numpy(rng.uniform(-0.08, 0.08, (T, 1, 1, 3)).astype(np.float32))
    delta[3] = 0.45      # beyond the +-0.4 clip: gradient must vanish there
    ref = {dt: oracle_pass(Wt, xu, delta, dt) for dt in (torch.float32, torch.float64)}
    return W, Wt, xu, delta, ref


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_backward_vs_oracle(setup, dtype):
    from flickering_adversarial_video_amd import ops
    from flickering_adversarial_video_amd._lib import FLK_NET_I3D
    W, Wt, xu, delta, ref = setup
    r32, r64 = ref[torch.float32], ref[torch.float64]
    f32 = dtype == "f32"
    net = ops.Net(FLK_NET_I3D, dtype, 1, T, 224, 224, W)
    args = ops.make_apply_args(xu.cuda(), delta.reshape(T, 3).contiguous().cuda(), fold_t=ops.I3D_FOLD)
    logits = net.forward(ops.perturb_apply_s2d(args, dtype))
    # ---- forward: every endpoint, logits, loss against the fp32 oracle (smooth quantities) ----
    for name in r32["ep"]:
        e = rel_err(torch.from_numpy(net.activation(name)), r32["ep"][name].permute(0, 2, 3, 4, 1))
        print(f"[{dtype}] {name}: max rel err {e:.3e}")
        assert e < (1e-3 if f32 else 6e-2), name
    e = rel_err(logits.cpu(), r32["logits"])
    print(f"[{dtype}] logits: max rel err {e:.3e}")
    assert e < (1e-3 if f32 else 5e-2)
    sm, dl, pc = ops.softmax_adv_loss(logits, r32["label"].cuda(), dialect="tf", improve_loss=True, margin=0.05)
    assert pc[0, 0].item() == pytest.approx(r32["loss"], rel=1e-3 if f32 else 5e-2, abs=1e-6)
    # ---- backward, link by link.  Gradient buffers hold d(loss)/d(pre-ReLU output) = d(loss)/d(endpoint) masked by endpoint > 0
    # (max-pool endpoints: unmasked).  For every link the oracle block is evaluated on the HIP path's OWN input endpoint and its
    # backward is fed the HIP path's OWN output gradient: what is compared is this block's arithmetic alone. ----
    gx = net.backward(dl)
    bf = not f32
    # tolerances: element error relative to the buffer maximum; allowed fraction of elements beyond it (flipped-decision neighbourhoods)
    FWD_TOL, BWD_TOL, BWD_FRAC, BWD_L2 = (1e-4, 1e-3, 1e-3, 1e-2) if f32 else (1.6e-2, 3e-2, 2e-2, 3e-2)
    hip_act = lambda n: torch.from_numpy(net.activation(n)).permute(0, 4, 1, 2, 3).contiguous().double()
    d0 = delta.double().clone().requires_grad_(True)
    prev_name, prev = "delta", am.tf_apply(xu.double() / 128 - 1, d0).permute(0, 4, 1, 2, 3).contiguous()
    worst = {}
    for name, fn in oracle_links(W, bf):
        with torch.no_grad():
            out = fn(prev, True)              # forward value of the endpoint (with its ReLU)
        pre = fn(prev, False)                 # the same block up to the pre-ReLU output: where the HIP gradient buffer lives
        if name == "Logits":
            got_f, got_g = logits.cpu().double(), dl.cpu().double()
        else:
            got_f, got_g = hip_act(name), hip_act("grad:" + name)
        e_f = rel_err(out.detach(), got_f)
        assert e_f < FWD_TOL, f"forward link {prev_name} -> {name}: {e_f:.3e}"
        if prev_name == "delta":
            (g_ref,) = torch.autograd.grad(pre, d0, grad_outputs=got_g)
            g_hip = ops.perturb_grad_reduce(args, gx.view(1, T // 2, 112, 112, 32)).cpu().reshape(g_ref.shape).double()
            g_link = g_hip
            e_l2, frac = rel_l2(g_hip, g_ref), outlier_frac(g_hip, g_ref, BWD_TOL)
            assert rel_err(g_hip, g_ref) < (1e-3 if f32 else 2e-2), f"d(loss)/d(delta) link: {rel_err(g_hip, g_ref):.3e}"
        else:
            (g_ref,) = torch.autograd.grad(pre, prev, grad_outputs=got_g)
            if not prev_name.startswith("MaxPool"):
                g_ref = torch.where(prev > 0, g_ref, torch.zeros_like(g_ref))
            g_hip = hip_act("grad:" + prev_name)
            e_l2, frac = rel_l2(g_hip, g_ref), outlier_frac(g_hip, g_ref, BWD_TOL)
        worst[name] = (e_f, e_l2, frac)
        print(f"[{dtype}] link {prev_name:>17s} -> {name:<17s} fwd max-rel {e_f:.2e} | bwd rel-L2 {e_l2:.2e}, elements beyond {BWD_TOL:g} of max: {frac:.2e}")
        assert frac <= BWD_FRAC and e_l2 < BWD_L2, f"backward link {name} -> {prev_name}: rel-L2 {e_l2:.3e}, outliers {frac:.3e}"
        # next link starts from the HIP path's own endpoint (like-for-like; an upstream flip cannot leak into the next comparison)
        prev_name = name
        if name != "Logits":
            prev = hip_act(name).requires_grad_(True)
    # ---- end to end against the fp64 oracle (information + sanity): direction of d(adv)/d(delta), clipped entries ----
    g = g_link.reshape(r64["g"].shape)
    e_hip, e_cpu = rel_err(g, r64["g"]), rel_err(r32["g"], r64["g"])
    cos = float(torch.nn.functional.cosine_similarity(g.double().flatten(), r64["g"].flatten(), 0))
    print(f"[{dtype}] d(adv)/d(delta) end to end vs fp64 oracle: HIP max-rel {e_hip:.3e} (fp32 CPU oracle {e_cpu:.3e}); cosine {cos:.6f}")
    assert cos > (0.999 if f32 else 0.92)      # bf16: measured 0.946 (the per-link bars above are the parity statement; this guards the sum)
    assert g[3].abs().max() == 0             # clipped delta entries get no gradient (kinetics_i3d_utils.py:104)


def test_attack_trajectory_vs_oracle(setup):
    """4 attack iterations from delta=0, fp32 mode: softmax / adversarial loss at 1e-3 against the oracle every
    step; learned delta against the fp64 oracle trajectory, no worse than the fp32 CPU oracle's own deviation."""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu, _, _ = setup
    steps = 4
    label, t32 = oracle_trajectory(Wt, xu, torch.float32, steps)
    _, t64 = oracle_trajectory(Wt, xu, torch.float64, steps)
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32")
    clean = eng(xu.cuda(), adv_flag=0).cpu()
    torch.testing.assert_close(clean, torch.softmax(i3d_ref.i3d_logits(xu.float() / 128 - 1, Wt[torch.float32]), -1), rtol=1e-3, atol=1e-6)
    for it in range(steps):
        res = eng.step(xu.cuda(), label.cuda(), lr=1e-3, beta0=BETAS[0], beta1=BETAS[1], beta2=BETAS[2], beta3=BETAS[3], margin=0.05).host()
        got = eng.perturbation.cpu()
        e_hip, e_cpu = rel_err(got, t64[it]["delta"]), rel_err(t32[it]["delta"], t64[it]["delta"])
        print(f"iter {it + 1}: adv {res['adv_loss']:.6f} (oracle {t64[it]['adv']:.6f}); delta max-rel vs fp64: HIP {e_hip:.3e}, fp32 CPU oracle {e_cpu:.3e}")
        # the loss of iteration k depends on delta_{k-1}: smooth in delta, so it stays within 1e-3 of the fp64 trajectory
        assert res["adv_loss"] == pytest.approx(t64[it]["adv"], rel=1e-3, abs=1e-6)
        assert res["total_loss"] == pytest.approx(t64[it]["total"], rel=1e-3, abs=1e-6)
        assert res["prob_to_min"] == pytest.approx(t64[it]["to_min"], rel=1e-3)
        torch.testing.assert_close(torch.from_numpy(res["softmax"]).double(), t64[it]["softmax"], rtol=2e-3, atol=1e-6)
        assert e_hip < 3 * e_cpu + 5e-3


def _coherent_fixture():
    from flickering_adversarial_video_amd import i3d_spec
    from oracle import fixtures
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=1234))
    W = fixtures.coherent_i3d_weights(xu, seed=5, label=233)
    Wt = {dt: {k: torch.from_numpy(v).to(dt) for k, v in W.items()} for dt in (torch.float32, torch.float64)}
    return W, Wt, xu


def test_attack_trajectory_well_conditioned():
    """THE north-star assertion: logits (softmax), adversarial loss and the LEARNED DELTA of the fp32 mode within 1e-3 of the
    reference maths (fp64 oracle) over 6 iterations of the single-video loop (i3d_adversarial_main_single_video_npy.py:211-217) on the
    well-conditioned fixture (oracle/fixtures.py).  The fixture is first shown to be well-conditioned: the torch-CPU fp32 oracle
    reproduces the fp64 trajectory to < 1e-4."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu = _coherent_fixture()
    steps = 6
    label, t32 = oracle_trajectory(Wt, xu, torch.float32, steps)
    label64, t64 = oracle_trajectory(Wt, xu, torch.float64, steps)
    assert int(label) == int(label64) == 233
    for it in range(steps):
        assert rel_err(t32[it]["delta"], t64[it]["delta"]) < 1e-4, "fixture is not well-conditioned"
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32")
    clean = eng.logits(xu.cuda(), adv_flag=0.0).cpu()
    e = rel_err(clean, i3d_ref.i3d_logits(xu.double() / 128 - 1, Wt[torch.float64]))
    print(f"clean logits max-rel vs fp64 oracle {e:.2e}")
    assert e < 1e-3
    fooled_at = {"hip": None, "oracle": None}
    for it in range(steps):
        res = eng.step(xu.cuda(), label.cuda(), lr=1e-3, beta0=BETAS[0], beta1=BETAS[1], beta2=BETAS[2], beta3=BETAS[3], margin=0.05).host()
        e_d = rel_err(eng.perturbation.cpu(), t64[it]["delta"])
        e_l = rel_err(eng._logits.cpu(), t64[it]["logits"])
        print(f"iter {it + 1}: adv {res['adv_loss']:.7f} (fp64 oracle {t64[it]['adv']:.7f}); delta max-rel {e_d:.2e}; logits max-rel {e_l:.2e} "
              f"(torch-CPU fp32 oracle delta {rel_err(t32[it]['delta'], t64[it]['delta']):.2e})")
        assert e_d < 1e-3, "learned delta"
        assert res["adv_loss"] == pytest.approx(t64[it]["adv"], rel=1e-3, abs=1e-7)
        assert res["total_loss"] == pytest.approx(t64[it]["total"], rel=1e-3, abs=1e-7)
        assert res["prob_to_min"] == pytest.approx(t64[it]["to_min"], rel=1e-3)
        torch.testing.assert_close(torch.from_numpy(res["softmax"]).double(), t64[it]["softmax"], rtol=1e-3, atol=1e-7)
        assert e_l < 1e-3
        for k, sm in (("hip", torch.from_numpy(res["softmax"])), ("oracle", t64[it]["softmax"])):
            if fooled_at[k] is None and int(sm.argmax()) != int(label):
                fooled_at[k] = it
    assert fooled_at["hip"] == fooled_at["oracle"]          # same iteration-to-fool (None: not fooled within the 6 steps)


def test_evaluate_fooling_rate_vs_oracle():
    """FlickerI3D.evaluate = kinetics_i3d.evaluate (kinetics_i3d_utils.py:217-250): the integer counts (fooled AND clean-correct,
    clean-correct) over an iterator of batches must equal the reference formula applied to the ORACLE's predictions, for the
    untargeted, targeted and exclude_misclassify=False variants.  Fixture: the random-sign weights with the logits bias centred
    on the four clips (otherwise the random network predicts one class whatever the input), clips of different contrast /
    brightness, one clip deliberately mislabelled; the oracle's top-2 logit gaps are asserted to be wide (> 1e-3 of the largest
    logit; the fp32 path reproduces logits to ~1e-6), so no argmax can hinge on rounding."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W = i3d_spec.synthetic_i3d_weights(42)
    bkey = i3d_ref.PREFIX + "Logits/Conv3d_0c_1x1/conv_3d/b"
    noise = i3d_spec.synthetic_clip_u8(4, T, seed=314).astype(np.float64)
    clips = torch.from_numpy(np.stack([np.clip(128 + a * (noise[i] - 128) + b, 0, 255)
                                       for i, (a, b) in enumerate([(1.0, 0), (0.3, -60), (0.3, 60), (0.6, -30)])]).astype(np.uint8))
    x = clips.double() / 128 - 1
    W[bkey] = np.zeros(400, np.float32)
    W[bkey] = (-i3d_ref.i3d_logits(x, {k: torch.from_numpy(v).double() for k, v in W.items()}).mean(0)).numpy().astype(np.float32)
    W64 = {k: torch.from_numpy(v).double() for k, v in W.items()}
    delta = torch.full((T, 1, 1, 3), -0.2, dtype=torch.float32)
    lg_clean = i3d_ref.i3d_logits(x, W64)
    lg_adv = i3d_ref.i3d_logits(am.tf_apply(x, delta.double()), W64)
    for lg in (lg_clean, lg_adv):
        top = lg.topk(2).values
        assert float(((top[:, 0] - top[:, 1]) / lg.abs().max(1).values).min()) > 1e-3, "fixture: an argmax hinges on rounding"
    gt = lg_clean.argmax(1).clone()
    fooled = lg_adv.argmax(1) != gt
    assert 0 < int(fooled.sum()) < 4, "fixture: want a mixed outcome"
    keep = int((~fooled).nonzero()[0])
    gt[keep] = (gt[keep] + 1) % 400                            # one not-fooled clip is mislabelled: excluded from the valid set
    target = int(lg_adv.argmax(1)[int(fooled.nonzero()[-1])])
    print("oracle clean argmax", lg_clean.argmax(1).tolist(), "adv argmax", lg_adv.argmax(1).tolist(), "labels", gt.tolist(), "target", target)
    eng = FlickerI3D(W, batch_size=2, frames=T, dtype="f32")
    eng.reset_perturbation(delta.numpy())
    batches = lambda: ((clips[i:i + 2].cuda(), gt[i:i + 2].cuda()) for i in (0, 2))
    for kw in (dict(), dict(targeted_attack=True, target_class_id=target), dict(exclude_misclassify=False)):
        miss, valid = am.fooling_counts(lg_adv, lg_clean, gt, targeted=kw.get("targeted_attack", False), target=kw.get("target_class_id"),
                                        exclude_misclassify=kw.get("exclude_misclassify", True))
        rate, total = eng.evaluate(batches(), **kw)
        print(kw, "-> oracle counts", (miss, valid), "engine", (rate, total))
        assert total == valid and rate == miss / valid
    assert am.fooling_counts(lg_adv, lg_clean, gt)[1] == 3


def test_step_rejects_bad_labels():
    """a CPU label tensor would hand the loss kernel a host pointer (GPU memory fault), a class id >= 400 an out-of-range index:
    both must be refused on the host; the kernel itself turns an out-of-range id into NaN outputs instead of reading out of bounds"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import _lib, i3d_spec, ops
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=T, dtype="bf16")
    x = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=1)).cuda()
    for bad in (torch.tensor([5]), torch.tensor([5], dtype=torch.int32).cuda(), torch.tensor([5, 6]).cuda(), torch.tensor([400]).cuda(),
                torch.tensor([-1]).cuda()):
        with pytest.raises(ValueError):
            eng.step(x, bad)
    assert eng.adam_t == 0
    eng.step(x, torch.tensor([399]).cuda())
    # the kernel's own guard (bypassing the host check)
    lg = torch.randn(2, 400, device="cuda")
    lab = torch.tensor([3, 400], device="cuda")
    sm, dl, pc = (torch.empty_like(lg), torch.empty_like(lg), torch.empty(2, 4, device="cuda"))
    a = _lib.LossArgs()
    a.B, a.C, a.improve_loss, a.margin, a.mean_scale = 2, 400, 1, 0.05, 1.0
    import ctypes
    _lib.check(_lib.load().flk_softmax_adv_loss(ctypes.byref(a), _lib.ptr(lg), _lib.ptr(lab), _lib.ptr(sm), _lib.ptr(dl), _lib.ptr(pc), _lib.stream_ptr()))
    assert torch.isfinite(pc[0, 0]) and torch.isfinite(dl[0]).all() and torch.isnan(pc[1, 0]) and torch.isnan(dl[1]).all()


def test_bf16_trajectory_well_conditioned():
    """bf16 mode on the well-conditioned fixture: the perturbation reaches the stem in fp32 (centred clip + position-class bias,
    Net.forward_flicker), so a delta of 1e-3 -- below half a bf16 ulp of most pixels, which a rounded x + delta would swallow -- moves
    the logits as the reference maths says.  Stated tolerances: learned delta 2e-2 (max-rel), logits 5e-2, and the CHANGE of the
    adversarial loss from the first iteration within 15% of the fp64 oracle's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu = _coherent_fixture()
    steps = 6
    label, t64 = oracle_trajectory(Wt, xu, torch.float64, steps)
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="bf16")
    assert eng.exact_delta_forward and eng.fused_delta_grad
    adv = []
    for it in range(steps):
        res = eng.step(xu.cuda(), label.cuda(), lr=1e-3, beta0=BETAS[0], beta1=BETAS[1], beta2=BETAS[2], beta3=BETAS[3], margin=0.05).host()
        e_d = rel_err(eng.perturbation.cpu(), t64[it]["delta"])
        e_l = rel_err(eng._logits.cpu(), t64[it]["logits"])
        adv.append(res["adv_loss"])
        print(f"iter {it + 1}: adv {res['adv_loss']:.6f} (fp64 oracle {t64[it]['adv']:.6f}); delta max-rel {e_d:.2e}; logits max-rel {e_l:.2e}")
        assert e_d < 2e-2 and e_l < 5e-2
    d_hip, d_ref = adv[-1] - adv[0], t64[-1]["adv"] - t64[0]["adv"]
    print(f"adversarial-loss change over {steps} iterations: {d_hip:.3e} (fp64 oracle {d_ref:.3e})")
    assert d_hip == pytest.approx(d_ref, rel=0.15)


def test_bf16_step_runs_and_tracks_fp32(setup):
    """bf16 performance mode: same iteration, looser agreement with the fp32 engine (stated: 5% on the loss,
    first Adam step -- a pure sign step -- agreeing on >= 82% of the delta entries, gradient cosine > 0.91; measured 85.4 % / 0.942:
    on the random-sign weights the gradient is a cancelling sum, so two roundings of it disagree on the sign of its small entries)."""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu, _, ref = setup
    label = i3d_ref.i3d_logits(xu.float() / 128 - 1, Wt[torch.float32]).argmax(-1).cuda()
    out = {}
    for dt in ("f32", "bf16"):
        eng = FlickerI3D(W, batch_size=1, frames=T, dtype=dt)
        r = eng.step(xu.cuda(), label).host()
        out[dt] = (r["adv_loss"], eng.perturbation.cpu().clone(), eng.delta_gradient().cpu().clone())
        del eng
    assert out["bf16"][0] == pytest.approx(out["f32"][0], rel=0.05, abs=1e-3)
    agree = (torch.sign(out["bf16"][1]) == torch.sign(out["f32"][1])).float().mean().item()
    cos = float(torch.nn.functional.cosine_similarity(out["bf16"][2].flatten(), out["f32"][2].flatten(), 0))
    print(f"bf16 vs f32: adv {out['bf16'][0]:.5f} vs {out['f32'][0]:.5f}; first-step sign agreement {agree:.3f}; grad cosine {cos:.4f}")
    assert agree >= 0.82 and cos > 0.91


def test_dense_delta_step(setup):
    """dense-delta baseline (kinetics_i3d_L12, kinetics_i3d_utils.py:308-521): one step from delta = 1e-8; the adversarial loss
    matches the oracle, L12 = T * 1e-8 + 1e-12, and the first TF-Adam step moves every pixel by lr against its gradient sign"""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu, _, _ = setup
    x = xu.float() / 128 - 1
    d = torch.full((T, 224, 224, 3), 1e-8, requires_grad=True)
    lg = i3d_ref.i3d_logits(am.tf_apply(x, d, clip_delta=False), Wt[torch.float32])
    label = lg.argmax(-1)
    adv, _, _ = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
    (g,) = torch.autograd.grad(adv + 0.5 * am.tf_l12(d), d)
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32", dense_delta=True)
    res = eng.step(xu.cuda(), label.cuda(), lr=1e-3, beta1=0.5).host()
    assert res["adv_loss"] == pytest.approx(adv.item(), rel=1e-3, abs=1e-6)
    assert res["L12"] == pytest.approx(T * 1e-8 + 1e-12, rel=1e-3)
    moved = eng.perturbation.cpu() - 1e-8
    big = g.abs() > 0.05 * g.abs().max()
    assert (torch.sign(moved[big]) == -torch.sign(g[big])).float().mean() > 0.995
    assert float(moved.abs().max()) == pytest.approx(1e-3, rel=1e-2)


def test_dense_delta_trajectory_well_conditioned():
    """the dense-delta baseline (kinetics_i3d_L12, kinetics_i3d_utils.py:308-521; loss = adv + beta0 * beta1 * L12,
    i3d_adversarial_main_universal.py:129-133) at the north-star bar on the well-conditioned fixture: the dense gradient
    d(adv)/d(delta) [T,224,224,3] of the first iteration and the LEARNED dense delta, the adversarial loss and L12 of 3 iterations
    within 1e-3 of the fp64 oracle (autograd through the oracle network, TF-Adam)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu = _coherent_fixture()
    W64 = Wt[torch.float64]
    x64 = xu.double() / 128 - 1
    label = i3d_ref.i3d_logits(x64, W64).argmax(-1)
    assert int(label) == 233
    B0, B1, LR, steps = 1.0, 0.5, 1e-3, 3
    d = torch.full((T, 224, 224, 3), 1e-8, dtype=torch.float64)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32", dense_delta=True)
    for it in range(1, steps + 1):
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x64, dv, clip_delta=False), W64)
        adv, _, _ = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
        l12 = am.tf_l12(dv)
        (g_adv,) = torch.autograd.grad(adv, dv, retain_graph=True)
        (g,) = torch.autograd.grad(adv + B0 * B1 * l12, dv)
        d, m, v = am.tf_adam_step(d, g, m, v, it, lr=LR)
        res = eng.step(xu.cuda(), label.cuda(), lr=LR, beta0=B0, beta1=B1, margin=0.05).host()
        e_g = rel_err(eng._gdense.cpu(), g_adv)
        e_d = rel_err(eng.perturbation.cpu(), d)
        print(f"iter {it}: adv {res['adv_loss']:.7f} (fp64 oracle {adv.item():.7f}); L12 {res['L12']:.6e} ({l12.item():.6e}); "
              f"dense d(adv)/d(delta) max-rel {e_g:.2e}; dense delta max-rel {e_d:.2e}")
        assert res["adv_loss"] == pytest.approx(adv.item(), rel=1e-3, abs=1e-7)
        assert res["L12"] == pytest.approx(l12.item(), rel=1e-3)
        assert e_g < 1e-3 and e_d < 1e-3


VARIANTS = {
    # name: (oracle loss, engine step kwargs, labels are the target class)
    "targeted_prob": (lambda lg, y: am.tf_improve_adversarial_loss(lg, y, 0.05, True, False)[0], dict(targeted=True), True),
    "use_logits": (lambda lg, y: am.tf_improve_adversarial_loss(lg, y, 0.05, False, True)[0], dict(use_logits=True), False),
    "targeted_logits": (lambda lg, y: am.tf_improve_adversarial_loss(lg, y, 0.05, True, True)[0], dict(targeted=True, use_logits=True), True),
    "ce": (lambda lg, y: am.tf_ce_adversarial_loss(lg, y, False)[0], dict(improve_loss=False), False),
    "ce_targeted": (lambda lg, y: am.tf_ce_adversarial_loss(lg, y, True)[0], dict(improve_loss=False, targeted=True), True),
    "cyclic": (lambda lg, y: am.tf_improve_adversarial_loss(lg, y, 0.05, False, False)[0], dict(cyclic=True, cyclic_pert=True), False),
}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_attack_variants_trajectory_well_conditioned(variant):
    """the attack variants of SURVEY N4 through the COMPLETE iteration (not only the loss head): targeted attacks, logits-mode margin,
    cross-entropy losses (kinetics_i3d_utils.py:253-307) and the cyclic clip / perturbation rolls (:100-142) -- adversarial loss,
    logits and learned delta of 2 iterations within 1e-3 of the fp64 oracle on the well-conditioned fixture."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    loss_fn, kw, targeted = VARIANTS[variant]
    W, Wt, xu = _coherent_fixture()
    W64 = Wt[torch.float64]
    x64 = xu.double() / 128 - 1
    y = torch.tensor([100]) if targeted else i3d_ref.i3d_logits(x64, W64).argmax(-1)
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32")
    shifts = iter([3, 11, 7, 2, 5, 9])                      # (clip shift, perturbation shift) per iteration when the rolls are on
    if "cyclic" in kw:
        eng._rng = type("FixedShifts", (), {"integers": lambda self, lo, hi: next(shifts)})()
        oracle_shifts = iter([3, 11, 7, 2, 5, 9])
    d = torch.zeros(T, 1, 1, 3, dtype=torch.float64)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    for it in range(1, 3):
        dv = d.clone().requires_grad_(True)
        if "cyclic" in kw:
            sx, sp = next(oracle_shifts), next(oracle_shifts)
            xa = am.tf_apply(x64, dv, shift_x=sx, cyclic_flag=1.0, shift_p=sp, cyclic_pert_flag=1.0)
        else:
            xa = am.tf_apply(x64, dv)
        lg = i3d_ref.i3d_logits(xa, W64)
        adv = loss_fn(lg, y)
        total, _ = am.tf_total_loss(adv, dv, *BETAS)
        (g,) = torch.autograd.grad(total, dv)
        d, m, v = am.tf_adam_step(d, g, m, v, it)
        res = eng.step(xu.cuda(), y.cuda(), lr=1e-3, beta0=BETAS[0], beta1=BETAS[1], beta2=BETAS[2], beta3=BETAS[3], margin=0.05, **kw).host()
        e_d, e_l = rel_err(eng.perturbation.cpu(), d), rel_err(eng._logits.cpu(), lg.detach())
        print(f"[{variant}] iter {it}: adv {res['adv_loss']:.7f} (fp64 oracle {adv.item():.7f}); delta max-rel {e_d:.2e}; logits max-rel {e_l:.2e}")
        assert res["adv_loss"] == pytest.approx(adv.item(), rel=1e-3, abs=1e-7)
        assert e_d < 1e-3 and e_l < 1e-3
        assert float(g.abs().max()) > 0


def test_universal_batch_trajectory_well_conditioned():
    """the universal attack's batch semantics (one shared delta, margin loss SUMMED over the clips of the batch,
    kinetics_i3d_utils.py:285; i3d_adversarial_main_universal.py:129-133) against the fp64 oracle: 2 different clips, 2 iterations,
    learned delta / logits / loss at 1e-3 on the well-conditioned fixture."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu = _coherent_fixture()
    W64 = Wt[torch.float64]
    xb = torch.cat([xu, torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=99))], 0)
    x64 = xb.double() / 128 - 1
    y = i3d_ref.i3d_logits(x64, W64).argmax(-1)
    eng = FlickerI3D(W, batch_size=2, frames=T, dtype="f32")
    d = torch.zeros(T, 1, 1, 3, dtype=torch.float64)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    for it in range(1, 3):
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x64, dv), W64)
        adv, _, _ = am.tf_improve_adversarial_loss(lg, y, 0.05, False, False)
        total, _ = am.tf_total_loss(adv, dv, *BETAS)
        (g,) = torch.autograd.grad(total, dv)
        d, m, v = am.tf_adam_step(d, g, m, v, it)
        res = eng.step(xb.cuda(), y.cuda(), lr=1e-3, beta0=BETAS[0], beta1=BETAS[1], beta2=BETAS[2], beta3=BETAS[3], margin=0.05).host()
        e_d, e_l = rel_err(eng.perturbation.cpu(), d), rel_err(eng._logits.cpu(), lg.detach())
        print(f"iter {it}: adv (sum over 2 clips) {res['adv_loss']:.7f} (fp64 oracle {adv.item():.7f}); delta max-rel {e_d:.2e}; logits max-rel {e_l:.2e}")
        assert res["adv_loss"] == pytest.approx(adv.item(), rel=1e-3, abs=1e-7)
        assert e_d < 1e-3 and e_l < 1e-3


def test_fp32_iteration_is_bitwise_reproducible(setup):
    """the parity mode is reproducible to the bit: no float atomics on its path (gather-form pool backward, fixed-order reductions) --
    two engines, three iterations, identical delta / gradient / loss bits (the bf16 mode's test is in tests/test_fullsize_gpu.py)"""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W, Wt, xu, _, _ = setup
    outs = []
    for _ in range(2):
        eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32")
        labels = eng.logits(xu.cuda(), adv_flag=0.0).argmax(-1).clone()
        for _ in range(3):
            r = eng.step(xu.cuda(), labels)
        outs.append((eng.perturbation.clone(), eng.delta_gradient().clone(), r["adv_loss"].clone()))
        del eng
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


def test_inference_engine_cyclic_flags():
    """kinetics_i3d_inference (kinetics_i3d_utils.py:574-647): rolls of the clip / the perturbation are tf.roll by the drawn
    shift, the perturbation is NOT clipped to 0.4, adv_flag=0 ignores it."""
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3DInference
    Tn = 16
    W = i3d_spec.synthetic_i3d_weights(42)
    eng = FlickerI3DInference(W, batch_size=1, frames=Tn, dtype="f32", seed=3)
    x = (torch.from_numpy(i3d_spec.synthetic_clip_u8(1, Tn, seed=11)).float() / 128 - 1).cuda()
    rng = np.random.default_rng(0)
    delta = rng.uniform(-0.6, 0.6, (Tn, 1, 1, 3)).astype(np.float32)          # beyond 0.4 on purpose
    eng.set_perturbation(delta)
    clean = eng(x, adv_flag=0).clone()
    assert torch.allclose(clean, eng(x, adv_flag=0, cyclic_eps_flag=1))      # delta is ignored at adv_flag = 0
    adv = eng(x, adv_flag=1).clone()
    assert not torch.allclose(adv, clean, atol=1e-4)
    # reference arithmetic on the host for the plain case: clip(x + delta, -1, 1), delta unclipped
    ref_in = torch.clamp(x + torch.from_numpy(delta).cuda().view(1, Tn, 1, 1, 3), -1, 1)
    eng.set_perturbation(np.zeros_like(delta))
    assert torch.allclose(eng(ref_in, adv_flag=1), adv, atol=2e-6)
    # cyclic perturbation: equals the plain path with the perturbation rolled by the drawn shift
    eng.set_perturbation(delta)
    p_cyc = eng(x, adv_flag=1, cyclic_eps_flag=1).clone()
    eng.set_perturbation(np.roll(delta, eng.last_shift_p, axis=0))
    assert torch.allclose(eng(x, adv_flag=1), p_cyc, atol=2e-6)
    # cyclic clip: equals the plain path on the clip rolled by the drawn shift
    eng.set_perturbation(delta)
    p_cyc = eng(x, adv_flag=1, cyclic_input_flag=1).clone()
    assert eng.last_shift_p == 0
    assert torch.allclose(eng(torch.roll(x, eng.last_shift_x, dims=1), adv_flag=1), p_cyc, atol=2e-6)


def test_autotune_changes_speed_only():
    """flk_net_autotune picks launch layouts per convolution; results must stay bitwise identical (same K order per output)."""
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W = i3d_spec.synthetic_i3d_weights(42)
    x = torch.from_numpy(i3d_spec.synthetic_clip_u8(2, T, seed=4)).cuda()
    eng = FlickerI3D(W, batch_size=2, frames=T, dtype="bf16")
    labels = eng.logits(x, adv_flag=0.0).argmax(-1).clone()
    l0 = eng.logits(x, adv_flag=0.0).clone()
    r0 = eng.step(x, labels, update=False)
    g0 = eng.delta_gradient().clone()
    eng.autotune(x)
    torch.testing.assert_close(eng.logits(x, adv_flag=0.0), l0, rtol=0, atol=0)
    eng.step(x, labels, update=False)
    torch.testing.assert_close(eng.delta_gradient(), g0, rtol=0, atol=0)


@pytest.mark.parametrize("frames", [90, 18])
def test_reference_default_clip_length(frames):
    """The reference runs on 90-frame clips (_SAMPLE_VIDEO_FRAMES, kinetics_i3d_utils.py): T/2 = 45, 23, 12 are odd, so
    the temporal SAME paddings have a pad-before and the strided pools take their fallback kernels.  Logits, loss and
    d(loss)/d(delta) of the fp32 mode against the fp32 CPU oracle (18 frames: the same odd-size paths, small)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    Tn = frames
    W = i3d_spec.synthetic_i3d_weights(42)
    Wt = {torch.float32: {k: torch.from_numpy(v) for k, v in W.items()}}
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, Tn, seed=90))
    delta = torch.from_numpy(np.random.default_rng(9).uniform(-0.05, 0.05, (Tn, 1, 1, 3)).astype(np.float32))
    x = xu.float() / 128 - 1
    d = delta.clone().requires_grad_(True)
    logits = i3d_ref.i3d_logits(am.tf_apply(x, d), Wt[torch.float32])
    label = logits.argmax(-1)
    loss, _, _ = am.tf_improve_adversarial_loss(logits, label, 0.05, False, False)
    (g,) = torch.autograd.grad(loss, d)
    eng = FlickerI3D(W, batch_size=1, frames=Tn, dtype="f32")
    eng.reset_perturbation(delta.numpy())
    got = eng.logits(xu.cuda(), adv_flag=1.0).cpu()
    assert rel_err(got, logits.detach()) < 1e-3
    r = eng.step(xu.cuda(), label.cuda(), update=False, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
    assert float(r["adv_loss"]) == pytest.approx(loss.item(), rel=1e-3, abs=1e-6)
    gg = eng.delta_gradient().cpu().reshape(g.shape)
    cos = float(torch.nn.functional.cosine_similarity(gg.double().flatten(), g.double().flatten(), 0))
    print(f"T={Tn}: logits rel err {rel_err(got, logits.detach()):.2e}, d(loss)/d(delta) cosine {cos:.6f}, max-rel {rel_err(gg, g):.2e}")
    assert cos > 0.999 and rel_err(gg, g) < 3e-2
    # the bf16 mode on the same odd sizes (fixed-point pool accumulation, fallback kernels), two clips per batch
    e16 = FlickerI3D(W, batch_size=2, frames=Tn, dtype="bf16")
    e16.reset_perturbation(delta.numpy())
    xu2 = torch.cat([xu, xu]).cuda()
    got16 = e16.logits(xu2, adv_flag=1.0).cpu()
    assert torch.equal(got16[0], got16[1]) and rel_err(got16[:1], logits.detach()) < 5e-2
    e16.step(xu2, torch.cat([label, label]).cuda(), update=False, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
    g16 = e16.delta_gradient().cpu().reshape(g.shape) / 2          # the margin loss is a SUM over the (identical) clips
    cos16 = float(torch.nn.functional.cosine_similarity(g16.double().flatten(), g.double().flatten(), 0))
    print(f"T={Tn} bf16: logits rel err {rel_err(got16[:1], logits.detach()):.2e}, gradient cosine {cos16:.4f}")
    assert cos16 > 0.90                          # measured 0.930 (T = 90), 0.958 (T = 18)


# ---------------------------------------------------------------------------------------------------------------------------------
# The BENCHMARKED geometry (BASELINE.json: I3D, 64 x 224 x 224, bs 1 and bs 8) against the ORACLE -- not against the HIP path itself
# (tests/test_fullsize_gpu.py holds the self-consistency properties).  At this size the launch layouts differ from the 16-frame
# tests: other tile shapes and wave grids, split-K at bs 1, 192-row tiles, the half-batch stem split on two streams at bs 8.
# The torch-CPU fp32 oracle costs ~1 s (bs 1) / ~10 s (bs 8) per forward + backward on the GPU box's 16 host cores.
T64 = 64


def _oracle32(W, xu, delta, n_threads=16, ep_grads=False):
    """torch-CPU fp32 oracle pass on a [B,T,224,224,3] uint8 clip: logits, every endpoint, summed margin loss, d(loss)/d(delta);
    ep_grads: also d(loss)/d(pre-ReLU endpoint) for every endpoint (what the HIP gradient buffers hold)"""
    torch.set_num_threads(min(n_threads, torch.get_num_threads() if torch.get_num_threads() > 1 else n_threads))
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    x = xu.float() / 128 - 1
    d = delta.clone().requires_grad_(True)
    logits, ep = i3d_ref.i3d_logits(am.tf_apply(x, d), Wt, return_endpoints=True)
    label = logits.argmax(-1)
    loss, _, _ = am.tf_improve_adversarial_loss(logits, label, 0.05, False, False)
    names = [n for n in GRAD_ENDPOINTS if n in ep] if ep_grads else []
    g, *ge = torch.autograd.grad(loss, [d] + [ep[n] for n in names])
    out = dict(logits=logits.detach(), ep={k: v.detach() for k, v in ep.items()}, loss=float(loss), label=label, g=g)
    if ep_grads:     # endpoints of Unit3D / Inception blocks are ReLU outputs: the pre-ReLU gradient is the post-ReLU one where the unit is on
        out["ge"] = {n: (gg * (ep[n].detach() > 0) if "MaxPool" not in n else gg) for n, gg in zip(names, ge)}
    return out


def _delta64(seed):
    d = torch.from_numpy(np.random.default_rng(seed).uniform(-0.06, 0.06, (T64, 1, 1, 3)).astype(np.float32))
    d[5] = 0.45          # beyond the +-0.4 clip: no gradient there
    return d


@pytest.mark.parametrize("batch", [1, 8])
def test_benchmark_geometry_fp32_vs_oracle(batch):
    """fp32 mode at 64 x 224 x 224, bs 1 and the headline bs 8 (distinct clips, ONE shared delta): every forward endpoint, the
    logits and the summed adversarial loss at 1e-3 of the torch-CPU fp32 oracle, d(loss)/d(delta) by cosine > 0.999 and max-rel
    < 3e-2 (the bars of the T = 90 test: on the random-sign weights two fp32 implementations differ in a few ReLU / max-pool
    decisions, which moves single gradient entries by ~1e-2 -- see the module docstring)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W = i3d_spec.synthetic_i3d_weights(42)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(batch, T64, seed=1234))        # bench.py's clips (seed 1234 + rank 0)
    delta = _delta64(64)
    ref = _oracle32(W, xu, delta)
    eng = FlickerI3D(W, batch_size=batch, frames=T64, dtype="f32")
    eng.reset_perturbation(delta.numpy())
    r = eng.step(xu.cuda(), ref["label"].cuda(), update=False, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
    worst = 0.0
    for name, want in ref["ep"].items():
        got = torch.from_numpy(eng.net.activation(name))
        e = rel_err(got, want.permute(0, 2, 3, 4, 1))
        worst = max(worst, e)
        assert e < 1e-3, f"bs {batch}: endpoint {name}: {e:.3e}"
        del got
    e_l = rel_err(eng._logits.cpu(), ref["logits"])
    e_loss = abs(float(r["adv_loss"]) - ref["loss"]) / abs(ref["loss"])
    g = eng.delta_gradient().cpu().reshape(ref["g"].shape)
    cos = float(torch.nn.functional.cosine_similarity(g.double().flatten(), ref["g"].double().flatten(), 0))
    e_g = rel_err(g, ref["g"])
    print(f"bs {batch}, T {T64}, fp32 vs torch-CPU fp32 oracle: worst endpoint {worst:.2e}, logits {e_l:.2e}, adv loss {e_loss:.2e} "
          f"({float(r['adv_loss']):.6f} vs {ref['loss']:.6f}), d(loss)/d(delta) cosine {cos:.6f} max-rel {e_g:.2e}")
    assert e_l < 1e-3 and e_loss < 1e-3
    assert cos > 0.999 and e_g < 3e-2
    assert g[5].abs().max() == 0
    assert torch.equal(r["argmax"].cpu(), ref["logits"].argmax(-1))


def test_benchmark_geometry_bf16_vs_rounded_oracle():
    """The TIMED dtype at 64 x 224 x 224, bs 1, through the engine's own iteration (centred clip + position-bias stem forward, fused
    stem delta-gradient, split-K plan), against the bf16-rounded oracle of oracle_links, link by link on the HIP path's own
    inputs, at the bf16 tolerances of the 16-frame test: every forward link 1.6e-2 of the endpoint maximum; the final
    d(loss)/d(delta) link (stem backward + clip masks + (b,h,w) reduction, fed the HIP path's own gradient of Conv3d_1a) 2e-2.
    End to end against the fp32 oracle: logits and loss 5e-2 (stated), gradient cosine reported and asserted at its measured
    value minus a margin."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    W = i3d_spec.synthetic_i3d_weights(42)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T64, seed=1234))
    delta = _delta64(64)
    ref = _oracle32(W, xu, delta, ep_grads=True)
    eng = FlickerI3D(W, batch_size=1, frames=T64, dtype="bf16")
    assert eng.fused_delta_grad and eng.exact_delta_forward
    eng.reset_perturbation(delta.numpy())
    r = eng.step(xu.cuda(), ref["label"].cuda(), update=False, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
    logits = eng._logits.cpu()
    hip_act = lambda n: torch.from_numpy(eng.net.activation(n)).permute(0, 4, 1, 2, 3).contiguous()
    d0 = delta.clone().requires_grad_(True)
    # the exact-delta forward hands the stem the UNROUNDED perturbed clip (bf16 clean pixel + fp32 perturbation): the link's input is
    # the fp32 clip, not its bf16 rounding
    prev_name, prev = "delta", am.tf_apply(xu.float() / 128 - 1, d0).permute(0, 4, 1, 2, 3).contiguous()
    links = oracle_links(W, True, dt=torch.float32)
    name0, fn0 = links[0]
    Wd32 = {k: torch.from_numpy(v).to(torch.bfloat16).float() if (k.endswith("/w") and "Logits" not in k) else torch.from_numpy(v) for k, v in W.items()}
    stem = lambda x, relu: i3d_ref.unit3d(x, Wd32, "Conv3d_1a_7x7", (7, 7, 7), (2, 2, 2), relu=relu)
    worst = 0.0
    for name, fn in links:
        with torch.no_grad():
            out = stem(prev, True).to(torch.bfloat16).float() if name == name0 else fn(prev, True)
        got = logits if name == "Logits" else hip_act(name)
        e_f = rel_err(out, got)
        worst = max(worst, e_f)
        print(f"[bf16 T={T64}] forward link {prev_name:>17s} -> {name:<17s} max-rel {e_f:.2e}")
        assert e_f < 1.6e-2, f"forward link {prev_name} -> {name}: {e_f:.3e}"
        if name == name0:
            got_g = hip_act("grad:" + name0)                         # d(loss)/d(pre-ReLU stem output), HIP
            (g_ref,) = torch.autograd.grad(stem(prev, False), d0, grad_outputs=got_g)
            g_hip = eng.delta_gradient().cpu().reshape(g_ref.shape)
            e_g = rel_err(g_hip, g_ref)
            print(f"[bf16 T={T64}] d(loss)/d(delta) link (fused stem kernel): max-rel {e_g:.2e}")
            assert e_g < 2e-2
        prev_name = name
        if name != "Logits":
            prev = got
    e_l = rel_err(logits, ref["logits"])
    e_loss = abs(float(r["adv_loss"]) - ref["loss"]) / abs(ref["loss"])
    g = eng.delta_gradient().cpu().reshape(ref["g"].shape)
    cos = float(torch.nn.functional.cosine_similarity(g.double().flatten(), ref["g"].double().flatten(), 0))
    print(f"[bf16 T={T64}] end to end vs fp32 oracle: logits {e_l:.2e}, adv loss {e_loss:.2e}, d(loss)/d(delta) cosine {cos:.4f}")
    # where the end-to-end cosine is lost (diagnostic, printed): the HIP bf16 gradient buffer of every endpoint against the fp32 oracle's
    # gradient of the same endpoint, from the logits down -- the running product of 57 layers' bf16 roundings on random-sign weights
    for name in reversed([n for n in GRAD_ENDPOINTS if n in ref["ge"]]):
        gh, go = hip_act("grad:" + name).double().flatten(), ref["ge"][name].double().flatten()
        print(f"[bf16 T={T64}] running gradient at {name:<17s}: cosine vs fp32 oracle {float(torch.nn.functional.cosine_similarity(gh, go, 0)):.4f}, "
              f"|hip| / |oracle| {float(gh.norm() / go.norm()):.4f}")
    assert e_l < 5e-2 and e_loss < 5e-2
    assert cos > 0.88                            # measured 0.910
    assert g[5].abs().max() == 0


def test_benchmark_geometry_bf16_bs8_stem_links_vs_rounded_oracle():
    """The headline batch itself (bs 8, 64 x 224 x 224, bf16, ONE shared delta): the stem segment runs per HALF of the batch on two
    streams (stem-from-uint8 forward per half, ONE fused delta-gradient kernel over both halves' gradient buffer).  Both links against
    the bf16-rounded oracle on the HIP path's own tensors: the stem forward of every clip (1.6e-2 of the endpoint maximum) and the
    delta-gradient link -- stem backward + clip masks + (b,h,w) reduction over all 8 clips, fed the HIP gradient of Conv3d_1a -- at 2e-2."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    B = 8
    W = i3d_spec.synthetic_i3d_weights(42)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T64, seed=1234))
    delta = _delta64(64)
    eng = FlickerI3D(W, batch_size=B, frames=T64, dtype="bf16")
    assert eng.fused_delta_grad and eng.exact_delta_forward
    eng.reset_perturbation(delta.numpy())
    labels = eng.logits(xu.cuda(), adv_flag=0.0).argmax(-1).clone()
    eng.step(xu.cuda(), labels, update=False, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
    hip_act = lambda n: torch.from_numpy(eng.net.activation(n)).permute(0, 4, 1, 2, 3).contiguous()
    Wd32 = {k: torch.from_numpy(v).to(torch.bfloat16).float() if (k.endswith("/w") and "Logits" not in k) else torch.from_numpy(v) for k, v in W.items()}
    stem = lambda x, relu: i3d_ref.unit3d(x, Wd32, "Conv3d_1a_7x7", (7, 7, 7), (2, 2, 2), relu=relu)
    torch.set_num_threads(16)
    d0 = delta.clone().requires_grad_(True)
    xin = am.tf_apply(xu.float() / 128 - 1, d0).permute(0, 4, 1, 2, 3).contiguous()
    pre = stem(xin, False)
    got = hip_act("Conv3d_1a_7x7")
    with torch.no_grad():
        want = torch.relu(pre.detach()).to(torch.bfloat16).float()
    for b in range(B):
        e_f = rel_err(want[b], got[b])
        print(f"[bf16 bs 8 T={T64}] stem forward, clip {b} (half {b // 4}): max-rel {e_f:.2e}")
        assert e_f < 1.6e-2
    (g_ref,) = torch.autograd.grad(pre, d0, grad_outputs=hip_act("grad:Conv3d_1a_7x7"))
    g_hip = eng.delta_gradient().cpu().reshape(g_ref.shape)
    e_g = rel_err(g_hip, g_ref)
    print(f"[bf16 bs 8 T={T64}] d(loss)/d(delta) link (fused stem kernel over both half-batches): max-rel {e_g:.2e}")
    assert e_g < 2e-2
    assert g_hip[5].abs().max() == 0


@pytest.mark.parametrize("dtype,batch", [("bf16", 4), ("bf16", 1), ("f32", 2)])
def test_forward_apply_equals_apply_then_forward(dtype, batch, monkeypatch):
    """flk_net_forward_apply (the plan launches the perturbation apply itself, per half of the batch on that half's stream) against
    flk_perturb_apply_s2d followed by flk_net_forward[_flicker]: the space-to-depth clip and the logits are bitwise equal -- shared and
    per-clip perturbations.  (FLK_STEM_U8=0: in bf16 the plan's default reads the uint8 clip in the stem itself and leaves the
    space-to-depth tensor unwritten -- tests/test_stem_fwd_gpu.py.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("FLK_STEM_U8", "0")
    from flickering_adversarial_video_amd import i3d_spec, ops
    from flickering_adversarial_video_amd._lib import FLK_NET_I3D
    W = i3d_spec.synthetic_i3d_weights(42)
    net = ops.Net(FLK_NET_I3D, dtype, batch, T, 224, 224, W)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(batch, T, seed=8)).cuda()
    rng = np.random.default_rng(1)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    for shape in ((T, 3), (batch, T, 3)):
        d = torch.from_numpy(rng.uniform(-0.3, 0.3, shape).astype(np.float32)).cuda()
        a = ops.make_apply_args(xu, d, fold_t=ops.I3D_FOLD, center=bool(net.has_forward_flicker))
        x1 = ops.perturb_apply_s2d(a, dtype, torch.empty((batch, T // 2, 112, 112, 32), dtype=tdt, device="cuda"))
        l1 = (net.forward_flicker(x1, a) if a.center else net.forward(x1)).clone()
        x2 = torch.zeros_like(x1)
        l2 = net.forward_apply(a, x2)
        assert torch.equal(x1, x2) and torch.equal(l1, l2), shape


def test_operator_launch_count_may_change_between_runs():
    """The plan arms fork / join stop events on operators that launched ONE kernel in the previous run (net.cpp run_ops).  The stem
    operator varies: flk_net_forward launches the convolution only, flk_net_forward_apply (fp32: apply + convolution) two kernels.
    Alternating the two entry points -- the armed operator launches more kernels than the run before -- must give the bits of a
    fresh plan that only ever ran one of them (an event riding on the FIRST of two kernels would let the waiting stream start early)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec, ops
    from flickering_adversarial_video_amd._lib import FLK_NET_I3D
    B = 4                                                 # the stem segment runs per half-batch on two streams
    W = i3d_spec.synthetic_i3d_weights(42)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T, seed=9)).cuda()
    d = torch.from_numpy(np.random.default_rng(3).uniform(-0.2, 0.2, (T, 3)).astype(np.float32)).cuda()

    def run(seq):
        net = ops.Net(FLK_NET_I3D, "f32", B, T, 224, 224, W)
        a = ops.make_apply_args(xu, d, fold_t=ops.I3D_FOLD, center=False)
        x = ops.perturb_apply_s2d(a, "f32", torch.empty((B, T // 2, 112, 112, 32), dtype=torch.float32, device="cuda"))
        outs = []
        for kind in seq:
            outs.append((net.forward(x) if kind == "fwd" else net.forward_apply(a, torch.empty_like(x))).clone())
        torch.cuda.synchronize()
        return outs

    ref = run(["fwd"])[0]
    for got in run(["fwd", "apply", "fwd", "apply", "apply", "fwd"]):
        assert torch.equal(got, ref)


@pytest.mark.gpu
def test_stem_delta_gradient_per_half_batch_equals_one_launch():
    """bf16 plans with an even batch >= 4 run the fused stem delta-gradient once per HALF of the batch, on the halves' own streams, and
    stage 2 behind the join (net.cpp: "Conv3d_1a_7x7/dgrad/half", flk_stem_delta_grad_part / _finish).  With the per-launch profile on the
    plan runs serially and takes the one-launch path: same mask, same products, the stage-1 partials summed in another chunking -- the two
    gradients agree to fp32 summation order (stated 2e-5 of the largest entry; measured 1e-7).  Alternating the two modes leaves every
    run of a mode bitwise equal to the first one (the half-batch operators launch nothing in the serial runs; the stop events armed on them
    follow the launch count of the previous run)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    B = 4
    eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=B, frames=T, dtype="bf16")
    assert eng.fused_delta_grad
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T, seed=21)).cuda()
    eng.perturbation.copy_(torch.from_numpy(np.random.default_rng(5).uniform(-0.1, 0.1, (T, 1, 1, 3)).astype(np.float32)))
    labels = eng.logits(xu, adv_flag=0.0).argmax(-1).clone()

    def grad(serial):
        eng.net.profile(serial)
        eng.step(xu, labels, update=False)
        g = eng.delta_gradient().clone()
        torch.cuda.synchronize()
        eng.net.profile(False)
        return g

    halves, whole = grad(False), grad(True)
    scale = float(whole.abs().max())
    assert scale > 0
    err = float((halves - whole).abs().max()) / scale
    print(f"stem delta-gradient, two half-batch launches vs one: max |diff| / max |g| = {err:.2e}")
    assert err <= 2e-5
    for serial, ref in ((False, halves), (True, whole), (False, halves), (False, halves), (True, whole)):
        assert torch.equal(grad(serial), ref)

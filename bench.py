#!/usr/bin/env python3
"""Headline benchmark: I3D flickering-attack iterations on MI355X (BASELINE.json metric).

One "step" = one attack iteration on one batch of synthetic clips resident in HBM:
    apply(delta) -> I3D forward -> adversarial loss -> backward-to-delta -> [RCCL all-reduce of the (T x 3)
    delta-gradient when N > 1] -> regulariser + Adam
(1 forward + 1 data-gradient backward; the reference's 2 extra redundant forwards per step are not reproduced).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  value = clip attack-iterations per second, whole job (B clips x N ranks per step);
conv TFLOP/s uses the algorithmic work of SURVEY.md 8(d): 444.6 GFLOP per clip per iteration at T=64 (fwd + dgrad).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md


def conv_gflop_per_clip(T):
    """algorithmic fwd + dgrad conv GFLOP per clip per iteration (111.15 GMAC @ T=64, linear in T)"""
    return 4 * 111.15 * T / 64.0


def _cpu_cores():
    # host cores this process may use (the 1-GPU boxes grant a 16-core share of a much larger host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("FLK_CPU_BASELINE_THREADS", "16"))))


def _cpu_iterations(W, xu, n_timed, budget_s):
    """n_timed (+1 warm-up) iterations of the oracle's attack iteration from delta = 0 on the uint8 clips xu [B,T,224,224,3];
    returns (seconds per iteration, n, first-iteration record {labels, logits, adv_loss}, last adv loss)"""
    from oracle import attack_math as am
    from oracle import i3d_ref
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    T = xu.shape[1]
    x = xu.float() / 128 - 1
    d = torch.zeros(T, 1, 1, 3)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    label, first, times, t_start, it = None, None, [], time.time(), 0
    while True:
        t0 = time.time()
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x, dv), Wt)
        if label is None:
            label = lg.argmax(-1)
        adv, _, _ = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
        total, _ = am.tf_total_loss(adv, dv, 1.0, 0.5, 0.5, 0.5)
        (g,) = torch.autograd.grad(total, dv)
        if first is None:
            first = {"labels": label.clone(), "logits": lg.detach().clone(), "adv_loss": float(adv.detach())}
        d, m, v = am.tf_adam_step(d, g, m, v, it + 1)
        dt = time.time() - t0
        it += 1
        if it > 1:
            times.append(dt)          # first iteration is the warm-up
        if len(times) >= n_timed or (time.time() - t_start > budget_s and len(times) >= 1):
            break
    return float(np.mean(times)), len(times), first, float(adv.detach())


def _same_clip_parity(W, xu, first, device, dtypes=("f32", "bf16")):
    """The GPU engine's FIRST iteration (delta = 0, the oracle's own labels) on the SAME uint8 clips the CPU baseline just ran,
    at the benchmark geometry: logits and adversarial loss relative to the oracle's (fp32 mode: the north-star 1e-3; the timed bf16
    mode: stated 5e-2).  oracle/ supplies the reference values here, nothing on the product path."""
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    B, T = xu.shape[0], xu.shape[1]
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))
    out = {"clips": B, "frames": T, "from": "delta = 0, labels = the oracle's clean argmax"}
    for dt in dtypes:
        eng = FlickerI3D(W, batch_size=B, frames=T, dtype=dt, device=device)
        r = eng.step(xu.cuda(), first["labels"].cuda(), update=False, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
        out[dt] = {"logits_rel_err": rel(eng._logits.cpu(), first["logits"]),
                   "adv_loss_rel_err": abs(float(r["adv_loss"]) - first["adv_loss"]) / max(abs(first["adv_loss"]), 1e-30),
                   "adv_loss": float(r["adv_loss"]), "oracle_adv_loss": first["adv_loss"],
                   "argmax_equal": bool(torch.equal(r["argmax"].cpu(), first["logits"].argmax(-1))),
                   "tolerance": 1e-3 if dt == "f32" else 5e-2}
        out[dt]["ok"] = bool(out[dt]["logits_rel_err"] < out[dt]["tolerance"] and out[dt]["adv_loss_rel_err"] < out[dt]["tolerance"])
        del eng
        torch.cuda.empty_cache()
    return out


def cpu_baseline(W, xu, device, budget_s=30.0):
    """The CPU restatement (oracle/, torch-CPU fp32, the host cores of the GPU box) timed on the SAME iteration and the SAME clips
    as the GPU run (xu = the benchmark's uint8 batch): bs = 1 (clip 0; 3 timed iterations) and the headline bs = 8 (2 timed
    iterations, ~10 s each).  The first oracle iteration of each doubles as the parity reference for a GPU step on the same clip
    (`parity_same_clip`).  This is the only place bench.py touches oracle/ besides the parity check below: it is the baseline
    being reported and the checker, never the product path."""
    cores = _cpu_cores()
    torch.set_num_threads(cores)
    T, B = xu.shape[1], xu.shape[0]
    sec1, n1, first1, adv1 = _cpu_iterations(W, xu[:1], 3, budget_s)
    out = {"value": 1.0 / sec1, "unit": "clip-iters/s", "cores": cores, "kind": "port",
           "sample": f"{n1} timed iterations (+1 warm-up) of the same I3D attack iteration on clip 0 of the benchmark batch (bs=1), T={T}, 224x224, torch-CPU fp32",
           "sec_per_iter": sec1, "conv_gflops": conv_gflop_per_clip(T) / sec1, "adv_loss_last": adv1,
           "parity_same_clip": _same_clip_parity(W, xu[:1], first1, device)}
    if B > 1 and sec1 * B <= 60.0:                                    # "and bs=8 if <= 60 s/iter"
        sec8, n8, first8, _ = _cpu_iterations(W, xu, 2, 3 * budget_s)
        out[f"bs{B}"] = {"value": B / sec8, "unit": "clip-iters/s", "sec_per_iter": sec8, "conv_gflops": B * conv_gflop_per_clip(T) / sec8,
                         "sample": f"{n8} timed iterations (+1 warm-up) on the benchmark batch itself (bs={B}), T={T}",
                         "parity_same_clip": _same_clip_parity(W, xu, first8, device)}
    return out


def parity_check(device):
    """In-run parity of the product path against the oracle (SURVEY 8(d) / BASELINE.md 3), fp32 mode, 16-frame clips:
      * logits, adversarial loss of one iteration on the benchmark's own seeded weights (noise clip): 1e-3 relative;
      * logits, adversarial loss and the LEARNED DELTA over 3 iterations on the well-conditioned fixture (oracle/fixtures.py: the
        random-sign benchmark weights make d(loss)/d(delta) a cancelling sum that two fp32 implementations reproduce only to ~5e-3,
        tests/test_i3d_gpu.py) against the fp64 oracle trajectory: 1e-3 relative.
    Returns the measured errors; "ok" is their conjunction.  oracle/ is the checker here, never the thing measured."""
    from oracle import attack_math as am
    from oracle import fixtures, i3d_ref
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    torch.set_num_threads(_cpu_cores())
    T = 16
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))
    out = {"frames": T, "dtype": "f32", "tolerance": 1e-3}
    t0 = time.time()
    # (1) the benchmark's weights
    W = i3d_spec.synthetic_i3d_weights(42)
    xu = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=1234))
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    lg = i3d_ref.i3d_logits(xu.float() / 128 - 1, Wt)
    label = lg.argmax(-1)
    adv, _, _ = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype="f32", device=device)
    r = eng.step(xu.cuda(), label.cuda(), update=False)
    out["noise_weights"] = {"logits_rel_err": rel(eng._logits.cpu(), lg), "adv_loss_rel_err": abs(float(r["adv_loss"]) - adv.item()) / abs(adv.item())}
    del eng
    # (2) learned delta on the well-conditioned fixture, fp64 oracle trajectory
    Wc = fixtures.coherent_i3d_weights(xu, seed=5, label=233)
    W64 = {k: torch.from_numpy(v).double() for k, v in Wc.items()}
    x64 = xu.double() / 128 - 1
    label = i3d_ref.i3d_logits(x64, W64).argmax(-1)
    eng = FlickerI3D(Wc, batch_size=1, frames=T, dtype="f32", device=device)
    eng16 = FlickerI3D(Wc, batch_size=1, frames=T, dtype="bf16", device=device)     # the timed mode, same trajectory, stated tolerance 5e-2
    d = torch.zeros(T, 1, 1, 3, dtype=torch.float64)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    e_delta = e_logits = e_adv = 0.0
    b_delta = b_logits = b_adv = 0.0
    for it in range(1, 4):
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x64, dv), W64)
        adv, _, _ = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
        total, _ = am.tf_total_loss(adv, dv, 1.0, 0.5, 0.5, 0.5)
        (g,) = torch.autograd.grad(total, dv)
        d, m, v = am.tf_adam_step(d, g, m, v, it)
        r = eng.step(xu.cuda(), label.cuda(), lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
        e_logits = max(e_logits, rel(eng._logits.cpu(), lg.detach()))
        e_adv = max(e_adv, abs(float(r["adv_loss"]) - adv.item()) / abs(adv.item()))
        e_delta = max(e_delta, rel(eng.perturbation.cpu(), d))
        r = eng16.step(xu.cuda(), label.cuda(), lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
        b_logits = max(b_logits, rel(eng16._logits.cpu(), lg.detach()))
        b_adv = max(b_adv, abs(float(r["adv_loss"]) - adv.item()) / abs(adv.item()))
        b_delta = max(b_delta, rel(eng16.perturbation.cpu(), d))
    out["bf16_well_conditioned_fixture"] = {"iterations": 3, "tolerance": 5e-2, "delta_rel_err": b_delta, "logits_rel_err": b_logits,
                                            "adv_loss_rel_err": b_adv, "exact_delta_forward": bool(eng16.exact_delta_forward)}
    del eng16
    out["well_conditioned_fixture"] = {"iterations": 3, "delta_rel_err": e_delta, "logits_rel_err": e_logits, "adv_loss_rel_err": e_adv,
                                       "reference": "fp64 oracle trajectory"}
    del eng
    torch.cuda.empty_cache()
    errs = list(out["noise_weights"].values()) + [e_delta, e_logits, e_adv]
    out["ok"] = bool(max(errs) < 1e-3 and max(b_delta, b_logits, b_adv) < 5e-2)
    out["seconds"] = time.time() - t0
    return out


def _plan_roofline(net, stepfn, reps=2):
    """roofline block of a plan from its per-launch profile (HIP events around every launch, serial mode): the convolution-class
    launches' algorithmic flops over their summed durations against the bf16 MFMA peak, and the three slowest launches"""
    net.profile(True)
    acc = {}
    for _ in range(reps):
        stepfn()
        for i, r in enumerate(net.profile_read()):
            e = acc.setdefault((i, r["name"], r["pass"], r["kind"]), {"ms": 0.0, "flops": r["flops"]})
            e["ms"] += r["ms"] / reps
    net.profile(False)
    conv = {k: v for k, v in acc.items() if k[3] == "conv"}
    ms, fl = sum(v["ms"] for v in conv.values()), sum(v["flops"] for v in conv.values())
    slow = sorted(conv.items(), key=lambda kv: -kv[1]["ms"])[:3]
    return {"kernel": "conv_igemm_kernel family (every convolution-class launch of the plan)", "bound": "mfma", "achieved": fl / ms / 1e9, "peak": PEAK_TFLOPS["bf16"],
            "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_TFLOPS["bf16"], "traffic": None, "launches_per_step": len(conv), "conv_ms_per_step": ms,
            "serial_ms_all_kernels": sum(v["ms"] for v in acc.values()),
            "slowest": [{"op": k[1], "pass": k[2], "ms": v["ms"], "TFLOPs": v["flops"] / v["ms"] / 1e9} for k, v in slow]}


def other_configs(device, steps=10, warmup=3):
    """BASELINE configs 2 and 3 in the same run (single-video attacks: no collective): ms per iteration and conv TFLOP/s; the reference's
    own shapes beside them -- I3D at its default clip length T = 90 (kinetics_i3d_utils.py:12) and the universal attack's mc3_18 at 16
    clips per device (r2plus1d_main_universal_attack.py:30-33,130-149) -- each with a roofline block from its per-launch profile"""
    from flickering_adversarial_video_amd import i3d_spec, videoresnet_spec as vs
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses
    out = {}

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=64, dtype="bf16", device=device)
    x = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, 64, seed=1234)).cuda()
    labels = eng.logits(x, adv_flag=0.0).argmax(-1).clone()
    sec = timed(lambda: eng.step(x, labels))
    out["config2_i3d_single_video_bs1_64x224x224_bf16"] = {"ms_per_iter": sec * 1e3, "iters_per_s": 1 / sec, "conv_tflops": conv_gflop_per_clip(64) / sec / 1e3,
                                                          "conv_frac_of_mfma_peak": conv_gflop_per_clip(64) / sec / 1e3 / PEAK_TFLOPS["bf16"],
                                                          "roofline": _plan_roofline(eng.net, lambda: eng.step(x, labels))}
    del eng
    # config 2 as the reference's users run it -- MANY videos, one after another (i3d_adversarial_main_single_video_npy.py:103-337) -- with 8
    # independent single-video attacks advancing in one batch (per-clip perturbations / Adam states; each video's trajectory is the one
    # it has alone): clip-iterations per second against the one-by-one loop above
    engb = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=8, frames=64, dtype="bf16", device=device, per_clip_delta=True)
    xb = torch.from_numpy(i3d_spec.synthetic_clip_u8(8, 64, seed=1234)).cuda()
    lb = engb.logits(xb, adv_flag=0.0).argmax(-1).clone()
    secb = timed(lambda: engb.step(xb, lb))
    out["config2_batched_8_independent_single_video_attacks"] = {"ms_per_iter": secb * 1e3, "clip_iters_per_s": 8 / secb,
                                                                  "speedup_over_one_by_one": (8 / secb) * sec,
                                                                  "conv_tflops": 8 * conv_gflop_per_clip(64) / secb / 1e3}
    del engb, xb
    torch.cuda.empty_cache()
    W = vs.synthetic_weights("r2plus1d_18", 42)
    eng = FlickerVideoResNet("r2plus1d_18", W, batch_size=1, sample_length=16, image_size=112, dtype="bf16", device=device)
    xv = torch.from_numpy(vs.synthetic_clip(1, 16, seed=1234)).cuda()
    lab = eng.logits(xv).argmax(-1).clone()
    crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True, logits=True)
    sec = timed(lambda: eng.step(xv, lab, crit))
    out["config3_r2plus1d_18_single_video_bs1_16x112x112_bf16"] = {"ms_per_iter": sec * 1e3, "iters_per_s": 1 / sec, "conv_tflops": VRN_GFLOP["r2plus1d_18"] / sec / 1e3,
                                                                  "conv_frac_of_mfma_peak": VRN_GFLOP["r2plus1d_18"] / sec / 1e3 / PEAK_TFLOPS["bf16"],
                                                                  "roofline": _plan_roofline(eng.net, lambda: eng.step(xv, lab, crit))}
    del eng
    # config 3 with 8 independent single-video attacks per batch (fit_many_videos(batch): per-clip perturbation / clamp bound / Adam state)
    engb = FlickerVideoResNet("r2plus1d_18", W, batch_size=8, sample_length=16, image_size=112, dtype="bf16", device=device, per_clip=True)
    xb = torch.from_numpy(vs.synthetic_clip(8, 16, seed=1234)).cuda()
    lb = engb.logits(xb).argmax(-1).clone()
    secb = timed(lambda: engb.step(xb, lb, crit))
    out["config3_batched_8_independent_single_video_attacks"] = {"ms_per_iter": secb * 1e3, "clip_iters_per_s": 8 / secb,
                                                                  "speedup_over_one_by_one": (8 / secb) * sec,
                                                                  "conv_tflops": 8 * VRN_GFLOP["r2plus1d_18"] / secb / 1e3,
                                                                  "conv_frac_of_mfma_peak": 8 * VRN_GFLOP["r2plus1d_18"] / secb / 1e3 / PEAK_TFLOPS["bf16"],
                                                                  "roofline": _plan_roofline(engb.net, lambda: engb.step(xb, lb, crit))}
    del engb, xb
    torch.cuda.empty_cache()
    # the reference's universal attack on torchvision's mc3_18 (BASE_MODEL = "mc3_18", 16-20 clips per device,
    # r2plus1d_main_universal_attack.py:30-33,130-149): ONE shared perturbation, batch 16, 16 x 112 x 112
    Wm = vs.synthetic_weights("mc3_18", 42)
    engm = FlickerVideoResNet("mc3_18", Wm, batch_size=16, sample_length=16, image_size=112, dtype="bf16", device=device)
    xm = torch.from_numpy(vs.synthetic_clip(16, 16, seed=1234)).cuda()
    lm = engm.logits(xm).argmax(-1).clone()
    secm = timed(lambda: engm.step(xm, lm, crit))
    out["mc3_18_universal_bs16_16x112x112_bf16"] = {"ms_per_iter": secm * 1e3, "clip_iters_per_s": 16 / secm, "conv_tflops": 16 * VRN_GFLOP["mc3_18"] / secm / 1e3,
                                                    "conv_frac_of_mfma_peak": 16 * VRN_GFLOP["mc3_18"] / secm / 1e3 / PEAK_TFLOPS["bf16"],
                                                    "roofline": _plan_roofline(engm.net, lambda: engm.step(xm, lm, crit))}
    del engm, xm
    torch.cuda.empty_cache()
    # I3D at the reference's default clip length (_SAMPLE_VIDEO_FRAMES = 90, kinetics_i3d_utils.py:12): T/2 = 45, 23, 12 are odd, so
    # the strided pools take their general-shape forms
    for bsz in (8, 1):
        e90 = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=bsz, frames=90, dtype="bf16", device=device)
        x90 = torch.from_numpy(i3d_spec.synthetic_clip_u8(bsz, 90, seed=1234)).cuda()
        l90 = e90.logits(x90, adv_flag=0.0).argmax(-1).clone()
        s90 = timed(lambda: e90.step(x90, l90))
        out[f"i3d_T90_bs{bsz}_90x224x224_bf16"] = {"ms_per_iter": s90 * 1e3, "clip_iters_per_s": bsz / s90, "conv_tflops": bsz * conv_gflop_per_clip(90) / s90 / 1e3,
                                                    "conv_frac_of_mfma_peak": bsz * conv_gflop_per_clip(90) / s90 / 1e3 / PEAK_TFLOPS["bf16"],
                                                    "roofline": _plan_roofline(e90.net, lambda: e90.step(x90, l90))}
        del e90, x90
        torch.cuda.empty_cache()
    return out


VRN_GFLOP = {"r2plus1d_18": 4 * 40.52, "r3d_18": 4 * 40.70, "mc3_18": 4 * 43.34}   # fwd + dgrad per clip @16x112x112


def bench_videoresnet(a, world, rank, local_rank):
    """BASELINE config 3: single-video attack on a torchvision VideoResNet, 16x112x112 (replicas only: no collective)."""
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses
    B, T = a.batch, a.frames
    W = vs.synthetic_weights(a.model, 42)
    eng = FlickerVideoResNet(a.model, W, batch_size=B, sample_length=T, image_size=112, dtype=a.dtype, device=local_rank)
    x = torch.from_numpy(vs.synthetic_clip(B, T, seed=1234 + rank)).cuda()
    labels = eng.logits(x).argmax(-1).clone()
    crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True, logits=True)      # r2plus1d_main_statistics_*.py:32-65
    for _ in range(a.warmup):
        eng.step(x, labels, crit)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = eng.step(x, labels, crit)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    value = B * a.steps / elapsed
    tf = VRN_GFLOP[a.model] * (T / 16.0) * value / 1e3
    out = {"metric": f"attack-iters/sec ({a.model} {T}x112x112, bs={B} single-video attack) + 3D-conv TFLOP/s", "value": value,
           "unit": "clip-iters/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
           "data": "synthetic (seeded normalised fp32 clip resident in HBM, seeded weights)", "conv_tflops": tf,
           "conv_frac_of_mfma_peak": tf / PEAK_TFLOPS[a.dtype],
           "config": {"workload": f"{a.model} flickering attack iteration, {T}x112x112, bs={B}", "parallelism": "single (replicas only)"},
           "adv_loss_last": float(res["adv_loss"])}
    eng.net.profile(True)
    eng.step(x, labels, crit)
    prof = eng.net.profile_read()
    conv = [r for r in prof if r["kind"] == "conv"]
    ms, fl = sum(r["ms"] for r in conv), sum(r["flops"] for r in conv)
    out["roofline"] = {"kernel": "conv_igemm_kernel", "bound": "mfma", "achieved": fl / ms / 1e9, "peak": PEAK_TFLOPS[a.dtype],
                       "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_TFLOPS[a.dtype], "traffic": None, "launches_per_step": len(conv),
                       "conv_ms_per_step": ms}
    print(json.dumps(out))


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (parallel.launch_ranks: one CHILD process per GPU
    through torch.distributed.run, exactly the driver's multi-GPU form; this parent has not touched the GPU -- nothing above
    imports the extension or calls torch.cuda), relay their output (rank 0 prints the JSON line) and exit with their code."""
    from flickering_adversarial_video_amd.parallel import launch_ranks
    sys.exit(launch_ranks(n, __file__, sys.argv[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (BASELINE: bs=8)")
    ap.add_argument("--frames", type=int, default=64, help="frames per clip (BASELINE: 64; reference default 90)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="i3d", choices=["i3d", "r2plus1d_18", "r3d_18", "mc3_18"],
                    help="i3d = the headline config; the VideoResNet models are BASELINE config 3 (use --batch 1 --frames 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity check against the oracle")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs 2 and 3 (single-video bs=1 timings)")
    ap.add_argument("--autotune", action="store_true", help="time the candidate launch layouts of every convolution once before the warm-up "
                    "(measured: no gain over the built-in heuristics, which is why it is off by default)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return _self_launch(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # never report n_gpus != what was asked for: a SCALE run would record a mislabelled line
        print(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to run", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("FLK_BENCH_ECHO_RANKS"):
        print(f"[bench] rank {rank} of {world}", file=sys.stderr, flush=True)
    # FLK_DIST_BACKEND=gloo rehearses the multi-rank plumbing on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("FLK_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend)

    if a.model != "i3d":
        return bench_videoresnet(a, world, rank, local_rank)
    from flickering_adversarial_video_amd import i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D

    B, T = a.batch, a.frames
    W = i3d_spec.synthetic_i3d_weights(42)
    eng = FlickerI3D(W, batch_size=B, frames=T, dtype=a.dtype, device=local_rank)
    x = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T, seed=1234 + rank)).cuda()     # resident in HBM, uint8 (TFRecord path)
    labels = eng.logits(x, adv_flag=0.0).argmax(-1).clone()                                # "correctly classified" clips
    hp = dict(lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)           # run_config.yml defaults

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if a.autotune:
        eng.autotune(x)                                # launch-layout search per convolution, outside the timed region
    for _ in range(a.warmup):
        eng.step(x, labels, **hp)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = eng.step(x, labels, **hp)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    last = res.host()

    ms_per_step = elapsed / a.steps * 1e3
    value = B * world * a.steps / elapsed
    conv_tflops = conv_gflop_per_clip(T) * value / 1e3
    out = {
        "metric": f"attack-iters/sec (I3D {T}x224x224, bs={B} per GPU; clip attack-iterations per second, whole job) + 3D-conv TFLOP/s",
        "value": value, "unit": "clip-iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic (seeded uint8 clips resident in HBM, seeded He-normal weights)",
        "attack_iters_per_sec": a.steps / elapsed,
        "conv_tflops": conv_tflops, "conv_tflops_per_gpu": conv_tflops / world,
        "conv_frac_of_mfma_peak": conv_tflops / world / PEAK_TFLOPS[a.dtype],
        "config": {"workload": f"I3D-RGB flickering attack iteration (1 fwd + 1 dgrad bwd + Adam), {T}x224x224 clips, bs={B} per GPU",
                   "global_batch": B * world, "frames": T, "parallelism": f"dp{world}" if world > 1 else "single",
                   "collective": "1 RCCL all-reduce of (T*3+3) fp32 per step" if world > 1 else "none"},
        "adv_loss_last": float(last["adv_loss"]), "workspace_GiB": eng.net.workspace_bytes / 2**30,
    }

    if not a.no_roofline:
        # dominant kernel = conv_igemm (all convolutions, forward and data-gradient): HIP events around every launch,
        # on the stream the kernels run on, over 3 extra (untimed) steps.  EVERY rank runs these steps (each step ends in
        # the delta-gradient all-reduce: a rank that skipped them would leave rank 0 waiting in a collective nobody joins);
        # only rank 0 reports.
        eng.net.profile(True)
        per_kind = {}
        fused = {"ms": 0.0, "flops": 0.0, "launches": 0}     # the stem's data-gradient slot: stem_delta_grad_kernel when the engine fuses it
        # conv_igemm launches split by the roofline that bounds each of them (time at the MFMA peak for its flops against time at
        # the HBM peak for its compulsory bytes): the 3x3x3 / 7x7x7 layers are MFMA-bound, the 1x1x1 GEMMs HBM-bound
        by_bound = {"mfma": {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0}, "hbm": {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0}}
        by_kernel = {}       # convolution-class launches by the kernel they went to (flk_net_profile_read's "kernel")
        reps = 3
        for _ in range(reps):
            eng.step(x, labels, **hp)
            for r in eng.net.profile_read():
                k = per_kind.setdefault(r["kind"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["bytes"] += r["bytes"]; k["launches"] += 1
                if getattr(eng, "fused_delta_grad", False) and r["name"] == "Conv3d_1a_7x7/dgrad":
                    fused["ms"] += r["ms"]; fused["flops"] += r["flops"]; fused["launches"] += 1
                elif r["kind"] == "conv":
                    bb = by_bound["mfma" if r["flops"] / (PEAK_TFLOPS[a.dtype] * 1e12) >= r["bytes"] / 8e12 else "hbm"]
                    bb["ms"] += r["ms"]; bb["flops"] += r["flops"]; bb["bytes"] += r["bytes"]; bb["launches"] += 1
                    kk = by_kernel.setdefault(r.get("kernel") or "conv_igemm_kernel", {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                    kk["ms"] += r["ms"]; kk["flops"] += r["flops"]; kk["bytes"] += r["bytes"]; kk["launches"] += 1
        eng.net.profile(False)
        cv = per_kind["conv"]
        # the dominant kernel alone: conv_igemm_kernel (every 3x3x3 / strided / small convolution, forward and data-gradient).  The stem's
        # forward (stem_fwd_u8_kernel), the 1x1x1 GEMMs (conv1x1_dma_kernel) and the stem's data-gradient (stem_delta_grad_kernel) are
        # kernels of their own and get their own entries below.
        # Its grouped form (conv_igemm_group_kernel: the SAME device function, conv_igemm_body, run over two convolutions in one grid --
        # Branch_1 + Branch_2 of an Inception block) is counted with it: since the Mixed_3* blocks are grouped too, most of the body's
        # time is spent in grouped launches.  A grouped launch is ONE launch.
        IG = ("conv_igemm_kernel", "conv_igemm_group_kernel", "conv_pc_kernel")      # (conv_pc_kernel: the persistent producer / consumer form the large 3x3x3 launches take, round 5)
        if any(k in by_kernel for k in IG):
            ig = {f: sum(by_kernel[k][f] for k in IG if k in by_kernel) for f in ("ms", "flops", "bytes", "launches")}
        else:
            ig = {k: cv[k] - fused[k] for k in ("ms", "flops", "launches")}
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
        # HBM traffic per launch comes from the committed rocprofv3 --pmc passes of this same command (bench.py cannot
        # profile itself): profiles/*_pmc_hbm_traffic.json, produced by tools/pmc_summary.py -- the file and the library
        # state it was measured at are named so that staleness is visible
        traffic, traffic_src = None, None
        if B == 8 and T == 64 and a.dtype == "bf16":
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))[-1:]:
                tj = json.load(open(f))
                tk = [tj["kernels"][k] for k in IG if k in tj["kernels"]]
                traffic = sum(t["read_bytes"] + t["write_bytes"] for t in tk) / sum(t["launches"] for t in tk)
                traffic_src = {"file": os.path.relpath(f, ROOT), "state": tj.get("state", "unknown")}
        out["roofline"] = {"kernel": "conv_igemm_kernel + conv_igemm_group_kernel (conv_igemm_body) + conv_pc_kernel: implicit-GEMM conv3d fwd + dgrad with LDS halo tiles -- every "
                                     "3x3x3 / strided layer, as single launches, as grouped launches of an Inception block's Branch_1 + Branch_2, and (the large layers at the benchmark "
                                     "batch) as persistent launches with wave-specialised producers", "bound": "mfma",
                           "achieved": ach, "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[a.dtype],
                           "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, gfx950-corrected)", "traffic_source": traffic_src,
                           "compulsory_bytes_per_launch": ig["bytes"] / ig["launches"] if ig.get("bytes") else None,
                           "traffic_over_compulsory": traffic / (ig["bytes"] / ig["launches"]) if (traffic and ig.get("bytes")) else None,
                           "launches_per_step": ig["launches"] // reps, "avg_launch_ms": ig["ms"] / ig["launches"],
                           "conv_ms_per_step": cv["ms"] / reps, "algorithmic_gflop_per_step": cv["flops"] / reps / 1e9,
                           "all_conv_achieved": cv["flops"] / (cv["ms"] * 1e-3) / 1e12}
        oth = {}
        for kn, kv in by_kernel.items():
            if not kv["ms"]:
                continue                                # (the two forms of conv_igemm_body are listed here as well, one entry each)
            tf, gb = kv["flops"] / (kv["ms"] * 1e-3) / 1e12, kv["bytes"] / (kv["ms"] * 1e-3) / 1e9
            mf = kv["flops"] / (PEAK_TFLOPS[a.dtype] * 1e12) >= kv["bytes"] / 8e12
            oth[kn] = {"bound": "mfma" if mf else "hbm", "launches_per_step": kv["launches"] // reps, "ms_per_step": kv["ms"] / reps,
                       "avg_launch_ms": kv["ms"] / kv["launches"], "achieved_TFLOPs": tf, "achieved_GBps_compulsory": gb,
                       "frac": tf / PEAK_TFLOPS[a.dtype] if mf else gb / 8000.0}
        out["roofline"]["other_conv_kernels"] = oth
        m, h = by_bound["mfma"], by_bound["hbm"]
        out["roofline"]["conv_igemm_by_bound"] = {
            "mfma_bound": {"launches_per_step": m["launches"] // reps, "ms_per_step": m["ms"] / reps,
                           "achieved_TFLOPs": m["flops"] / (m["ms"] * 1e-3) / 1e12 if m["ms"] else None,
                           "frac": m["flops"] / (m["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS[a.dtype] if m["ms"] else None},
            "hbm_bound": {"launches_per_step": h["launches"] // reps, "ms_per_step": h["ms"] / reps,
                          "achieved_GBps": h["bytes"] / (h["ms"] * 1e-3) / 1e9 if h["ms"] else None, "peak_GBps": 8000.0,
                          "frac": h["bytes"] / (h["ms"] * 1e-3) / 8e12 if h["ms"] else None,
                          "note": "compulsory bytes (input + output + epilogue operands once) of the 1x1x1 GEMMs"},
            "note": "every convolution-class launch of the step (all three kernels, without the stem's data-gradient slot) by the roofline that bounds it"}
        if fused["launches"]:
            out["roofline"]["stem_delta_grad_kernel"] = {"ms": fused["ms"] / reps, "achieved": fused["flops"] / (fused["ms"] * 1e-3) / 1e12,
                                                         "unit": "TFLOP/s (algorithmic flops of the stem data-gradient it replaces)",
                                                         "includes": "the clip-mask pre-pass (0.04 ms), run inline in this serial profile; in the timed region it runs beside the stem forward",
                                                         "frac": fused["flops"] / (fused["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS[a.dtype]}
        out["kernel_ms_per_step"] = {k: v["ms"] / reps for k, v in per_kind.items()}
        if "pool" in per_kind and per_kind["pool"]["ms"] > 0:
            pk = per_kind["pool"]
            out["pool_GBps"] = pk["bytes"] / (pk["ms"] * 1e-3) / 1e9

    parity_failed = []
    if rank == 0 and world == 1:
        x_host = x.cpu()
        del eng, x
        torch.cuda.empty_cache()
        if not a.no_other_configs:
            out["other_configs"] = other_configs(local_rank)
        if not a.no_parity:
            out["parity"] = parity_check(local_rank)
            if not out["parity"]["ok"]:
                parity_failed.append("parity")
                print(f"[bench] PARITY CHECK FAILED: {out['parity']}", file=sys.stderr)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cb = cpu_baseline(W, x_host, local_rank)
            out["gpu_over_cpu"] = out["value"] / cb.get(f"bs{B}", cb)["value"]
            bad = [k for k in (cb, cb.get(f"bs{B}", {})) for dt in ("f32", "bf16") if "parity_same_clip" in k and not k["parity_same_clip"][dt]["ok"]]
            if bad:
                parity_failed.append("parity_same_clip")
                print(f"[bench] SAME-CLIP PARITY FAILED: {[b['parity_same_clip'] for b in bad]}", file=sys.stderr)

    if rank == 0:
        # a throughput line from a numerically broken build must not read as a result: the line says so and the process fails
        out["parity_ok"] = not parity_failed
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()

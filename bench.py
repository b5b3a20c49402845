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


def cpu_baseline(W, T, budget_s=30.0):
    """The CPU restatement (oracle/, torch-CPU fp32, all host cores) timed on the SAME iteration at bs=1.
    This is the only place bench.py touches oracle/: it is the baseline being reported, never the product path."""
    from oracle import attack_math as am
    from oracle import i3d_ref
    from flickering_adversarial_video_amd import i3d_spec
    # host cores this process may use (the 1-GPU boxes grant a 16-core share of a much larger host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("FLK_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    x = torch.from_numpy(i3d_spec.synthetic_clip_u8(1, T, seed=1234)).float() / 128 - 1
    d = torch.zeros(T, 1, 1, 3)
    m, v = torch.zeros_like(d), torch.zeros_like(d)
    label = None
    times = []
    t_start = time.time()
    it = 0
    while True:
        t0 = time.time()
        dv = d.clone().requires_grad_(True)
        lg = i3d_ref.i3d_logits(am.tf_apply(x, dv), Wt)
        if label is None:
            label = lg.argmax(-1)
        adv, _, _ = am.tf_improve_adversarial_loss(lg, label, 0.05, False, False)
        total, _ = am.tf_total_loss(adv, dv, 1.0, 0.5, 0.5, 0.5)
        (g,) = torch.autograd.grad(total, dv)
        d, m, v = am.tf_adam_step(d, g, m, v, it + 1)
        dt = time.time() - t0
        it += 1
        if it > 1:
            times.append(dt)          # first iteration is the warm-up
        if len(times) >= 3 or (time.time() - t_start > budget_s and len(times) >= 1):
            break
    sec = float(np.mean(times))
    return {"value": 1.0 / sec, "unit": "clip-iters/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed iterations (+1 warm-up) of the same I3D attack iteration at bs=1, T={T}, 224x224, torch-CPU fp32",
            "sec_per_iter": sec, "conv_gflops": conv_gflop_per_clip(T) / sec, "adv_loss_last": float(adv.detach())}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (BASELINE: bs=8)")
    ap.add_argument("--frames", type=int, default=64, help="frames per clip (BASELINE: 64; reference default 90)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="i3d", choices=["i3d", "r2plus1d_18", "r3d_18", "mc3_18"],
                    help="i3d = the headline config; the VideoResNet models are BASELINE config 3 (use --batch 1 --frames 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--autotune", action="store_true", help="time the candidate launch layouts of every convolution once before the warm-up "
                    "(measured: no gain over the built-in heuristics, which is why it is off by default)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"[bench] warning: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
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
        reps = 3
        for _ in range(reps):
            eng.step(x, labels, **hp)
            for r in eng.net.profile_read():
                k = per_kind.setdefault(r["kind"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["bytes"] += r["bytes"]; k["launches"] += 1
        eng.net.profile(False)
        cv = per_kind["conv"]
        ach = cv["flops"] / (cv["ms"] * 1e-3) / 1e12
        # HBM traffic per launch comes from the committed rocprofv3 --pmc passes of this same command (bench.py cannot
        # profile itself): profiles/*_pmc_hbm_traffic.json, produced by tools/pmc_summary.py
        traffic = None
        if B == 8 and T == 64 and a.dtype == "bf16":
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))[-1:]:
                traffic = json.load(open(f))["kernels"]["conv_igemm_kernel"]["bytes_per_launch"]
        out["roofline"] = {"kernel": "conv_igemm_kernel (implicit-GEMM conv3d fwd + dgrad, all layers)", "bound": "mfma",
                           "achieved": ach, "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[a.dtype],
                           "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, gfx950-corrected)", "launches_per_step": cv["launches"] // reps,
                           "avg_launch_ms": cv["ms"] / cv["launches"], "conv_ms_per_step": cv["ms"] / reps,
                           "algorithmic_gflop_per_step": cv["flops"] / reps / 1e9}
        out["kernel_ms_per_step"] = {k: v["ms"] / reps for k, v in per_kind.items()}
        if "pool" in per_kind and per_kind["pool"]["ms"] > 0:
            pk = per_kind["pool"]
            out["pool_GBps"] = pk["bytes"] / (pk["ms"] * 1e-3) / 1e9

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        del eng
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(W, T)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

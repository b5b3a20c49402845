#!/usr/bin/env python3
"""Per-layer timing of one VideoResNet attack iteration (HIP events around every launch of the plan): batched single-video attacks."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flickering_adversarial_video_amd import videoresnet_spec as vs
from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="r2plus1d_18"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
W = vs.synthetic_weights(a.arch, 42)
eng = FlickerVideoResNet(a.arch, W, batch_size=a.batch, sample_length=16, image_size=112, dtype="bf16", per_clip=a.batch > 1)
x = torch.from_numpy(vs.synthetic_clip(a.batch, 16, seed=1234)).cuda()
lab = eng.logits(x).argmax(-1).clone()
crit = Losses(beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True, logits=True)
for _ in range(2): eng.step(x, lab, crit)
eng.net.profile(True)
acc = {}
for _ in range(a.reps):
    eng.step(x, lab, crit)
    for i, r in enumerate(eng.net.profile_read()):
        k = (i, r["name"], r["pass"], r.get("kernel", ""))
        e = acc.setdefault(k, dict(ms=0.0, flops=r["flops"], bytes=r["bytes"]))
        e["ms"] += r["ms"] / a.reps
tot = sum(v["ms"] for v in acc.values())
print(f"{'op':44s} {'pass':4s} {'ms':>8s} {'%':>6s} {'TFLOP/s':>9s} {'GB/s':>8s}  kernel")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k[1][:44]:44s} {k[2]:4s} {v['ms']:8.3f} {100*v['ms']/tot:6.1f} {v['flops']/v['ms']/1e9 if v['flops'] else 0:9.1f} {v['bytes']/v['ms']/1e6 if v['bytes'] else 0:8.0f}  {k[3].replace('_kernel','')}")
print("total ms", tot)

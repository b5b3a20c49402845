#!/usr/bin/env python3
"""Per-layer timing of one I3D attack iteration (HIP events around every launch of the plan)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flickering_adversarial_video_amd import i3d_spec
from flickering_adversarial_video_amd.i3d_engine import FlickerI3D

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--json", default=None)
a = ap.parse_args()
W = i3d_spec.synthetic_i3d_weights(42)
eng = FlickerI3D(W, batch_size=a.batch, frames=a.frames, dtype=a.dtype)
x = torch.from_numpy(i3d_spec.synthetic_clip_u8(a.batch, a.frames)).cuda()
labels = eng.logits(x, adv_flag=0.0).argmax(-1).clone()
for _ in range(2): eng.step(x, labels)
eng.net.profile(True)
acc = {}
for _ in range(a.reps):
    eng.step(x, labels)
    for i, r in enumerate(eng.net.profile_read()):
        k = (i, r["name"], r["pass"], r["kind"], r.get("kernel", ""))
        e = acc.setdefault(k, dict(ms=0.0, flops=r["flops"], bytes=r["bytes"]))
        e["ms"] += r["ms"] / a.reps
rows = [dict(name=k[1], **{"pass": k[2]}, kind=k[3], kernel=k[4], **v) for k, v in acc.items()]
tot = sum(r["ms"] for r in rows)
# roofline per launch: time the MFMA peak (bf16 2500 / fp32 ~157 TFLOP/s) or a streaming kernel's HBM rate (~4.5 TB/s of the 8 TB/s
# peak) would need for the launch's algorithmic flops / compulsory bytes, whichever is larger; frac = that bound / measured time
PEAK_TF = 2500.0 if a.dtype == "bf16" else 157.0
print(f"{'op':58s} {'pass':4s} {'ms':>8s} {'%':>6s} {'TFLOP/s':>9s} {'GB/s':>8s} {'bound':>5s} {'frac':>5s}  kernel")
for r in sorted(rows, key=lambda r: -r["ms"]):
    tf = r["flops"] / r["ms"] / 1e9 if r["flops"] else 0
    gb = r["bytes"] / r["ms"] / 1e6 if r["bytes"] else 0
    t_mfma, t_hbm = r["flops"] / (PEAK_TF * 1e9), r["bytes"] / 8e9          # ms at the peaks
    bound, frac = ("mfma", t_mfma / r["ms"]) if t_mfma >= t_hbm else ("hbm", t_hbm / r["ms"])
    print(f"{r['name'][:58]:58s} {r['pass']:4s} {r['ms']:8.3f} {100*r['ms']/tot:6.1f} {tf:9.1f} {gb:8.0f} {bound:>5s} {frac:5.2f}  {r['kernel'].replace('_kernel', '')}")
print("total ms", tot)
by = {}
for r in rows:
    t_mfma, t_hbm = r["flops"] / (PEAK_TF * 1e9), r["bytes"] / 8e9
    k = "mfma-bound launches" if t_mfma >= t_hbm else "hbm-bound launches"
    e = by.setdefault(k, [0.0, 0.0]); e[0] += r["ms"]; e[1] += max(t_mfma, t_hbm)
for k, (ms, floor) in by.items():
    print(f"{k}: {ms:.3f} ms measured, {floor:.3f} ms at the peak ({PEAK_TF:.0f} TFLOP/s / 8 TB/s) -> {floor / ms:.2f}")
if a.json: json.dump(rows, open(a.json, "w"), indent=1)

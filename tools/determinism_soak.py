#!/usr/bin/env python3
"""Two engines with the same weights, clips and hyper-parameters run the same attack side by side: after every block of iterations their
perturbations, Adam moments and logits must be BITWISE equal (the plan's reductions are fixed-order, the pool backward adds integers) --
a race inside a kernel (a halo image overwritten under a reader, a ring slot refilled early) shows up as a difference within a few
hundred launches.  usage: determinism_soak.py [iterations] [batch] [frames]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import i3d_spec
from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
W = i3d_spec.synthetic_i3d_weights(42)
engs = [FlickerI3D(W, batch_size=B, frames=T, dtype="bf16") for _ in range(2)]
x = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T, seed=1234)).cuda()
labels = engs[0].logits(x, adv_flag=0.0).argmax(-1).clone()
hp = dict(lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
t0 = time.time()
for it in range(1, n + 1):
    for e in engs:
        e.step(x, labels, **hp)
    if it % 25 == 0 or it == n:
        torch.cuda.synchronize()
        same = all(torch.equal(getattr(engs[0], k), getattr(engs[1], k)) for k in ("perturbation", "adam_m", "adam_v"))
        la, lb = engs[0].logits(x), engs[1].logits(x)
        same = same and torch.equal(la, lb)
        print(f"iter {it}: {'bitwise equal' if same else 'DIFFERENT'} | |delta|max {float(engs[0].perturbation.abs().max()):.5f} | {1e3 * (time.time() - t0) / (2 * it):.2f} ms/iter", flush=True)
        if not same:
            sys.exit(1)
print("determinism soak ok")

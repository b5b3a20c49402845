import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import i3d_spec
from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
W = i3d_spec.synthetic_i3d_weights(42)
eng = FlickerI3D(W, batch_size=8, frames=64, dtype="bf16")
x = torch.from_numpy(i3d_spec.synthetic_clip_u8(8, 64, seed=1234)).cuda()
labels = eng.logits(x, adv_flag=0.0).argmax(-1).clone()
hp = dict(lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
hist = []
t0 = time.time()
for it in range(1, 601):
    r = eng.step(x, labels, **hp)
    if it % 100 == 0:
        h = r.host()
        torch.cuda.synchronize()
        print(f"iter {it}: adv {h['adv_loss']:.4f} total {h['total_loss']:.4f} thickness {h['thickness_relative']:.3f}% roughness {h['roughness_relative']:.3f}% "
              f"prob_to_min {h['prob_to_min']:.4f} fooled {bool(h['is_adversarial'])} | {1e3 * (time.time() - t0) / it:.2f} ms/iter", flush=True)
        assert all(map(lambda v: torch.isfinite(torch.as_tensor(v)).all(), (h['adv_loss'], h['total_loss'])))
print("delta max", float(eng.perturbation.abs().max()))

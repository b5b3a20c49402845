#!/usr/bin/env python3
"""Attack iterations per second when every step takes a NEW batch from TFRecord files (the universal attack's real loop,
i3d_adversarial_main_universal.py:45-203) through prefetch.DeviceBatches, beside the resident-batch rate bench.py reports.
Writes a synthetic uint8 TFRecord file first (320 clips of 64 x 224 x 224 x 3 = 3.1 GB, page-cache resident afterwards).  The rate is
taken over steps 8.. of an epoch: starting the reader thread and filling the ring costs ~10 ms once per epoch."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flickering_adversarial_video_amd import i3d_spec, prefetch, tfrecord_io as tio
from flickering_adversarial_video_amd.i3d_engine import FlickerI3D

B, T, NCLIPS = 8, 64, 320
d = tempfile.mkdtemp()
path = os.path.join(d, "synthetic.tfrecords")
rng = np.random.default_rng(0)
base = [rng.integers(0, 256, (T, 224, 224, 3), dtype=np.uint8) for _ in range(8)]
tio.write_records(path, (tio.make_example(base[i % 8], i % 400) for i in range(NCLIPS)), with_payload_crc=False)
print(f"wrote {os.path.getsize(path) / 1e6:.0f} MB", flush=True)
eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=B, frames=T, dtype="bf16")
hp = dict(lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05)
x0 = torch.from_numpy(np.stack(base)).cuda()
y0 = eng.logits(x0, adv_flag=0.0).argmax(-1).clone()
for _ in range(5):
    eng.step(x0, y0, **hp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    eng.step(x0, y0, **hp)
torch.cuda.synchronize()
res = (time.perf_counter() - t0) / 40
print(f"resident batch: {res * 1e3:.2f} ms per step = {B / res:.0f} clip-iterations/s", flush=True)
# reader alone (parse + copy into the pinned ring + H2D), no attack
db = prefetch.DeviceBatches([path], B, T)
for epoch in range(2):
    t0 = time.perf_counter(); n = 0
    for x, y in db:
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"loader alone, pass {epoch}: {n} batches in {dt:.3f} s = {n * B / dt:.0f} clips/s = {n * B * T * 150528 / dt / 1e9:.2f} GB/s", flush=True)
# the real loop: new batch every step
for epoch in range(2):
    n = 0
    for x, y in db:
        if n == 8:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.step(x, torch.from_numpy(y).cuda(), **hp)
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"attack with a new batch per step, pass {epoch}, steps 8..{n}: {dt / (n - 8) * 1e3:.2f} ms per step = {(n - 8) * B / dt:.0f} clip-iterations/s", flush=True)
os.remove(path)
if os.environ.get("FLK_PIPE_DIAG"):
    tio.write_records(path, (tio.make_example(base[i % 8], i % 400) for i in range(NCLIPS)), with_payload_crc=False)
    for tag, dbx in (("reader + H2D running, attack on the resident batch", prefetch.DeviceBatches([path], B, T)),
                     ("reader only (no H2D), attack on the resident batch", prefetch.DeviceBatches([path], B, T, device="cpu"))):
        for epoch in range(2):
            t0 = time.perf_counter(); n = 0
            for x, y in dbx:
                eng.step(x0, y0, **hp)
                n += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{tag}: {dt / n * 1e3:.2f} ms per step", flush=True)
    os.remove(path)

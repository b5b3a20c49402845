#!/usr/bin/env python3
"""Time flk_stem_fwd_u8 alone (HIP events; bs clips of 64x224x224).  FLK_SF_ABLATE (timing-only -DSF_ABLATE builds) selects the
ablation, a -DFLK_STEM_NI8 build the 128-positions-per-wave instance; the output of an ablated run is garbage and goes nowhere."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flickering_adversarial_video_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--frames", type=int, default=64); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
rng = np.random.default_rng(0)
xu = torch.from_numpy(rng.integers(0, 256, (a.batch, a.frames, 224, 224, 3), dtype=np.uint8)).cuda()
d = torch.from_numpy(rng.uniform(-0.1, 0.1, (a.frames, 3)).astype(np.float32)).cuda()
w7 = (rng.standard_normal((7, 7, 7, 3, 64)) * (2.0 / 1029) ** 0.5).astype(np.float32)
sc, bi = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
args = ops.make_apply_args(xu, d, fold_t=ops.I3D_FOLD, center=True)
tab = ops.stem_delta_bias_table(args, w7, np.ones(64, np.float32))
w = ops.StemFwdU8Weights(w7)
out = torch.empty((a.batch, a.frames // 2, 112, 112, 64), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.stem_fwd_u8(args, w, sc, bi, tab, out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    ops.stem_fwd_u8(args, w, sc, bi, tab, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
gf = 2.0 * a.batch * (a.frames // 2) * 112 * 112 * 1029 * 64 / 1e9
print(f"stem_fwd_u8 bs {a.batch} T {a.frames} NI {os.environ.get('FLK_STEM_NI', 'default')} ablate {os.environ.get('FLK_SF_ABLATE', '0')}: "
      f"{ms:.4f} ms, {gf / ms:.1f} TFLOP/s algorithmic")

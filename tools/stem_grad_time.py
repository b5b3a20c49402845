#!/usr/bin/env python3
"""Isolated timing of the fused stem delta-gradient kernel (csrc/stem_grad.hip) at the benchmark shape, with the debug knobs
FLK_SG_DBG / FLK_SG_NCHUNK of the library (set them in the environment): A/B work on the kernel's phases.  FLK_SG_DBG needs a timing
build (FLK_HIPCC_EXTRA=-DFLK_ABLATE python -m flickering_adversarial_video_amd.build --force): the product build has no such switch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flickering_adversarial_video_amd import ops
B, T = int(os.environ.get("B", 8)), int(os.environ.get("T", 64))
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 256, (B, T, 224, 224, 3), dtype=np.uint8)).cuda()
G = (torch.randn(B, T // 2, 112, 112, 64, device="cuda") * 0.01).to(torch.bfloat16)
delta = torch.from_numpy(rng.uniform(-0.1, 0.1, (T, 3)).astype(np.float32)).cuda()
args = ops.make_apply_args(x, delta, fold_t=ops.I3D_FOLD)
w = ops.StemDeltaGradWeights((rng.standard_normal((7, 7, 7, 3, 64)) * 0.05).astype(np.float32), np.ones(64, np.float32))
gd = torch.empty(T, 3, device="cuda"); sc = torch.empty(ops.load().flk_stem_delta_grad_scratch_bytes(B, T, 224) // 4 + 1, device="cuda")
for _ in range(3): ops.stem_delta_grad(args, G, w, gd, sc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): ops.stem_delta_grad(args, G, w, gd, sc)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"B={B} T={T} FLK_SG_DBG={os.environ.get('FLK_SG_DBG', '0')} FLK_SG_NCHUNK={os.environ.get('FLK_SG_NCHUNK', '-')}: {ms:.4f} ms  "
      f"({B * 26.43e9 * 2 * T / 64 / ms / 1e9:.0f} TFLOP/s algorithmic)")

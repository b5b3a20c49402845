#!/usr/bin/env python3
"""Isolated timing of the stride-1 3x3x3 max-pool forward (csrc/pool.hip: the W-run kernel of the Inception Branch_3 pools) at the geometry
of an I3D block of the benchmark (default Mixed_3c: 8 x 32 x 28 x 28 positions, 256 channels), with the debug knob FLK_PF_DBG of the
library (set it in the environment; timing builds only: FLK_HIPCC_EXTRA=-DFLK_ABLATE python -m flickering_adversarial_video_amd.build
--force -- the product build has no such switch): A/B work on the kernel's phases."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flickering_adversarial_video_amd import ops
B, T, H, W = [int(v) for v in os.environ.get("GEO", "8,32,28,28").split(",")]
C = int(os.environ.get("C", 256))
torch.manual_seed(0)
x = torch.relu(torch.randn(B, T, H, W, C, device="cuda")).to(torch.bfloat16)
for _ in range(3): ops.maxpool3d(x, (3, 3, 3), (1, 1, 1))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): ops.maxpool3d(x, (3, 3, 3), (1, 1, 1))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
npos = B * T * H * W
print(f"{B}x{T}x{H}x{W} C={C} FLK_PF_DBG={os.environ.get('FLK_PF_DBG', '0')}: {ms * 1e3:.1f} us  ({npos * C * 5 / ms / 1e6:.0f} GB/s: in + out + index bytes)")

#!/usr/bin/env python3
"""Isolated timing of the fused Branch_3 backward (csrc/pool.hip: flk_maxpool3d_bwd_gemm) at the geometry of an I3D block of the benchmark
(default Mixed_3c: 8 x 32 x 28 x 28 positions, 256 pooled channels, K = 64), with the debug knobs FLK_PG_DBG / FLK_POOL_GEMM_REG of the
library (set them in the environment): A/B work on the kernel's phases.  FLK_PG_DBG needs a timing build of the library
(FLK_HIPCC_EXTRA=-DFLK_ABLATE python -m flickering_adversarial_video_amd.build --force): the product build has no such switch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flickering_adversarial_video_amd import ops
B, T, H, W = [int(v) for v in os.environ.get("GEO", "8,32,28,28").split(",")]
C, K = int(os.environ.get("C", 256)), int(os.environ.get("K", 64))
torch.manual_seed(0)
x = torch.relu(torch.randn(B, T, H, W, C, device="cuda")).to(torch.bfloat16)
out, idx, ctx = ops.maxpool3d(x, (3, 3, 3), (1, 1, 1))
g = (torch.randn(B, T, H, W, K, device="cuda") * 0.01).to(torch.bfloat16)
wp = ops.PoolGemmWeights((np.random.default_rng(0).standard_normal((K, C)) * 0.1).astype(np.float32))
for _ in range(3): gin = ops.maxpool3d_bwd_gemm(ctx, g, wp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): ops.maxpool3d_bwd_gemm(ctx, g, wp)          # (allocates its output from torch's cache: no device call)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
npos = B * T * H * W
print(f"{B}x{T}x{H}x{W} C={C} K={K} FLK_PG_DBG={os.environ.get('FLK_PG_DBG', '0')} FLK_POOL_GEMM_REG={os.environ.get('FLK_POOL_GEMM_REG', '1')}: "
      f"{ms * 1e3:.1f} us  ({npos * (2 * K + C + 2 * C) / ms / 1e6:.0f} GB/s of compulsory bytes: g + idx + gin)")

#!/usr/bin/env python3
"""In-kernel clock and cycles per K step of conv_pc_kernel (diagnostic -DPC_STAMP build of the library: tools/build_variant.py stamp --src
conv_pc.hip -DPC_STAMP; run with FLK_LIB_PATH=.../libflk_stamp.so).  usage: pc_stamps.py cin cout B T H W [T = data-gradient]"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import ops, _lib
cin, cout, B, T, H, W = (int(v) for v in sys.argv[1:7]); tr = len(sys.argv) > 7
x = torch.randn(B, T, H, W, cout if tr else cin, device="cuda").to(torch.bfloat16)
w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)
pw = ops.ConvWeights(w, torch.bfloat16, 4, transpose=tr)
co = cin if tr else cout
epi = dict(mask=torch.randn(B, T, H, W, co, device="cuda").to(torch.bfloat16)) if tr else dict(scale=torch.rand(co, device="cuda") + 0.5, bias=torch.randn(co, device="cuda") * 0.1, relu=True)
pad = (1, 1, 1)
for _ in range(200): ops.conv3d_pc([(x, pw, dict(pad=pad, out_grid=(T, H, W), **epi))])      # (long enough for the clock to settle)
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (8 * 256))()
assert lib.flk_pc_stamps_read(buf, 256) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
a = a[a[:, 2] > 0]
clk = a[:, 0] / a[:, 1] * 0.1      # GHz
cps = a[:, 0] / a[:, 2]
print(f"{len(a)} workgroups: in-kernel clock median {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f}); cycles per K step median {np.median(cps):.0f} "
      f"(min {cps.min():.0f}, max {cps.max():.0f}); NI {int(a[0, 3])}: {int(a[0, 3]) * 64} cycles of MFMA issue per step; ns per step {np.median(a[:, 1] * 10 / a[:, 2]):.0f}")
print(f"per item (cycles, medians): set-up {np.median(a[:, 4]):.0f}, K loop {np.median(a[:, 0]):.0f}, epilogue {np.median(a[:, 5]):.0f}, item end -> next item's start {np.median(a[:, 6] - a[:, 7]):.0f}")

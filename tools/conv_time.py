"""Time ONE convolution geometry in isolation (20 launches x 5 repetitions after a 20-launch warm-up; min / median).
A/B different builds or env knobs by alternating processes -- single measurements inside a long run drift with the clock state.

usage: conv_time.py k cin cout nf B T H W [T]      (k x k x k taps, stride 1 SAME; a trailing argument = data-gradient; KT=1 in the environment: 1 x k x k)
PC=1 in the environment: the launch goes to flk_conv3d_pc (the producer / consumer kernel; 3x3x3, nf 4) instead of flk_conv3d."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import ops
# usage: conv_time.py kt cin cout nf B T H W [transpose]
kt, cin, cout, nf, B, T, H, W = (int(v) for v in sys.argv[1:9])
tr = len(sys.argv) > 9
torch.manual_seed(0)
x = torch.randn(B, T, H, W, cout if tr else cin, device="cuda").to(torch.bfloat16)
kt0 = int(os.environ.get("KT", kt))      # KT=1: (1, k, k) taps -- the spatial half of a (2+1)D unit
w = (np.random.default_rng(0).standard_normal((kt0, kt, kt, cin, cout)) * 0.05).astype(np.float32)
pw = ops.ConvWeights(w, torch.bfloat16, nf, transpose=tr)
out = torch.empty(B, T, H, W, cin if tr else cout, device="cuda", dtype=torch.bfloat16)
pad = (kt0 - 1 - (kt0 - 1) // 2, kt - 1 - (kt - 1) // 2, kt - 1 - (kt - 1) // 2) if tr else None
splitk = bool(int(os.environ.get("SPLITK", "0")))      # give the launch a split-K workspace (FLK_CONV_KSPLIT forces the slice count)
if splitk:
    ref = ops.conv3d(x, pw, pad=pad, out_grid=(T, H, W)).float()
    got = ops.conv3d(x, pw, pad=pad, out_grid=(T, H, W), splitk=True).float()
    print("split-K vs one slice: max |diff| %.3e (max |ref| %.3e)" % ((got - ref).abs().max().item(), ref.abs().max().item()))
import ctypes as C
_keep = []
pc = bool(int(os.environ.get("PC", "0")))
# EPI=1: the epilogue operands the plan uses -- forward: batch-norm scale, bias, ReLU; data-gradient: the ReLU mask of the layer's input
epi = {}
if int(os.environ.get("EPI", "0")):
    co = cin if tr else cout
    if tr: epi = dict(mask=torch.randn(B, T, H, W, co, device="cuda").to(torch.bfloat16))
    else: epi = dict(scale=torch.rand(co, device="cuda") + 0.5, bias=torch.randn(co, device="cuda") * 0.1, relu=True)
if pc:
    ref = ops.conv3d(x, pw, pad=pad, out_grid=(T, H, W), **epi)
    got = ops.conv3d_pc([(x, pw, dict(pad=pad, out_grid=(T, H, W), **epi))])[0]
    print("producer / consumer kernel vs flk_conv3d: bitwise equal", bool(torch.equal(ref, got)))
def f():
    if pc: ops.conv3d_pc([(x, pw, dict(pad=pad, out_grid=(T, H, W), out=out, **epi))])
    else: ops.conv3d(x, pw, pad=pad, out_grid=(T, H, W), out=out, splitk=splitk, **epi)
for _ in range(20): f()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(sys.argv[1:], "ms min %.4f med %.4f" % (min(ts), sorted(ts)[2]), flush=True)

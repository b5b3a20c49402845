import os, re, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import ops
rows = []
for l in open("profiles/r01s_conv_launch_configs.txt"):
    m = re.search(r"conv (\d)x(\d)x(\d) s(\d)(\d)(\d) cin (\d+) cout (\d+) out (\d+)x(\d+)x(\d+)x(\d+) \| nf (\d+) wn (\d+)", l)
    if m:
        rows.append(tuple(int(v) for v in m.groups()))
rows = sorted(set(rows))
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
tot_cur = tot_best = 0.0
for (kt, kh, kw, st, sh, sw, cin, cout, B, T, H, W, nf_cur, wn_cur) in rows:
    if (kt, kh, kw) == (4, 4, 4): continue
    x = torch.randn(B, T, H, W, cin, device="cuda").to(torch.bfloat16)
    w = (np.random.default_rng(0).standard_normal((kt, kh, kw, cin, cout)) * 0.05).astype(np.float32)
    out = torch.empty(B, T, H, W, cout, device="cuda", dtype=torch.bfloat16)
    res = {}
    for nf in (2, 4, 8):
        if nf * 16 > cout * 2 and nf != 2 and cout <= 16 * nf // 2: continue
        pw = ops.ConvWeights(w, torch.bfloat16, nf)
        try:
            res[nf] = t(lambda: ops.conv3d(x, pw, out=out, relu=True))
        except Exception as e:
            res[nf] = float("nan")
        del pw
    cur = res.get(nf_cur, float("nan")); best_nf = min(res, key=lambda k: res[k] if res[k] == res[k] else 1e9)
    tot_cur += cur; tot_best += res[best_nf]
    flag = "  <== " if res[best_nf] < 0.93 * cur else ""
    print(f"{kt}x{kh}x{kw} cin {cin:4d} cout {cout:4d} {B}x{T}x{H}x{W}: " + " ".join(f"nf{k}={v*1e3:6.1f}us" for k, v in res.items()) + f" | cur nf{nf_cur} best nf{best_nf}{flag}", flush=True)
print("sum current", tot_cur, "sum best", tot_best)

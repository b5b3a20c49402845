"""Kernel-level parity (GPU): every HIP kernel, called through the C ABI, against the CPU oracle
(torch-CPU fp32 conv3d / max_pool3d autograd and oracle/attack_math.py) on the same seeded inputs.

Tolerances: fp32 mode (exact-fp32 MFMA) 1e-4 relative -- accumulation order only; the north-star bar is
1e-3.  bf16 mode: inputs/weights are rounded to bf16 on both sides, so the difference is fp32
accumulation order plus one bf16 rounding of the output: 1e-2 relative to the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * scale)


def q(x, dtype):
    """round to the storage dtype (bf16 mode) and back to fp32"""
    return x.to(dtype).float()


def tol(dtype, ref):
    s = float(ref.abs().max()) + 1e-12
    return (1e-4, 1e-5 * s) if dtype == torch.float32 else (1e-2, 1e-2 * s)


def cl(x):   # NCDHW -> NDHWC
    return x.permute(0, 2, 3, 4, 1).contiguous()


def cf(x):   # NDHWC -> NCDHW
    return x.permute(0, 4, 1, 2, 3).contiguous()


def ref_conv(x_cl, w_dhwio, stride, pad_before, out_grid):
    """oracle: torch-CPU conv3d with explicit (asymmetric) zero padding; x NDHWC fp32"""
    x = cf(x_cl)
    k = w_dhwio.shape[:3]
    pads = []
    for i in (2, 1, 0):
        n = x.shape[2 + i]
        after = (out_grid[i] - 1) * stride[i] + k[i] - pad_before[i] - n
        pads += [pad_before[i], max(after, 0)]
    y = F.conv3d(F.pad(x, pads), w_dhwio.permute(4, 3, 0, 1, 2).contiguous(), stride=stride)
    return cl(y)[:, :out_grid[0], :out_grid[1], :out_grid[2]]


CONV_CASES = [
    # name, B,T,H,W, cin,cout, k, stride, nf
    ("3x3x3_64_192", 2, 4, 14, 14, 64, 192, (3, 3, 3), (1, 1, 1), 8),
    ("3x3x3_16_32", 1, 5, 9, 11, 16, 32, (3, 3, 3), (1, 1, 1), 2),
    ("3x3x3_24_64_odd", 1, 3, 7, 7, 24, 64, (3, 3, 3), (1, 1, 1), 4),
    ("3x3x3_48_128", 1, 2, 7, 7, 48, 128, (3, 3, 3), (1, 1, 1), 8),
    ("1x1x1_192_96", 2, 2, 28, 28, 192, 96, (1, 1, 1), (1, 1, 1), 8),
    ("1x1x1_528_256", 1, 2, 14, 14, 528, 256, (1, 1, 1), (1, 1, 1), 8),
    ("4x4x4_32_64_stem", 1, 4, 16, 16, 32, 64, (4, 4, 4), (1, 1, 1), 4),
    ("1x3x3_s2", 1, 4, 15, 16, 64, 144, (1, 3, 3), (1, 2, 2), 8),
    ("3x1x1_s2", 1, 9, 8, 8, 144, 64, (3, 1, 1), (2, 1, 1), 4),
    ("1x1x1_s2_ds", 1, 4, 14, 14, 64, 128, (1, 1, 1), (2, 2, 2), 8),
    # 96-channel tiles (nf = 6, bf16 ring kernels only): two full tiles; one tile + a partial one; a 1x1x1 with >= 4 slabs (mode 3) and one
    # with 2 (mode 0)
    ("3x3x3_64_192_nf6", 2, 4, 14, 14, 64, 192, (3, 3, 3), (1, 1, 1), 6),
    ("3x3x3_32_136_nf6", 1, 3, 7, 9, 32, 136, (3, 3, 3), (1, 1, 1), 6),
    ("1x1x1_192_176_nf6", 2, 2, 28, 28, 192, 176, (1, 1, 1), (1, 1, 1), 6),
    ("1x1x1_64_96_nf6", 1, 2, 14, 14, 64, 96, (1, 1, 1), (1, 1, 1), 6),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward(ops, case, dtype):
    _, B, T, H, W, cin, cout, k, s, nf = case
    if nf == 6 and dtype != torch.bfloat16:
        pytest.skip("96-channel tiles exist in bf16 only")
    x = q(rnd((B, T, H, W, cin), 1), dtype)
    w = q(rnd((*k, cin, cout), 2, (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5), dtype)
    og, pad = zip(*(ops.same_pad(n, kk, ss) for n, kk, ss in zip((T, H, W), k, s)))
    ref = ref_conv(x, w, s, pad, og)
    pw = ops.ConvWeights(w.numpy(), dtype, nf)
    out = ops.conv3d(x.to(dtype).cuda(), pw, stride=s)
    assert out.shape == ref.shape
    r, a = tol(dtype, ref)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=r, atol=a)


SPLITK_CASES = [
    # name, B,T,H,W, cin,cout, k, nf      (few output tiles, long K loop: the Mixed_5* regime and the late VideoResNet layers)
    ("3x3x3_192_384", 2, 8, 7, 7, 192, 384, (3, 3, 3), 8),
    ("3x3x3_160_96_odd_slabs", 1, 4, 7, 7, 160, 96, (3, 3, 3), 4),
    ("1x3x3_256_256", 1, 4, 14, 14, 256, 256, (1, 3, 3), 8),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_conv_splitk(ops, case, dtype):
    """deterministic split-K (flk_conv_args.splitk_ws): the input-channel slabs are divided over blockIdx.y, the slices' fp32 partial
    sums are added in slice order by a second launch that also runs the epilogue (scale, bias, add, relu, mask).  Against the
    torch-CPU oracle at the usual tolerances; the same call twice gives the same bits."""
    import ctypes as C
    from flickering_adversarial_video_amd._lib import load
    _, B, T, H, W, cin, cout, k, nf = case
    x = q(rnd((B, T, H, W, cin), 1), dtype)
    w = q(rnd((*k, cin, cout), 2, (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5), dtype)
    scale, bias = rnd((cout,), 3).abs() + 0.5, rnd((cout,), 4) * 0.1
    add = q(rnd((B, T, H, W, cout), 5), dtype)
    mask = q(rnd((B, T, H, W, cout), 6), dtype)
    og, pad = zip(*(ops.same_pad(n, kk, 1) for n, kk in zip((T, H, W), k)))
    ref = torch.relu(ref_conv(x, w, (1, 1, 1), pad, og) * scale + bias + add) * (mask > 0)
    pw = ops.ConvWeights(w.numpy(), dtype, nf)
    kw = dict(scale=scale.cuda(), bias=bias.cuda(), add=add.to(dtype).cuda(), mask=mask.to(dtype).cuda(), relu=True)
    xd = x.to(dtype).cuda()
    out = ops.conv3d(xd, pw, splitk=True, **kw)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=r, atol=a)
    assert torch.equal(out, ops.conv3d(xd, pw, splitk=True, **kw))
    # the plan really splits this geometry (a workspace size of 0 would mean the launch above ran in one slice)
    args = ops.ConvArgs()
    args.B, args.To, args.Ho, args.Wo, args.OT, args.OH, args.OW = B, T, H, W, T, H, W
    args.kt, args.kh, args.kw, args.st, args.sh, args.sw, args.ost, args.osh, args.osw = *k, 1, 1, 1, 1, 1, 1
    args.cin, args.cout = cin, cout
    assert load().flk_conv_splitk_bytes(C.byref(args), pw.handle) > 0
    one = ops.conv3d(xd, pw, **kw)                       # one slice: equal up to the summation order
    torch.testing.assert_close(out.float(), one.float(), rtol=r, atol=a)


def fold_stem(w7):
    """[7,7,7,3,cout] -> [4,4,4,32,cout]: tap 2*j + q of an axis goes to folded tap j, parity q; channel (qt*2+qh)*8 + qw*3 + c"""
    cout = w7.shape[4]
    wf = torch.zeros(4, 4, 4, 32, cout)
    for kt in range(7):
        for kh in range(7):
            for kw in range(7):
                ch = ((kt & 1) * 2 + (kh & 1)) * 8 + (kw & 1) * 3
                wf[kt >> 1, kh >> 1, kw >> 1, ch:ch + 3] = w7[kt, kh, kw]
    return wf


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("nf", [4, 8])
def test_conv_folded_stem(ops, dtype, nf):
    """Conv3d_1a_7x7 (7x7x7 / 2, SAME) as the 4x4x4 convolution over the fold_t = 3 clip with K steps assembled from the
    non-zero chunks (mode 4, bf16) -- against the strided 7x7x7 oracle convolution on the unfolded clip."""
    B, T, H, W, cout = 1, 8, 32, 48, 64
    x = q(rnd((B, T, H, W, 3), 3), dtype)
    w7 = q(rnd((7, 7, 7, 3, cout), 4, (2.0 / 1029) ** 0.5), dtype)
    og, pad = zip(*(ops.same_pad(n, 7, 2) for n in (T, H, W)))
    ref = ref_conv(x, w7, (2, 2, 2), pad, og)
    xs = x.reshape(B, T // 2, 2, H // 2, 2, W // 2, 2, 3).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B, T // 2, H // 2, W // 2, 4, 6)
    xs = torch.cat([xs, torch.zeros(*xs.shape[:5], 2)], -1).reshape(B, T // 2, H // 2, W // 2, 32)
    pw = ops.ConvWeights.s2d_stem(fold_stem(w7).numpy(), dtype, nf)
    out = ops.conv3d(xs.to(dtype).cuda(), pw, pad=(1, 1, 1), out_grid=og)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=r, atol=a)
    # the generic dense packing of the same folded tensor gives the same result
    pg = ops.ConvWeights(fold_stem(w7).numpy(), dtype, nf)
    out_g = ops.conv3d(xs.to(dtype).cuda(), pg, pad=(1, 1, 1), out_grid=og)
    torch.testing.assert_close(out.float().cpu(), out_g.float().cpu(), rtol=r, atol=a)
    # a folded tensor with a non-zero weight in a structurally zero chunk is refused (bf16 packs K steps from the others)
    bad = fold_stem(w7).clone()
    bad[3, 0, 0, 16] = 1.0
    if dtype == torch.bfloat16:
        with pytest.raises(RuntimeError, match="must be zero"):
            ops.ConvWeights.s2d_stem(bad.numpy(), dtype, nf)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_epilogue_slices(ops, dtype):
    """channel slices of wider buffers (concat elimination) + scale/bias/add/relu/mask epilogue"""
    B, T, H, W, cin, cout = 2, 3, 10, 12, 32, 48
    xfull = q(rnd((B, T, H, W, 80), 3), dtype)                 # conv reads channels [16, 48)
    w = q(rnd((3, 3, 3, cin, cout), 4, 0.05), dtype)
    scale, bias = rnd((cout,), 5).abs() + 0.5, rnd((cout,), 6)
    add = q(rnd((B, T, H, W, 64), 7), dtype)                   # read at coff 8
    mask = q(rnd((B, T, H, W, 56), 8), dtype)                  # read at coff 8
    ref = ref_conv(xfull[..., 16:48].contiguous(), w, (1, 1, 1), (1, 1, 1), (T, H, W))
    ref = ref * scale + bias + add[..., 8:56]
    ref = torch.relu(ref)
    ref = torch.where(mask[..., 8:56] > 0, ref, torch.zeros_like(ref))
    out = torch.full((B, T, H, W, 72), 7.0, dtype=dtype).cuda()  # write at coff 16; the rest must stay untouched
    pw = ops.ConvWeights(w.numpy(), dtype, 4)
    ops.conv3d(xfull.to(dtype).cuda(), pw, in_coff=16, cin=cin, out=out, out_coff=16, scale=scale.cuda(), bias=bias.cuda(),
               add=add.to(dtype).cuda(), add_coff=8, mask=mask.to(dtype).cuda(), mask_coff=8, relu=True)
    o = out.float().cpu()
    r, a = tol(dtype, ref)
    torch.testing.assert_close(o[..., 16:64], ref, rtol=r, atol=a * 4)
    assert (o[..., :16] == 7).all() and (o[..., 64:] == 7).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("k,cin,cout,nf", [((3, 3, 3), 64, 96, 4), ((1, 1, 1), 256, 64, 8), ((4, 4, 4), 32, 64, 2),
                                           ((3, 3, 3), 16, 48, 2), ((3, 3, 3), 96, 128, 6)], ids=["3x3x3", "1x1x1", "4x4x4stem", "3x3x3_16", "3x3x3_nf6"])
def test_conv_data_gradient(ops, dtype, k, cin, cout, nf):
    """transposed operator == torch autograd's conv3d backward-input (stride 1), BN scale folded"""
    if nf == 6 and dtype != torch.bfloat16:
        pytest.skip("96-channel tiles exist in bf16 only")
    B, T, H, W = 1, 4, 9, 10
    pad = tuple((kk - 1) // 2 for kk in k)
    w = q(rnd((*k, cin, cout), 11, 0.05), dtype)
    a_scale = rnd((cout,), 12).abs() + 0.5
    g = q(rnd((B, T, H, W, cout), 13), dtype)
    x = torch.zeros((B, T, H, W, cin), requires_grad=True)
    y = ref_conv(x, w, (1, 1, 1), pad, (T, H, W)) * a_scale
    (gx_ref,) = torch.autograd.grad(y, x, g)
    pw = ops.ConvWeights(w.numpy(), dtype, nf, row_scale=a_scale.numpy(), transpose=True)
    dpad = tuple(kk - 1 - p for kk, p in zip(k, pad))
    gx = ops.conv3d(g.to(dtype).cuda(), pw, pad=dpad, out_grid=(T, H, W))
    r, a = tol(dtype, gx_ref)
    # bf16: a*W is rounded once more when folded
    torch.testing.assert_close(gx.float().cpu(), gx_ref, rtol=r * 2, atol=a * 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("k,s,dims", [((1, 3, 3), (1, 2, 2), (4, 28, 28)), ((3, 3, 3), (2, 2, 2), (8, 14, 14)),
                                      ((2, 2, 2), (2, 2, 2), (4, 14, 14)), ((3, 3, 3), (1, 1, 1), (3, 7, 7)),
                                      ((3, 3, 3), (2, 2, 2), (5, 7, 9)), ((3, 3, 3), (1, 1, 1), (5, 13, 30)),
                                      ((1, 3, 3), (1, 1, 1), (3, 9, 9)), ((3, 3, 3), (2, 2, 2), (23, 14, 14)), ((1, 3, 3), (1, 2, 2), (3, 15, 13))],
                         ids=["2a", "4a", "5a", "branch3", "odd", "branch3_ragged_tiles", "1x3x3_s1", "4a_T90_odd_frames", "1x3x3_s2_odd"])
def test_maxpool_fwd_bwd(ops, dtype, k, s, dims):
    B, C_ = 2, 24
    T, H, W = dims
    x = torch.relu(q(rnd((B, T, H, W, C_), 21), dtype))        # post-ReLU data: all-zero windows tie constantly
    xr = cf(x).requires_grad_(True)
    pads = []
    for i in (2, 1, 0):
        n = (T, H, W)[i]
        out = -(-n // s[i]); tot = max((out - 1) * s[i] + k[i] - n, 0)
        pads += [tot // 2, tot - tot // 2]
    yr = F.max_pool3d(F.pad(xr, pads, value=float("-inf")), k, s)
    out, idx, ctx = ops.maxpool3d(x.to(dtype).cuda(), k, s)
    torch.testing.assert_close(out.float().cpu(), cl(yr.detach()), rtol=0, atol=0)
    g = q(rnd(tuple(out.shape), 22), dtype)
    (gr,) = torch.autograd.grad(yr, xr, cf(g))
    gin = ops.maxpool3d_bwd(ctx, g.to(dtype).cuda())
    # routing of ties may differ only where the activation is 0 -> ReluGrad kills it; compare on x > 0 and the totals
    gref = cl(gr)
    pos = x > 0
    r, a = tol(dtype, gref)
    torch.testing.assert_close(gin.float().cpu()[pos], gref[pos], rtol=r, atol=a)
    masked = ops.maxpool3d_bwd(ctx, g.to(dtype).cuda(), mask=x.to(dtype).cuda()).float().cpu()
    torch.testing.assert_close(masked, torch.where(pos, gref, torch.zeros_like(gref)), rtol=r, atol=a)
    # first-max-in-scan-order is torch-CPU's rule too: without ties broken differently the full tensors agree
    torch.testing.assert_close(gin.float().cpu(), gref, rtol=r, atol=a)
    # relu_input: windows with a non-positive maximum record "no cell" -> backward WITHOUT a mask == masked backward
    out2, idx2, ctx2 = ops.maxpool3d(x.to(dtype).cuda(), k, s, relu_input=True)
    torch.testing.assert_close(out2.float().cpu(), out.float().cpu(), rtol=0, atol=0)
    nomask = ops.maxpool3d_bwd(ctx2, g.to(dtype).cuda()).float().cpu()
    # stride-1 windows use the scatter kernel (LDS float atomics): the order of the <= 27 additions per cell is not fixed
    torch.testing.assert_close(nomask, masked, rtol=1e-5, atol=1e-6 * float(g.abs().max()))


@pytest.mark.parametrize("dims,C_,K", [((4, 14, 14), 72, 64), ((3, 7, 9), 32, 32), ((4, 7, 7), 832, 128), ((5, 11, 6), 96, 96)])
def test_maxpool_bwd_gemm_fused(ops, dims, C_, K, monkeypatch):
    """flk_maxpool3d_bwd_gemm (Branch_3 backward of an Inception block in one kernel, i3d.py:211-216 backward): gin =
    MaxPool3DGrad(idx, g @ Wt) against (a) torch-CPU autograd of max_pool3d fed the fp32 product -- the kernel keeps the product in
    fp32, so only the bf16 rounding of the stored gin separates them -- and (b) the two-launch chain it replaces (1x1x1 data-gradient ->
    bf16 -> scatter), at bf16 tolerance; two runs give the same bits (integer LDS atomics)."""
    dtype = torch.bfloat16
    B = 2
    T, H, W = dims
    x = torch.relu(q(rnd((B, T, H, W, C_), 31), dtype))
    xr = cf(x).requires_grad_(True)
    yr = F.max_pool3d(F.pad(xr, [1, 1, 1, 1, 1, 1], value=float("-inf")), (3, 3, 3), (1, 1, 1))
    out, idx, ctx = ops.maxpool3d(x.to(dtype).cuda(), (3, 3, 3), (1, 1, 1))
    g = q(rnd((B, T, H, W, K + 8), 32), dtype)                       # the gradient lives at channel offset 8 of a wider buffer
    wt = q(rnd((K, C_), 33, 0.2), dtype)                             # Wt[k][c], bf16-representable
    gpl = (g[..., 8:8 + K].reshape(-1, K) @ wt).reshape(B, T, H, W, C_)          # fp32 product
    (gr,) = torch.autograd.grad(yr, xr, cf(gpl))
    gref = cl(gr)
    wp = ops.PoolGemmWeights(wt.numpy())
    gin = ops.maxpool3d_bwd_gemm(ctx, g.to(dtype).cuda(), wp, g_coff=8)
    r, a = tol(dtype, gref)
    torch.testing.assert_close(gin.float().cpu(), gref, rtol=r, atol=a)
    assert torch.equal(gin, ops.maxpool3d_bwd_gemm(ctx, g.to(dtype).cuda(), wp, g_coff=8))
    chain = ops.maxpool3d_bwd(ctx, gpl.to(dtype).cuda())              # the two-launch form: the product rounded to bf16 first
    torch.testing.assert_close(gin.float().cpu(), chain.float().cpu(), rtol=2e-2, atol=2e-2 * float(gref.abs().max()))
    # the loop form of the kernel (FLK_POOL_GEMM_REG=0; tiles reached by more than 512 windows) adds the same integers: same bits
    monkeypatch.setenv("FLK_POOL_GEMM_REG", "0")
    assert torch.equal(gin, ops.maxpool3d_bwd_gemm(ctx, g.to(dtype).cuda(), wp, g_coff=8))


BLOCKS = {"small": (1, 2, 7, 7, 64, (32, 24, 48, 16, 32, 16)),
          "Mixed_5c": (1, 2, 7, 7, 832, (384, 192, 384, 48, 128, 128)),
          "Mixed_3b": (1, 4, 28, 28, 192, (64, 96, 128, 16, 32, 32)),
          "Mixed_4c": (2, 4, 14, 14, 512, (160, 112, 224, 24, 64, 64))}


def pick_nf(c, taps):
    return 8 if c > 64 else (4 if c > 32 else 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("blk", list(BLOCKS))
def test_inception_block_forward_backward(ops, dtype, blk):
    """One Mixed block (i3d.py:194-219) composed from the per-op C ABI exactly as csrc/net.cpp wires it
    (slice writes into the concat buffer; backward = branch-3 dgrad -> pool grad -> 3x3x3 dgrads with ReLU
    masks -> three accumulating 1x1 dgrads), against torch-CPU autograd."""
    B, T, H, W, cin, (c0, c1a, c1b, c2a, c2b, c3) = BLOCKS[blk]
    ctot = c0 + c1b + c2b + c3
    rng = np.random.default_rng(0)

    def mk(k, ci, co, seed):
        w = q(rnd((*k, ci, co), seed, (2.0 / (ci * k[0] * k[1] * k[2])) ** 0.5), dtype)
        return w, rnd((co,), seed + 100).abs() * 0.5 + 0.75, rnd((co,), seed + 200) * 0.1

    one, three = (1, 1, 1), (3, 3, 3)
    P = {"0": mk(one, cin, c0, 1), "1a": mk(one, cin, c1a, 2), "1b": mk(three, c1a, c1b, 3), "2a": mk(one, cin, c2a, 4),
         "2b": mk(three, c2a, c2b, 5), "3": mk(one, cin, c3, 6)}
    xin = torch.relu(q(rnd((B, T, H, W, cin), 7), dtype))     # block input = a ReLU output
    gout = q(rnd((B, T, H, W, ctot), 8), dtype)

    # ---- oracle ----
    xr = xin.clone().requires_grad_(True)

    def unit(x, key, k):
        w, a, b = P[key]
        return torch.relu(ref_conv(x, w, (1, 1, 1), tuple((kk - 1) // 2 for kk in k), (T, H, W)) * a + b)

    b0 = unit(xr, "0", one)
    y1a = unit(xr, "1a", one); b1 = unit(y1a, "1b", three)
    y2a = unit(xr, "2a", one); b2 = unit(y2a, "2b", three)
    pl = cl(F.max_pool3d(F.pad(cf(xr), (1, 1, 1, 1, 1, 1), value=float("-inf")), 3, 1))
    b3 = unit(pl, "3", one)
    out_ref = torch.cat([b0, b1, b2, b3], -1)
    G_ref = torch.where(out_ref > 0, gout, torch.zeros_like(gout))      # masked gradient the G buffer holds
    gx_ref, gpl_ref, g1a_ref, g2a_ref = torch.autograd.grad(out_ref, [xr, pl, y1a, y2a], gout)
    g1a_ref = torch.where(y1a > 0, g1a_ref, torch.zeros_like(g1a_ref))
    g2a_ref = torch.where(y2a > 0, g2a_ref, torch.zeros_like(g2a_ref))
    gx_ref = torch.where(xin > 0, gx_ref, torch.zeros_like(gx_ref))

    # ---- HIP, wired like net.cpp ----
    dev = lambda t: t.to(dtype).cuda()
    x = dev(xin)
    out = torch.zeros((B, T, H, W, ctot), dtype=dtype).cuda()
    mid = torch.zeros((B, T, H, W, c1a + c2a), dtype=dtype).cuda()
    Wf = {k: ops.ConvWeights(v[0].numpy(), dtype, pick_nf(v[0].shape[4], 0)) for k, v in P.items()}
    Wb = {k: ops.ConvWeights(v[0].numpy(), dtype, pick_nf(v[0].shape[3], 0), row_scale=v[1].numpy(), transpose=True)
          for k, v in P.items()}
    sc = {k: (v[1].cuda(), v[2].cuda()) for k, v in P.items()}
    ops.conv3d(x, Wf["0"], out=out, out_coff=0, scale=sc["0"][0], bias=sc["0"][1], relu=True)
    ops.conv3d(x, Wf["1a"], out=mid, out_coff=0, scale=sc["1a"][0], bias=sc["1a"][1], relu=True)
    ops.conv3d(mid, Wf["1b"], in_coff=0, cin=c1a, out=out, out_coff=c0, scale=sc["1b"][0], bias=sc["1b"][1], relu=True)
    ops.conv3d(x, Wf["2a"], out=mid, out_coff=c1a, scale=sc["2a"][0], bias=sc["2a"][1], relu=True)
    ops.conv3d(mid, Wf["2b"], in_coff=c1a, cin=c2a, out=out, out_coff=c0 + c1b, scale=sc["2b"][0], bias=sc["2b"][1], relu=True)
    pool, idx, pctx = ops.maxpool3d(x, three, one)
    ops.conv3d(pool, Wf["3"], out=out, out_coff=c0 + c1b + c2b, scale=sc["3"][0], bias=sc["3"][1], relu=True)
    r, a = tol(dtype, out_ref)
    torch.testing.assert_close(out.float().cpu(), out_ref.detach(), rtol=r * 2, atol=a * 2)

    G = dev(torch.where(out.float().cpu() > 0, gout, torch.zeros_like(gout)))
    Gpl = ops.conv3d(G, Wb["3"], in_coff=c0 + c1b + c2b, cin=c3)
    gxa = ops.maxpool3d_bwd(pctx, Gpl)
    Gmid = torch.zeros_like(mid)
    ops.conv3d(G, Wb["2b"], in_coff=c0 + c1b, cin=c2b, out=Gmid, out_coff=c1a, mask=mid, mask_coff=c1a)
    ops.conv3d(G, Wb["1b"], in_coff=c0, cin=c1b, out=Gmid, out_coff=0, mask=mid, mask_coff=0)
    if dtype == torch.float32:
        for nm, got, want in (("branch-3 1x1 dgrad", Gpl, gpl_ref), ("1b dgrad", Gmid[..., :c1a], g1a_ref), ("2b dgrad", Gmid[..., c1a:], g2a_ref)):
            r, a = tol(dtype, want)
            frac = ((got.float().cpu() - want).abs() > a * 4 + r * 4 * want.abs()).float().mean().item()
            assert frac < 0.01, (nm, frac)
    Gin = torch.zeros_like(x)
    ops.conv3d(G, Wb["0"], in_coff=0, cin=c0, out=Gin, add=gxa)
    ops.conv3d(Gmid, Wb["1a"], in_coff=0, cin=c1a, out=Gin, add=Gin)
    ops.conv3d(Gmid, Wb["2a"], in_coff=c1a, cin=c2a, out=Gin, add=Gin, mask=x)
    r, a = tol(dtype, gx_ref)
    # a ReLU mask can flip where an activation is within rounding of 0 (fp32 too: different summation order)
    # -> isolated O(1) differences on that unit's receptive field; everything else must agree tightly
    bad = ((Gin.float().cpu() - gx_ref).abs() > a * 4 + r * 4 * gx_ref.abs()).float().mean().item()
    assert bad < (0.01 if dtype == torch.float32 else 0.03), bad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_two_segment_output_and_input(ops, dtype):
    """fused Inception 1x1x1 convolutions: one GEMM writing [b0 | b1a+b2a] into two buffers (out/out2), and its
    data-gradient reading K from two gradient buffers (in/in2, each segment padded to a slab in the packed weights)"""
    B, T, H, W, cin = 2, 3, 7, 9, 72
    c0, c1 = 112, 40                     # 112 is not a multiple of the bf16 slab (32): exercises the segment padding
    x = q(rnd((B, T, H, W, cin), 31), dtype)
    w = q(rnd((1, 1, 1, cin, c0 + c1), 32, 0.1), dtype)
    sc, bi = rnd((c0 + c1,), 33).abs() + 0.5, rnd((c0 + c1,), 34) * 0.1
    ref = torch.relu(ref_conv(x, w, (1, 1, 1), (0, 0, 0), (T, H, W)) * sc + bi)
    outA = torch.full((B, T, H, W, c0 + 16), 3.0, dtype=dtype).cuda()       # segment 1 at coff 8
    outB = torch.full((B, T, H, W, c1 + 8), 3.0, dtype=dtype).cuda()        # segment 2 at coff 8
    pw = ops.ConvWeights(w.numpy(), dtype, 8)
    ops.conv3d(x.to(dtype).cuda(), pw, out=outA, out_coff=8, out2=outB, out2_coff=8, cout1=c0, scale=sc.cuda(), bias=bi.cuda(), relu=True)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(outA.float().cpu()[..., 8:8 + c0], ref[..., :c0], rtol=r, atol=a)
    torch.testing.assert_close(outB.float().cpu()[..., 8:], ref[..., c0:], rtol=r, atol=a)
    assert (outA.float().cpu()[..., :8] == 3).all() and (outA.float().cpu()[..., 8 + c0:] == 3).all() and (outB.float().cpu()[..., :8] == 3).all()
    # data gradient: gx = [gA | gB] . (a * W)^T
    gA, gB = q(rnd((B, T, H, W, c0 + 8), 35), dtype), q(rnd((B, T, H, W, c1), 36), dtype)
    g = torch.cat([gA[..., 8:], gB], -1)
    xr = torch.zeros((B, T, H, W, cin), requires_grad=True)
    (gx_ref,) = torch.autograd.grad(ref_conv(xr, w, (1, 1, 1), (0, 0, 0), (T, H, W)) * sc, xr, g)
    wT = w[0, 0, 0].t().contiguous().reshape(1, 1, 1, c0 + c1, cin)
    pb = ops.ConvWeights(wT.numpy(), dtype, 4, row_scale=sc.numpy(), cin_split=c0)
    gx = ops.conv3d(gA.to(dtype).cuda(), pb, in_coff=8, cin=c0 + c1, in2=gB.to(dtype).cuda(), in2_coff=0)
    r, a = tol(dtype, gx_ref)
    torch.testing.assert_close(gx.float().cpu(), gx_ref, rtol=r * 2, atol=a * 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_multi_tap_two_input_segments_large_halo(ops, dtype):
    """A 3x3x3 convolution whose input channels come from TWO buffers (in / in2, weights packed with cin_split) on a grid large enough for a
    > 256-cell halo: the large-halo forms of conv_igemm_kernel stage by LDS-DMA with per-lane source addresses since round 5, and the source
    segment (base pointer, channel stride, valid chunks) changes from slab to slab.  Against the torch-CPU oracle on the concatenated input."""
    B, T, H, W = 1, 6, 18, 20
    c0, c1, cout = 40, 24, 48            # 40: the first segment's last slab is half valid (bf16 slab = 32 channels, fp32 = 16)
    xa, xb = q(rnd((B, T, H, W, c0 + 8), 51), dtype), q(rnd((B, T, H, W, c1), 52), dtype)
    w = q(rnd((3, 3, 3, c0 + c1, cout), 53, (2.0 / (27 * (c0 + c1))) ** 0.5), dtype)
    sc, bi = rnd((cout,), 54).abs() + 0.5, rnd((cout,), 55) * 0.1
    x = torch.cat([xa[..., 8:], xb], -1)
    ref = torch.relu(ref_conv(x, w, (1, 1, 1), (1, 1, 1), (T, H, W)) * sc + bi)
    pw = ops.ConvWeights(w.numpy(), dtype, 4, cin_split=c0)
    y = ops.conv3d(xa.to(dtype).cuda(), pw, in_coff=8, cin=c0 + c1, in2=xb.to(dtype).cuda(), in2_coff=0, scale=sc.cuda(), bias=bi.cuda(), relu=True)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(y.float().cpu(), ref, rtol=r, atol=a)


@pytest.mark.parametrize("nf", [4, 6, 8])
def test_conv1x1_dma_ring_all_features(ops, nf):
    """conv1x1_dma_kernel (bf16 1x1x1 GEMMs of >= 2048 positions: both operands through the LDS-DMA ring) with everything the fused
    Inception GEMMs use at once -- a position count that is not a multiple of the 256-position tile, cin % 32 != 0 (invalid channel
    chunks fetch chunk 0 against zero weights), two output segments with offsets, then the data-gradient form: K gathered from two
    gradient buffers + accumulate operand + ReLU mask -- against torch-CPU"""
    dtype = torch.bfloat16
    B, T, H, W, cin = 1, 3, 27, 29, 72                       # 2349 positions = 9 tiles + 45 rows
    c0, c1 = 112, 40
    x = q(rnd((B, T, H, W, cin + 8), 41), dtype)              # read at coff 8
    w = q(rnd((1, 1, 1, cin, c0 + c1), 42, 0.1), dtype)
    sc, bi = rnd((c0 + c1,), 43).abs() + 0.5, rnd((c0 + c1,), 44) * 0.1
    ref = torch.relu(ref_conv(x[..., 8:].contiguous(), w, (1, 1, 1), (0, 0, 0), (T, H, W)) * sc + bi)
    outA = torch.full((B, T, H, W, c0 + 16), 3.0, dtype=dtype).cuda()
    outB = torch.full((B, T, H, W, c1 + 8), 3.0, dtype=dtype).cuda()
    pw = ops.ConvWeights(w.numpy(), dtype, nf)
    ops.conv3d(x.to(dtype).cuda(), pw, in_coff=8, cin=cin, out=outA, out_coff=8, out2=outB, out2_coff=8, cout1=c0, scale=sc.cuda(), bias=bi.cuda(), relu=True)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(outA.float().cpu()[..., 8:8 + c0], ref[..., :c0], rtol=r, atol=a)
    torch.testing.assert_close(outB.float().cpu()[..., 8:], ref[..., c0:], rtol=r, atol=a)
    assert (outA.float().cpu()[..., :8] == 3).all() and (outA.float().cpu()[..., 8 + c0:] == 3).all() and (outB.float().cpu()[..., :8] == 3).all()
    # data-gradient form: gx = ([gA | gB] . (a W)^T + add) masked
    gA, gB = q(rnd((B, T, H, W, c0 + 8), 45), dtype), q(rnd((B, T, H, W, c1), 46), dtype)
    g = torch.cat([gA[..., 8:], gB], -1)
    add = q(rnd((B, T, H, W, cin + 8), 47), dtype)
    mask = q(rnd((B, T, H, W, cin), 48), dtype)
    gx_ref = torch.einsum("bthwk,ck->bthwc", g * sc, w[0, 0, 0]) + add[..., 8:]
    gx_ref = torch.where(mask > 0, gx_ref, torch.zeros_like(gx_ref))
    wT = w[0, 0, 0].t().contiguous().reshape(1, 1, 1, c0 + c1, cin)
    pb = ops.ConvWeights(wT.numpy(), dtype, nf, row_scale=sc.numpy(), cin_split=c0)
    gx = ops.conv3d(gA.to(dtype).cuda(), pb, in_coff=8, cin=c0 + c1, in2=gB.to(dtype).cuda(), in2_coff=0,
                    add=add.to(dtype).cuda(), add_coff=8, mask=mask.to(dtype).cuda())
    r, a = tol(dtype, gx_ref)
    torch.testing.assert_close(gx.float().cpu(), gx_ref, rtol=r * 2, atol=a * 2)
    # the register-queue path (3x fewer positions: below the ring's threshold) gives the same values up to summation order
    xs = x[:, :1, :20, :20].contiguous()
    small = ops.conv3d(xs.to(dtype).cuda(), pw, in_coff=8, cin=cin, scale=sc.cuda(), bias=bi.cuda(), relu=True)
    torch.testing.assert_close(small.float().cpu(), ref[:, :1, :20, :20], rtol=r, atol=a)


@pytest.mark.parametrize("case", ["u8", "f32_torch_dialect", "cyclic_u8", "u8_H36"])
def test_stem_delta_grad_fused(case):
    """flk_stem_delta_grad (csrc/stem_grad.hip): d(loss)/d(delta[t,c]) in ONE kernel from the stem's output gradient G -- against
    fp64 torch-CPU: gx = conv_transpose3d(G, W * bn_scale) (the data-gradient of the 7x7x7 / 2 SAME convolution, i3d.py:169), masked by
    the clip of the perturbed clip and summed over (b,h,w) (kinetics_i3d_utils.py:100-142).  G is bf16 on both sides, weights fp32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    from flickering_adversarial_video_amd import ops
    from oracle import attack_math as am
    rng = np.random.default_rng(17)
    B, T, H, W = 2, 8, 224, 224
    if case == "u8_H36":
        H = 36                                          # not a multiple of the mask pre-pass's 8-row blocks: the last block has 4 rows
    To, Ho, Wo = T // 2, H // 2, W // 2
    w7 = (rng.standard_normal((7, 7, 7, 3, 64)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    G = torch.from_numpy(rng.standard_normal((B, To, Ho, Wo, 64)).astype(np.float32)).to(torch.bfloat16)
    G[:, :, 5:9] = 0                                   # some structure: zero rows, one strong row
    G[:, :, Ho - 4] *= 8
    delta = rng.uniform(-0.5, 0.5, (T, 3)).astype(np.float32)       # some entries beyond the +-0.4 clip
    if case == "f32_torch_dialect":
        x = torch.from_numpy(rng.uniform(-2.2, 2.9, (B, T, H, W, 3)).astype(np.float32))    # beyond [min_v, max_v] on both sides
        kw = dict(dialect="torch", dclip=0.2, inv_std=tuple(1.0 / s for s in am.DEFAULT_STD), lo=am.TORCH_MIN_VALUE, hi=am.TORCH_MAX_VALUE)
        xf = x.double()
    else:
        xu = torch.from_numpy(rng.integers(0, 256, (B, T, H, W, 3), dtype=np.uint8))
        xu[:, :, :, :20] = 0                            # dark / bright bands: pixels that saturate as soon as delta has the right sign
        xu[:, :, :, 200:] = 255
        x = xu
        kw = dict(dialect="tf", dclip=0.4)
        xf = xu.double() / 128 - 1
    sx, sp = (3, 5) if case == "cyclic_u8" else (0, 0)
    args = ops.make_apply_args(x.cuda(), torch.from_numpy(delta).cuda(), shift_x=sx, shift_p=sp, fold_t=ops.I3D_FOLD, **kw)
    wts = ops.StemDeltaGradWeights(w7, scale)
    got = ops.stem_delta_grad(args, G.cuda(), wts).cpu().double()
    # ---- reference ----
    d = torch.from_numpy(delta).double().requires_grad_(True)
    dc = d.clamp(-kw["dclip"], kw["dclip"]) * torch.tensor(kw.get("inv_std", (1.0, 1.0, 1.0)), dtype=torch.float64)
    xr = torch.roll(xf, sx, 1) if sx else xf
    pr = torch.roll(dc, sp, 0) if sp else dc
    xa = (xr + pr.view(1, T, 1, 1, 3)).clamp(kw.get("lo", -1.0), kw.get("hi", 1.0))
    wt = torch.from_numpy(w7 * scale).double().permute(4, 3, 0, 1, 2).contiguous()           # [co, c, kt, kh, kw]
    xp = F.pad(xa.permute(0, 4, 1, 2, 3), (2, 3, 2, 3, 2, 3))                                 # TF SAME for k 7, s 2 on even sizes
    y = F.conv3d(xp, wt, None, stride=2)                                                      # [B,64,To,Ho,Wo]
    (ref,) = torch.autograd.grad(y, d, grad_outputs=G.double().permute(0, 4, 1, 2, 3))
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"[{case}] fused stem delta-gradient: max-rel {err:.2e} (|ref| max {float(ref.abs().max()):.3e})")
    assert err < 2e-5
    clipped = torch.from_numpy(np.abs(delta) > kw["dclip"])
    assert clipped.any() and float(got[clipped].abs().max()) == 0.0


def test_stem_delta_grad_in_engine():
    """the I3D engine's bf16 step with the fused kernel (default) against the two-kernel path (FLK_STEM_FUSED=0 semantics via
    net.backward + perturb_grad_reduce): same G, so the two delta-gradients differ only by the bf16 rounding of the data-gradient
    weights and of gx -- and the learned delta after 3 steps stays within 2 %."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec, ops
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    T, B = 16, 2
    W = i3d_spec.synthetic_i3d_weights(42)
    x = torch.from_numpy(i3d_spec.synthetic_clip_u8(B, T, seed=8)).cuda()
    eng = FlickerI3D(W, batch_size=B, frames=T, dtype="bf16")
    assert eng.fused_delta_grad and eng.net.has_backward_delta
    labels = eng.logits(x, adv_flag=0.0).argmax(-1).clone()
    eng.reset_perturbation(np.random.default_rng(1).uniform(-0.45, 0.45, (T, 3)).astype(np.float32))
    r = eng.step(x, labels, update=False)
    g_fused = eng.delta_gradient().clone()
    eng.fused_delta_grad = False
    eng.step(x, labels, update=False)
    g_two = eng.delta_gradient().clone()
    e = float((g_fused - g_two).abs().max() / g_two.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(g_fused.flatten(), g_two.flatten(), 0))
    print(f"fused vs two-kernel delta-gradient: max-rel {e:.2e}, cosine {cos:.6f}")
    assert e < 2e-2 and cos > 0.9999
    assert bool((g_fused[eng.eps_rgb.abs() > 0.4] == 0).all())
    # fp32 engines and dense perturbations keep the two-kernel path
    assert not FlickerI3D(W, batch_size=1, frames=T, dtype="f32").fused_delta_grad
    assert not FlickerI3D(W, batch_size=1, frames=T, dtype="bf16", dense_delta=True).fused_delta_grad


GROUP_CASES = [
    # name, B,T,H,W, (cin1, cout1, nf1), (cin2, cout2, nf2), nfw      -- an Inception block's Branch_1 / Branch_2 3x3x3 pair in ONE launch
    ("4b_fwd", 2, 4, 14, 14, (96, 208, 8), (16, 48, 4), 4),            # wn 2 + wn 1 on four fragments per wave
    ("4b_dgrad", 2, 4, 14, 14, (208, 96, 8), (48, 16, 2), 2),          # wn 4 + wn 1 on two fragments per wave
    ("4f_fwd", 1, 4, 14, 14, (160, 320, 4), (32, 128, 8), 4),          # wn 1 + wn 2, five channel tiles + one
    ("5b_fwd", 2, 2, 7, 7, (160, 320, 4), (32, 128, 8), 2),            # wn 2 + wn 4 (the Mixed_5 tiles: 7 x 7 planes)
    ("5c_dgrad_ragged", 1, 3, 7, 9, (384, 192, 4), (128, 48, 4), 2),   # ragged tiles, a partial channel tile in member 2
    ("4f_fwd_ring", 1, 4, 14, 14, (160, 320, 4), (32, 128, 4), -4),     # RING members (nfw < 0 here: ring with 64-channel tiles): five tiles + two
    ("3b_fwd_ring8", 1, 4, 12, 12, (96, 128, 8), (16, 32, 8), -8),      # ring, 128-channel tiles: member 2 fills a quarter of one
]


@pytest.mark.parametrize("case", GROUP_CASES, ids=[c[0] for c in GROUP_CASES])
def test_conv_group(ops, case):
    """flk_conv3d_group: two 3x3x3 convolutions over the same position grid in one grid of conv_igemm_group_kernel.  Each member against
    the torch-CPU oracle (bf16 tolerances) with the epilogue the plan uses (scale, bias, ReLU forward; ReLU mask backward), and BITWISE
    against flk_conv3d launched separately on the same packed weights (the members keep their own K order and tiles)."""
    _, B, T, H, W, m1, m2, nfw = case
    dtype = torch.bfloat16
    ctot = m1[0] + m2[0]
    xin = q(rnd((B, T, H, W, ctot), 11), dtype)                       # one input buffer, two channel slices (the plan's `mid`)
    otot = m1[1] + m2[1] + 8
    members, refs, singles = [], [], []
    out_g = torch.zeros((B, T, H, W, otot), dtype=dtype, device="cuda")
    out_s = torch.zeros_like(out_g)
    xg = xin.to(dtype).cuda()
    maskt = q(rnd((B, T, H, W, otot), 12), dtype)
    in_off, out_off = 0, 8
    for i, (cin, cout, nf) in enumerate((m1, m2)):
        w = q(rnd((3, 3, 3, cin, cout), 20 + i, (2.0 / (27 * cin)) ** 0.5), dtype)
        sc = rnd((cout,), 30 + i).abs() + 0.5
        bi = rnd((cout,), 40 + i, 0.1)
        ref = ref_conv(xin[..., in_off:in_off + cin].contiguous(), w, (1, 1, 1), (1, 1, 1), (T, H, W)) * sc + bi
        ref = torch.relu(ref) * (maskt[..., out_off:out_off + cout] > 0)
        refs.append((ref, out_off, cout))
        pw = ops.ConvWeights(w.numpy(), dtype, nf)
        kw = dict(in_coff=in_off, cin=cin, out_coff=out_off, scale=sc.cuda(), bias=bi.cuda(), relu=True, mask=maskt.to(dtype).cuda(),
                  mask_coff=out_off)
        members.append((xg, pw, dict(kw, out=out_g)))
        singles.append((pw, dict(kw, out=out_s)))
        in_off += cin
        out_off += cout
    ring = nfw < 0
    nfw = abs(nfw)
    ops.conv3d_group(members, nfw, ring=ring)
    for pw, kw in singles:
        ops.conv3d(xg, pw, **kw)
    torch.cuda.synchronize()
    for ref, off, cout in refs:
        r, a = tol(dtype, ref)
        torch.testing.assert_close(out_g[..., off:off + cout].float().cpu(), ref, rtol=r, atol=a)
    assert torch.equal(out_g, out_s)                                    # bitwise the separate launches (untouched channels stay zero)
    again = torch.zeros_like(out_g)
    ops.conv3d_group([(x_, w_, dict(k_, out=again)) for x_, w_, k_ in members], nfw, ring=ring)
    assert torch.equal(again, out_g)


T3_CASES = [
    # name, B, T, H, W, cin, cout, nf     -- (3,1,1) stride 1 pad 1, T in {16, 8, 4, 2}, H * W a multiple of 16, >= 2048 positions: the
    # whole-T LDS-DMA ring (conv_t3_dma_kernel); tiles = all frames of 16 / T groups of 16 spatial positions
    ("T16_144_64_layer1", 2, 16, 8, 8, 144, 64, 4),       # one group per tile; K = 4.5 slabs (the half slab fetches chunk 0 against zero weights)
    ("T8_64_144_ragged", 4, 8, 12, 12, 64, 144, 4),       # two groups per tile, 9 chunks per frame: the last tile of a clip has ONE valid group; 2.25 channel tiles
    ("T8_288_128_layer2", 2, 8, 16, 16, 288, 128, 8),     # nf = 8
    ("T4_64_64", 2, 4, 16, 20, 64, 64, 4),                # four groups per tile
    ("T2_64_64", 8, 2, 12, 12, 64, 64, 4),                # eight groups: every row has a zero-padded tap; partial last tile
    ("T6_halo_fallback", 2, 6, 14, 14, 144, 64, 4),       # T = 6: the halo kernels (no ring instance), same test
]


@pytest.mark.parametrize("case", T3_CASES, ids=[c[0] for c in T3_CASES])
def test_conv_temporal_dma(ops, case):
    """conv_t3_dma_kernel: the temporal half of a (2+1)D unit.  Forward with the plan's epilogue (scale, bias, residual add, ReLU) and the
    data-gradient form (transposed weights x BN scale, ReLU mask) against torch-CPU conv3d / its autograd at the bf16 tolerances; clip
    boundaries (zero padding in t only, never across clips) are where a flat-position kernel can go wrong -- several clips per case.  And
    bitwise against the halo kernel (the same weights packed in 32-channel tiles, which the ring does not take) -- same K order per output."""
    _, B, T, H, W, cin, cout, nf = case
    dtype = torch.bfloat16
    x = q(rnd((B, T, H, W, cin), 51), dtype)
    w = q(rnd((3, 1, 1, cin, cout), 52, (2.0 / (3 * cin)) ** 0.5), dtype)
    sc, bi = rnd((cout,), 53).abs() + 0.5, rnd((cout,), 54, 0.1)
    add = q(rnd((B, T, H, W, cout), 55), dtype)
    ref = torch.relu(ref_conv(x, w, (1, 1, 1), (1, 0, 0), (T, H, W)) * sc + bi + add)
    pw = ops.ConvWeights(w.numpy(), dtype, nf)
    kw = dict(scale=sc.cuda(), bias=bi.cuda(), add=add.to(dtype).cuda(), relu=True)
    out = ops.conv3d(x.to(dtype).cuda(), pw, **kw)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=r, atol=a)
    # data-gradient: g [.., cout] -> gx [.., cin], masked
    a_scale = rnd((cout,), 56).abs() + 0.5
    g = q(rnd((B, T, H, W, cout), 57), dtype)
    mask = q(rnd((B, T, H, W, cin), 58), dtype)
    xz = torch.zeros((B, T, H, W, cin), requires_grad=True)
    (gx_ref,) = torch.autograd.grad(ref_conv(xz, w, (1, 1, 1), (1, 0, 0), (T, H, W)) * a_scale, xz, g)
    gx_ref = gx_ref * (mask > 0)
    nfb = 8 if cin >= 128 else 4
    pwb = ops.ConvWeights(w.numpy(), dtype, nfb, row_scale=a_scale.numpy(), transpose=True)
    gx = ops.conv3d(g.to(dtype).cuda(), pwb, pad=(1, 0, 0), out_grid=(T, H, W), mask=mask.to(dtype).cuda())
    r, a = tol(dtype, gx_ref)
    torch.testing.assert_close(gx.float().cpu(), gx_ref, rtol=r * 2, atol=a * 2)
    # the halo kernels compute the same sums in the same order ((slab, tap) steps of 32 channels): weights packed in 32-channel tiles
    # (nf = 2) are outside the ring's route, so this call runs conv_igemm_kernel -- same bits
    out_h = ops.conv3d(x.to(dtype).cuda(), ops.ConvWeights(w.numpy(), dtype, 2), **kw)
    assert torch.equal(out_h, out)


@pytest.mark.parametrize("transpose", [False, True], ids=["fwd", "dgrad"])
def test_conv_ring_weights_a_row_ahead(ops, transpose):
    """conv_igemm_kernel mode 5 (ring kernels of 3-tap rows on large halos: the weights fetched a row of taps ahead from inline asm, one
    16-byte piece per thread and step on 64-channel tiles, two on 128- / 96-channel tiles) -- a shape large enough to take it (588
    workgroups), forward with the BN / ReLU epilogue and data-gradient with the ReLU mask against torch-CPU, and the three tile widths
    bitwise against each other (the same K order per output)."""
    dtype = torch.bfloat16
    B, T, H, W, cin, cout = 1, 12, 56, 56, 64, 192
    if transpose:
        cin, cout = cout, cin
    w = q(rnd((3, 3, 3, cin, cout), 61, (2.0 / (27 * cin)) ** 0.5), dtype)
    torch.set_num_threads(16)
    if not transpose:
        x = q(rnd((B, T, H, W, cin), 62), dtype)
        sc, bi = rnd((cout,), 63).abs() + 0.5, rnd((cout,), 64, 0.1)
        ref = torch.relu(ref_conv(x, w, (1, 1, 1), (1, 1, 1), (T, H, W)) * sc + bi)
        kw = dict(scale=sc.cuda(), bias=bi.cuda(), relu=True)
        run = lambda nf: ops.conv3d(x.to(dtype).cuda(), ops.ConvWeights(w.numpy(), dtype, nf), **kw)
    else:
        a_scale = rnd((cout,), 65).abs() + 0.5
        g = q(rnd((B, T, H, W, cout), 66), dtype)
        mask = q(rnd((B, T, H, W, cin), 67), dtype)
        xz = torch.zeros((B, T, H, W, cin), requires_grad=True)
        (ref,) = torch.autograd.grad(ref_conv(xz, w, (1, 1, 1), (1, 1, 1), (T, H, W)) * a_scale, xz, g)
        ref = ref * (mask > 0)
        run = lambda nf: ops.conv3d(g.to(dtype).cuda(), ops.ConvWeights(w.numpy(), dtype, nf, row_scale=a_scale.numpy(), transpose=True),
                                    pad=(1, 1, 1), out_grid=(T, H, W), mask=mask.to(dtype).cuda())
    out4, out8, out6 = run(4), run(8), run(6)
    r, a = tol(dtype, ref)
    torch.testing.assert_close(out4.float().cpu(), ref, rtol=r * 2, atol=a * 2)
    assert torch.equal(out4, out8) and torch.equal(out4, out6)
    assert torch.equal(run(4), out4)


PC_CASES = [
    # name, B, T, H, W, [(cin, cout), ...] members, data-gradient form?   -- flk_conv3d_pc: 3x3x3 stride 1, weights packed with nf = 4
    ("fwd_64_192_8x8x8_tiles", 1, 16, 16, 24, [(64, 192)], False),              # 512-row tiles, two slabs, three channel tiles (Conv3d_2c's shape class)
    ("dgrad_192_64", 1, 8, 16, 16, [(192, 64)], True),                          # six slabs; ReLU mask in the epilogue
    ("fwd_96_128_28x28_448_row_tiles", 1, 16, 28, 28, [(96, 128)], False),      # 16x4x7 tiles: seven fragments per consumer wave (NI = 7), three slabs (odd step count)
    ("group_128_192+32_96", 2, 8, 14, 14, [(128, 192), (32, 96)], False),       # two members (Mixed_3c Branch_1 + Branch_2): different K loops, 96 = 1.5 channel tiles
    ("group_dgrad_128_96+32_16", 1, 16, 14, 14, [(128, 96), (32, 16)], True),   # narrow outputs: 16 channels in a 64-wide tile
    ("ragged_40_72_partial_tiles", 3, 5, 9, 11, [(40, 72)], False),             # 40 input channels (a half-valid slab), grid not a multiple of any tile, several clips
    ("one_slab_16_32", 1, 8, 8, 8, [(16, 32)], False),                          # a 27-step item: fewer steps in flight than the ring is deep for the prologue
    # (1,3,3) taps -- the spatial half of a (2+1)D unit (torchvision Conv2Plus1D, model.py:421): 9-step slabs, no temporal halo
    ("k133_fwd_64_144", 2, 8, 28, 28, [(64, 144)], False, 1),                   # r2plus1d layer1's 64 -> 144: two slabs, 144 = 2.25 channel tiles
    ("k133_dgrad_144_64", 1, 8, 28, 28, [(144, 64)], True, 1),                  # its data-gradient: five slabs (the last half-valid), ReLU mask
    ("k133_one_slab_ragged", 2, 3, 9, 10, [(24, 40)], False, 1),                # one 9-step slab per item, partial tiles
]


@pytest.mark.parametrize("case", PC_CASES, ids=[c[0] for c in PC_CASES])
def test_conv_pc(ops, case):
    """flk_conv3d_pc (csrc/conv_pc.hip): the persistent producer / consumer form of the large 3x3x3 layers.  Each member against the torch-CPU
    oracle (bf16 tolerances) with the epilogue the plan uses -- scale, bias, ReLU forward; transposed weights x BN scale, ReLU mask in the
    data-gradient -- and BITWISE against flk_conv3d on the same packed weights (same K order per output, same epilogue); a second run gives
    the same bits (no ordering inside the kernel decides a value)."""
    _, B, T, H, W, mem, tr = case[:7]
    kt = case[7] if len(case) > 7 else 3
    pt = (kt - 1) // 2
    dtype = torch.bfloat16
    ci_tot = sum(c[0] for c in mem)
    co_tot = sum(c[1] for c in mem) + 8
    xin = q(rnd((B, T, H, W, ci_tot), 11), dtype)
    xg = xin.to(dtype).cuda()
    maskt = q(rnd((B, T, H, W, co_tot), 12), dtype)
    out_p = torch.zeros((B, T, H, W, co_tot), dtype=dtype, device="cuda")
    out_s = torch.zeros_like(out_p)
    members, singles, refs = [], [], []
    in_off, out_off = 0, 8
    for i, (cin, cout) in enumerate(mem):
        if tr:      # the data-gradient operator of a forward layer cout -> cin: G (cin channels here) -> gx (cout channels), BN scale folded
            w_fwd = q(rnd((kt, 3, 3, cout, cin), 20 + i, (2.0 / (9 * kt * cout)) ** 0.5), dtype)
            sc = rnd((cin,), 30 + i).abs() + 0.5
            g = xin[..., in_off:in_off + cin].contiguous()
            x0 = torch.zeros((B, T, H, W, cout), requires_grad=True)
            y = ref_conv(x0, w_fwd, (1, 1, 1), (pt, 1, 1), (T, H, W)) * sc
            (ref,) = torch.autograd.grad(y, x0, g)
            ref = ref * (maskt[..., out_off:out_off + cout] > 0)
            pw = ops.ConvWeights(w_fwd.numpy(), dtype, 4, transpose=True, row_scale=sc.numpy())
            kw = dict(in_coff=in_off, cin=cin, out_coff=out_off, mask=maskt.to(dtype).cuda(), mask_coff=out_off)
        else:
            w = q(rnd((kt, 3, 3, cin, cout), 20 + i, (2.0 / (9 * kt * cin)) ** 0.5), dtype)
            sc = rnd((cout,), 30 + i).abs() + 0.5
            bi = rnd((cout,), 40 + i, 0.1)
            ref = torch.relu(ref_conv(xin[..., in_off:in_off + cin].contiguous(), w, (1, 1, 1), (pt, 1, 1), (T, H, W)) * sc + bi)
            pw = ops.ConvWeights(w.numpy(), dtype, 4)
            kw = dict(in_coff=in_off, cin=cin, out_coff=out_off, scale=sc.cuda(), bias=bi.cuda(), relu=True)
        refs.append((ref, out_off, cout))
        members.append((xg, pw, dict(kw, out=out_p)))
        singles.append((pw, dict(kw, out=out_s)))
        in_off += cin
        out_off += cout
    ops.conv3d_pc(members)
    for pw, kw in singles:
        ops.conv3d(xg, pw, **kw)
    torch.cuda.synchronize()
    for ref, off, cout in refs:
        r, a = tol(dtype, ref)
        torch.testing.assert_close(out_p[..., off:off + cout].float().cpu(), ref, rtol=r * (2 if tr else 1), atol=a * (2 if tr else 1))   # (a * W is rounded once more when folded)
    assert torch.equal(out_p, out_s)                                    # bitwise flk_conv3d (untouched channels stay zero)
    again = torch.zeros_like(out_p)
    ops.conv3d_pc([(x_, w_, dict(k_, out=again)) for x_, w_, k_ in members])
    assert torch.equal(again, out_p)

"""The restated network topologies against PUBLISHED figures (the networks themselves live in third-party packages that cannot be
imported here: dm-sonnet / TF-1.15 for I3D, torchvision 0.5.0 for the VideoResNets -- SURVEY 8(c)): multiply-accumulate counts and
parameter counts computed from the oracle's own layer tables.  A wrong kernel size, stride, channel count or a missing layer in the
restatement changes these totals."""
from oracle import i3d_ref
from oracle import videoresnet_ref as vr


def _same(n, s):
    return -(-n // s)


def test_i3d_macs_and_parameters_match_the_reference_architecture():
    """InceptionI3d on 64 x 224 x 224 (i3d.py:144-479): 111.15 GMAC per clip forward and 12.68 M convolution weights (SURVEY
    Appendix A.1 -- the figure bench.py's TFLOP/s uses); the published model card counts 12.3 M parameters for the 400-class RGB
    stream without the batch-norm statistics, which is what the weights minus nothing-but-conv give within 3 %."""
    units = {n: (k, s, ci, co) for n, k, s, ci, co in i3d_ref.unit_names()}
    dims, macs, weights = (64, 224, 224), 0, 0

    def conv(name):
        nonlocal dims, macs, weights
        k, s, ci, co = units[name]
        dims = tuple(_same(d, st) for d, st in zip(dims, s))
        macs += dims[0] * dims[1] * dims[2] * k[0] * k[1] * k[2] * ci * co
        weights += k[0] * k[1] * k[2] * ci * co

    conv("Conv3d_1a_7x7")
    assert dims == (32, 112, 112)
    dims = (32, 56, 56)                                   # MaxPool3d_2a_3x3, (1,3,3) / (1,2,2) SAME
    conv("Conv3d_2b_1x1"); conv("Conv3d_2c_3x3")
    dims = (32, 28, 28)                                   # MaxPool3d_3a_3x3
    for name, spec in i3d_ref.MIXED:
        if name.startswith("MaxPool"):
            dims = tuple(_same(d, s) for d, s in zip(dims, spec[1]))
            continue
        block_in = dims
        for unit in ("/Branch_0/Conv3d_0a_1x1", "/Branch_1/Conv3d_0a_1x1", "/Branch_1/Conv3d_0b_3x3", "/Branch_2/Conv3d_0a_1x1",
                     "/Branch_2/" + i3d_ref.b2_3x3_name(name), "/Branch_3/Conv3d_0b_1x1"):
            dims = block_in
            conv(name + unit)
    assert dims == (8, 7, 7)
    # Logits: 2x7x7 VALID average pool -> 7 temporal positions -> 1x1x1 conv 1024 -> 400 (i3d.py:459-474)
    macs += 7 * 1024 * 400
    weights += 1024 * 400
    assert abs(macs / 1e9 - 111.15) < 0.06, macs / 1e9
    assert abs(weights / 1e6 - 12.68) < 0.02, weights / 1e6
    assert abs(weights / 12.3e6 - 1) < 0.04


def test_videoresnet_macs_and_parameters_match_torchvision_documentation():
    """torchvision's documented figures for 16 x 112 x 112 clips: r2plus1d_18 40.52 GFLOPs (multiply-accumulates) / 31.5 M parameters,
    r3d_18 40.70 / 33.4 M, mc3_18 43.34 / 11.7 M"""
    published = {"r2plus1d_18": (40.52, 31.5), "r3d_18": (40.70, 33.4), "mc3_18": (43.34, 11.7)}
    for arch, (gmac, mparams) in published.items():
        macs, params = 0, 0
        table = {row[0]: row for row in vr.layer_table(arch)}
        dims = (16, 112, 112)

        def conv(pre, stride, pad):
            nonlocal macs, params
            _, co, ci, k, bn = table[pre]
            out = tuple((d + 2 * p - kk) // s + 1 for d, p, kk, s in zip(dims, pad, k, stride))
            macs += out[0] * out[1] * out[2] * k[0] * k[1] * k[2] * ci * co
            params += k[0] * k[1] * k[2] * ci * co + 2 * co               # conv weight + BatchNorm gamma, beta
            return out

        if arch == "r2plus1d_18":
            dims = conv("stem.0", (1, 2, 2), (0, 3, 3))
            dims = conv("stem.3", (1, 1, 1), (1, 0, 0))
        else:
            dims = conv("stem.0", (1, 2, 2), (1, 3, 3))
        for name, kind, stride, has_ds in vr.blocks(arch):
            block_in = dims
            for cname, st in ((".conv1", stride), (".conv2", 1)):
                if kind == "2plus1d":
                    dims = conv(name + cname + ".0.0", (1, st, st), (0, 1, 1))
                    dims = conv(name + cname + ".0.3", (st, 1, 1), (1, 0, 0))
                elif kind == "3d":
                    dims = conv(name + cname + ".0", (st, st, st), (1, 1, 1))
                else:
                    dims = conv(name + cname + ".0", (1, st, st), (0, 1, 1))
            if has_ds:
                keep, dims = dims, block_in
                conv(name + ".downsample.0", vr.ds_stride(kind, stride), (0, 0, 0))
                dims = keep
        macs += 512 * 400
        params += 512 * 400 + 400
        assert abs(macs / 1e9 - gmac) < 0.02, (arch, macs / 1e9)
        assert abs(params / 1e6 - mparams) < 0.06, (arch, params / 1e6)

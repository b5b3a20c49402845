"""CPU restatement of InceptionI3d (reference i3d.py) -- TEST INFRASTRUCTURE, see oracle/__init__.py.

Follows i3d.py:32-71 (Unit3D = Conv3D SAME no-bias -> BatchNorm(inference, no gamma, eps 1e-3) -> ReLU)
and i3d.py:144-479 (topology).  dm-sonnet 1.23 / TF 1.15 are third-party and absent: their op
semantics (asymmetric SAME padding with the extra pad AFTER, -inf max-pool padding, scale-less BN)
are restated from the pinned packages' documented behaviour.  parity unpinned:
the reference holds no tests or golden vectors for the network.

Weights are a dict keyed by the checkpoint variable names (kinetics_i3d_utils.py:41-62), e.g.
``RGB/inception_i3d/Conv3d_1a_7x7/conv_3d/w`` [kt,kh,kw,Cin,Cout],
``.../batch_norm/{beta,moving_mean,moving_variance}`` [1,1,1,1,C], and for the logits layer
``RGB/inception_i3d/Logits/Conv3d_0c_1x1/conv_3d/{w,b}``.
Input is NDHWC float32 [B,T,H,W,3]; internally torch NCDHW conv3d (fp32, CPU) with explicit padding.
"""
import torch
import torch.nn.functional as F

PREFIX = "RGB/inception_i3d/"
BN_EPS = 1e-3  # dm-sonnet 1.23 snt.BatchNorm default


def same_pad(n, k, s):
    """TF SAME: out = ceil(n/s); total = max((out-1)*s + k - n, 0); before = total//2; after = rest."""
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def _pad3(x, k, s, value):
    (t0, t1), (h0, h1), (w0, w1) = (same_pad(x.shape[2 + i], k[i], s[i]) for i in range(3))
    if t0 + t1 + h0 + h1 + w0 + w1 == 0:
        return x
    return F.pad(x, (w0, w1, h0, h1, t0, t1), value=value)


def unit3d(x, W, name, k, s=(1, 1, 1), bn=True, relu=True, bias=False):
    """i3d.py:51-71. x NCDHW."""
    w = W[PREFIX + name + "/conv_3d/w"]                     # [kt,kh,kw,Cin,Cout]
    w = w.permute(4, 3, 0, 1, 2).contiguous()               # -> [Cout,Cin,kt,kh,kw]
    b = W[PREFIX + name + "/conv_3d/b"].reshape(-1) if bias else None
    y = F.conv3d(_pad3(x, k, s, 0.0), w, b, stride=s)
    if bn:
        beta = W[PREFIX + name + "/batch_norm/beta"].reshape(1, -1, 1, 1, 1)
        mean = W[PREFIX + name + "/batch_norm/moving_mean"].reshape(1, -1, 1, 1, 1)
        var = W[PREFIX + name + "/batch_norm/moving_variance"].reshape(1, -1, 1, 1, 1)
        y = (y - mean) * torch.rsqrt(var + BN_EPS) + beta
    return F.relu(y) if relu else y


def maxpool_same(x, k, s):
    """tf.nn.max_pool3d(padding=SAME): padded cells never win (-inf)."""
    return F.max_pool3d(_pad3(x, k, s, float("-inf")), k, s)


# (name, [b0, b1a, b1b, b2a, b2b, b3]) -- i3d.py:194-455.  Mixed_5b's branch-2 3x3 is named
# Conv3d_0a_3x3 in the reference (i3d.py:418); every other block names it Conv3d_0b_3x3.
MIXED = [
    ("Mixed_3b", (64, 96, 128, 16, 32, 32)),
    ("Mixed_3c", (128, 128, 192, 32, 96, 64)),
    ("MaxPool3d_4a_3x3", ((3, 3, 3), (2, 2, 2))),
    ("Mixed_4b", (192, 96, 208, 16, 48, 64)),
    ("Mixed_4c", (160, 112, 224, 24, 64, 64)),
    ("Mixed_4d", (128, 128, 256, 24, 64, 64)),
    ("Mixed_4e", (112, 144, 288, 32, 64, 64)),
    ("Mixed_4f", (256, 160, 320, 32, 128, 128)),
    ("MaxPool3d_5a_2x2", ((2, 2, 2), (2, 2, 2))),
    ("Mixed_5b", (256, 160, 320, 32, 128, 128)),
    ("Mixed_5c", (384, 192, 384, 48, 128, 128)),
]


def b2_3x3_name(block):
    return "Conv3d_0a_3x3" if block == "Mixed_5b" else "Conv3d_0b_3x3"


def mixed(x, W, name):
    """One Inception block (i3d.py:194-219 and repeats)."""
    b0 = unit3d(x, W, name + "/Branch_0/Conv3d_0a_1x1", (1, 1, 1))
    b1 = unit3d(x, W, name + "/Branch_1/Conv3d_0a_1x1", (1, 1, 1))
    b1 = unit3d(b1, W, name + "/Branch_1/Conv3d_0b_3x3", (3, 3, 3))
    b2 = unit3d(x, W, name + "/Branch_2/Conv3d_0a_1x1", (1, 1, 1))
    b2 = unit3d(b2, W, name + "/Branch_2/" + b2_3x3_name(name), (3, 3, 3))
    b3 = maxpool_same(x, (3, 3, 3), (1, 1, 1))
    b3 = unit3d(b3, W, name + "/Branch_3/Conv3d_0b_1x1", (1, 1, 1))
    return torch.cat([b0, b1, b2, b3], 1)


def i3d_logits(x_ndhwc, W, return_endpoints=False):
    """InceptionI3d._build(final_endpoint='Logits') (i3d.py:144-474). Returns [B,400] logits."""
    ep = {}
    x = x_ndhwc.permute(0, 4, 1, 2, 3).contiguous()
    x = unit3d(x, W, "Conv3d_1a_7x7", (7, 7, 7), (2, 2, 2)); ep["Conv3d_1a_7x7"] = x
    x = maxpool_same(x, (1, 3, 3), (1, 2, 2)); ep["MaxPool3d_2a_3x3"] = x
    x = unit3d(x, W, "Conv3d_2b_1x1", (1, 1, 1)); ep["Conv3d_2b_1x1"] = x
    x = unit3d(x, W, "Conv3d_2c_3x3", (3, 3, 3)); ep["Conv3d_2c_3x3"] = x
    x = maxpool_same(x, (1, 3, 3), (1, 2, 2)); ep["MaxPool3d_3a_3x3"] = x
    for name, spec in MIXED:
        x = maxpool_same(x, *spec) if name.startswith("MaxPool") else mixed(x, W, name)
        ep[name] = x
    # Logits head: avg_pool3d 2x7x7 VALID s1 -> dropout(keep 1.0) -> 1x1x1 conv + bias -> squeeze -> mean_T
    x = F.avg_pool3d(x, (2, 7, 7), (1, 1, 1))
    x = unit3d(x, W, "Logits/Conv3d_0c_1x1", (1, 1, 1), bn=False, relu=False, bias=True)
    logits = x.squeeze(4).squeeze(3).mean(2)
    return (logits, ep) if return_endpoints else logits


def unit_names():
    """All Unit3D paths with (kernel, stride, cin, cout) in forward order -- used by tests to build
    weight dicts; mirrors the table in SURVEY Appendix A.1."""
    out = [("Conv3d_1a_7x7", (7, 7, 7), (2, 2, 2), 3, 64),
           ("Conv3d_2b_1x1", (1, 1, 1), (1, 1, 1), 64, 64),
           ("Conv3d_2c_3x3", (3, 3, 3), (1, 1, 1), 64, 192)]
    cin = 192
    for name, spec in MIXED:
        if name.startswith("MaxPool"):
            continue
        c0, c1a, c1b, c2a, c2b, c3 = spec
        one, three, s1 = (1, 1, 1), (3, 3, 3), (1, 1, 1)
        out += [(name + "/Branch_0/Conv3d_0a_1x1", one, s1, cin, c0),
                (name + "/Branch_1/Conv3d_0a_1x1", one, s1, cin, c1a),
                (name + "/Branch_1/Conv3d_0b_3x3", three, s1, c1a, c1b),
                (name + "/Branch_2/Conv3d_0a_1x1", one, s1, cin, c2a),
                (name + "/Branch_2/" + b2_3x3_name(name), three, s1, c2a, c2b),
                (name + "/Branch_3/Conv3d_0b_1x1", one, s1, cin, c3)]
        cin = c0 + c1b + c2b + c3
    out.append(("Logits/Conv3d_0c_1x1", (1, 1, 1), (1, 1, 1), 1024, 400))
    return out

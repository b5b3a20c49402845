"""CPU restatement of torchvision 0.5.0 ``torchvision.models.video`` r2plus1d_18 / r3d_18 / mc3_18 -- TEST
INFRASTRUCTURE, see oracle/__init__.py.

The reference instantiates these from a third-party dependency that is NOT in its tree:
``getattr(torch_video_models, base_model)(True, True)`` (utils_cv/action_recognition/model.py:421), pinned
``torchvision==0.5.0`` / ``torch==1.4.0`` (requirements.txt:218,216).  torchvision is not installed here, so the published
architecture is restated (the MAC totals reproduce torchvision's documented 40.52 / 40.70 / 43.34
GFLOPs).  The reference holds no tests or golden vectors at this boundary: parity unpinned.

Weights: dict keyed by the torchvision ``state_dict`` names (``stem.0.weight``, ``layer1.0.conv1.0.0.weight`` ...),
conv weights in torch layout [Cout,Cin,kt,kh,kw]; BatchNorm3d in eval mode (gamma, beta, running stats, eps 1e-5).
Input NCDHW [B,3,T,H,W], normalised (dataset.py:28-29).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
PLANES = (64, 128, 256, 512)


def midplanes(inplanes, planes):
    """BasicBlock: computed ONCE per block from (inplanes, planes), used by conv1 and conv2"""
    return (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)


def bn(x, W, pre):
    return F.batch_norm(x, W[pre + ".running_mean"], W[pre + ".running_var"], W[pre + ".weight"], W[pre + ".bias"], False, 0.0, BN_EPS)


def conv_builder(arch, layer):
    """r2plus1d_18: Conv2Plus1D everywhere; r3d_18: Conv3DSimple; mc3_18: Conv3DSimple in layer1, Conv3DNoTemporal after"""
    if arch == "r2plus1d_18":
        return "2plus1d"
    if arch == "r3d_18" or layer == 1:
        return "3d"
    return "notemporal"


def ds_stride(kind, s):
    return (1, s, s) if kind == "notemporal" else (s, s, s)


def conv_unit(x, W, pre, kind, stride):
    """conv_builder(inplanes, planes, midplanes, stride): returns the conv output (no trailing BN)"""
    if kind == "3d":
        return F.conv3d(x, W[pre + ".weight"], None, (stride,) * 3, 1)
    if kind == "notemporal":
        return F.conv3d(x, W[pre + ".weight"], None, (1, stride, stride), (0, 1, 1))
    y = F.conv3d(x, W[pre + ".0.weight"], None, (1, stride, stride), (0, 1, 1))
    y = F.relu(bn(y, W, pre + ".1"))
    return F.conv3d(y, W[pre + ".3.weight"], None, (stride, 1, 1), (1, 0, 0))


def basic_block(x, W, pre, kind, stride, has_ds, final_relu=True):
    """final_relu=False returns the pre-ReLU sum (where a d(loss)/d(pre-activation) gradient buffer lives; used by the link tests)"""
    out = F.relu(bn(conv_unit(x, W, pre + ".conv1.0", kind, stride), W, pre + ".conv1.1"))
    out = bn(conv_unit(out, W, pre + ".conv2.0", kind, 1), W, pre + ".conv2.1")
    res = x
    if has_ds:
        res = bn(F.conv3d(x, W[pre + ".downsample.0.weight"], None, ds_stride(kind, stride)), W, pre + ".downsample.1")
    return F.relu(out + res) if final_relu else out + res


def stem(x, W, arch, final_relu=True):
    """torchvision R2Plus1dStem / BasicStem"""
    act = F.relu if final_relu else (lambda t: t)
    if arch == "r2plus1d_18":
        y = F.relu(bn(F.conv3d(x, W["stem.0.weight"], None, (1, 2, 2), (0, 3, 3)), W, "stem.1"))
        return act(bn(F.conv3d(y, W["stem.3.weight"], None, 1, (1, 0, 0)), W, "stem.4"))
    return act(bn(F.conv3d(x, W["stem.0.weight"], None, (1, 2, 2), (1, 3, 3)), W, "stem.1"))


def blocks(arch):
    """[(endpoint name, kind, stride, has_downsample)] in forward order"""
    out, inpl = [], 64
    for li, planes in enumerate(PLANES, start=1):
        kind = conv_builder(arch, li)
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            out.append((f"layer{li}.{bi}", kind, stride, stride != 1 or inpl != planes))
            inpl = planes
    return out


def videoresnet_logits(x, W, arch="r2plus1d_18", return_endpoints=False):
    ep = {}
    if arch == "r2plus1d_18":
        y = F.relu(bn(F.conv3d(x, W["stem.0.weight"], None, (1, 2, 2), (0, 3, 3)), W, "stem.1"))
        ep["stem.mid"] = y
        y = F.relu(bn(F.conv3d(y, W["stem.3.weight"], None, 1, (1, 0, 0)), W, "stem.4"))
    else:
        y = F.relu(bn(F.conv3d(x, W["stem.0.weight"], None, (1, 2, 2), (1, 3, 3)), W, "stem.1"))
    ep["stem"] = y
    inpl = 64
    for li, planes in enumerate(PLANES, start=1):
        kind = conv_builder(arch, li)
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            has_ds = stride != 1 or inpl != planes
            y = basic_block(y, W, f"layer{li}.{bi}", kind, stride, has_ds)
            ep[f"layer{li}.{bi}"] = y
            inpl = planes
    feat = y.mean(dim=(2, 3, 4))
    logits = F.linear(feat, W["fc.weight"], W["fc.bias"])
    return (logits, ep) if return_endpoints else logits


def layer_table(arch):
    """[(state_dict prefix, kind, cout, cin, (kt,kh,kw))] of every conv weight + BN, in forward order"""
    t = []
    if arch == "r2plus1d_18":
        t += [("stem.0", 45, 3, (1, 7, 7), "stem.1"), ("stem.3", 64, 45, (3, 1, 1), "stem.4")]
    else:
        t += [("stem.0", 64, 3, (3, 7, 7), "stem.1")]
    inpl = 64
    for li, planes in enumerate(PLANES, start=1):
        kind = conv_builder(arch, li)
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            pre = f"layer{li}.{bi}"
            mid = midplanes(inpl, planes)
            for cname, ci, co in ((".conv1", inpl, planes), (".conv2", planes, planes)):
                if kind == "2plus1d":
                    t += [(pre + cname + ".0.0", mid, ci, (1, 3, 3), pre + cname + ".0.1"),
                          (pre + cname + ".0.3", co, mid, (3, 1, 1), pre + cname + ".1")]
                else:
                    t += [(pre + cname + ".0", co, ci, (3, 3, 3) if kind == "3d" else (1, 3, 3), pre + cname + ".1")]
            if stride != 1 or inpl != planes:
                t += [(pre + ".downsample.0", planes, inpl, (1, 1, 1), pre + ".downsample.1")]
            inpl = planes
    return t

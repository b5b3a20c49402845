"""Synthetic fixtures for the parity tests -- TEST INFRASTRUCTURE, see oracle/__init__.py.

``coherent_i3d_weights``: a WELL-CONDITIONED InceptionI3d weight set for asserting the learned perturbation at the north-star
1e-3.  Why it is needed (measured, tests/test_i3d_gpu.py): on seeded He-normal (random-sign) weights two fp32 implementations of
the same forward pass disagree on ~10-30 of the ~3e7 ReLU / max-pool decisions of a 16-frame clip (units within fp32 rounding of
a tie: 4 ReLU + 23 pool flips between torch-CPU fp32 and fp64 on a smooth clip, 1 + 7 on a noise clip).  With random-sign weights
d(loss)/d(delta) is a random-sign sum over N units per layer, so ONE flipped unit moves it by ~1/sqrt(N) (1e-3 for the 4e5-unit
Mixed_4 layers): the torch-CPU fp32 oracle itself reproduces the fp64 gradient only to 2e-3...6e-3, whatever the clip (smooth or
noise), and Adam carries that into delta.  That is a property of the random-sign fixture, not of the path.

Here every convolution weight is >= 0 (|He-normal|), so all paths contribute to the gradient with the same sign and a flipped
decision moves it by ~1/N; batch-norm statistics are calibrated on the fixture clip itself (per-channel mean / variance of the
convolution output, like real BN statistics), which keeps about half of the units of every layer switched off -- the ReLU and
max-pool masks are as non-trivial as on the random fixture.  The logits layer favours one class so that the clean prediction has
a comfortable margin.  Measured: torch-CPU fp32 reproduces the fp64 delta after 6 Adam steps to 3e-6.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import i3d_ref

P = i3d_ref.PREFIX


def coherent_i3d_weights(xu, seed=5, label=233, num_classes=400):
    """xu: uint8 clip [1,T,224,224,3] the statistics are calibrated on.  Returns {checkpoint variable name: float32 array}."""
    rng = np.random.default_rng(seed)
    W = {}
    cout_of = {name: co for name, _, _, _, co in i3d_ref.unit_names()}
    x = (xu.double() / 128 - 1).permute(0, 4, 1, 2, 3).contiguous()

    def unit(x, name, k, s=(1, 1, 1)):
        cin, cout = x.shape[1], cout_of[name]
        w = np.abs(rng.standard_normal((*k, cin, cout))).astype(np.float32) * np.float32(np.sqrt(2.0 / (k[0] * k[1] * k[2] * cin)))
        W[P + name + "/conv_3d/w"] = w
        y = F.conv3d(i3d_ref._pad3(x, k, s, 0.0), torch.from_numpy(w).double().permute(4, 3, 0, 1, 2).contiguous(), None, stride=s)
        shp = (1, 1, 1, 1, cout)
        mean = y.mean(dim=(0, 2, 3, 4)).numpy().astype(np.float32).reshape(shp)
        var = np.maximum(y.var(dim=(0, 2, 3, 4), unbiased=False).numpy(), 1e-6).astype(np.float32).reshape(shp)
        beta = (rng.standard_normal(cout) * 0.3).astype(np.float32).reshape(shp)
        W[P + name + "/batch_norm/moving_mean"], W[P + name + "/batch_norm/moving_variance"], W[P + name + "/batch_norm/beta"] = mean, var, beta
        m, v, b = (torch.from_numpy(a).double().reshape(1, -1, 1, 1, 1) for a in (mean, var, beta))
        return F.relu((y - m) * torch.rsqrt(v + i3d_ref.BN_EPS) + b)

    x = unit(x, "Conv3d_1a_7x7", (7, 7, 7), (2, 2, 2))
    x = i3d_ref.maxpool_same(x, (1, 3, 3), (1, 2, 2))
    x = unit(x, "Conv3d_2b_1x1", (1, 1, 1))
    x = unit(x, "Conv3d_2c_3x3", (3, 3, 3))
    x = i3d_ref.maxpool_same(x, (1, 3, 3), (1, 2, 2))
    for name, spec in i3d_ref.MIXED:
        if name.startswith("MaxPool"):
            x = i3d_ref.maxpool_same(x, *spec)
            continue
        b0 = unit(x, name + "/Branch_0/Conv3d_0a_1x1", (1, 1, 1))
        b1 = unit(unit(x, name + "/Branch_1/Conv3d_0a_1x1", (1, 1, 1)), name + "/Branch_1/Conv3d_0b_3x3", (3, 3, 3))
        b2 = unit(unit(x, name + "/Branch_2/Conv3d_0a_1x1", (1, 1, 1)), name + "/Branch_2/" + i3d_ref.b2_3x3_name(name), (3, 3, 3))
        b3 = unit(i3d_ref.maxpool_same(x, (3, 3, 3), (1, 1, 1)), name + "/Branch_3/Conv3d_0b_1x1", (1, 1, 1))
        x = torch.cat([b0, b1, b2, b3], 1)
    wfc = np.abs(rng.standard_normal((x.shape[1], num_classes))).astype(np.float32) * np.float32(0.002)
    wfc[:, label] += np.float32(0.02)
    W[P + "Logits/Conv3d_0c_1x1/conv_3d/w"] = wfc.reshape(1, 1, 1, x.shape[1], num_classes)
    W[P + "Logits/Conv3d_0c_1x1/conv_3d/b"] = np.zeros(num_classes, np.float32)
    return W


def coherent_videoresnet_weights(W_base, x_ncdhw, arch, label=233, seed=5):
    """Well-conditioned VideoResNet weights (same idea as coherent_i3d_weights) from a base weight dict of the right shapes
    (torchvision state_dict names): every convolution weight becomes |w|, BatchNorm gamma = 1 with a small random beta and running
    statistics calibrated on the fixture clip (one oracle forward pass in which every BatchNorm records the mean / variance of its
    own input), fc >= 0 favouring ``label``.  x_ncdhw: the normalised clip [1,3,T,H,W] (float64)."""
    from . import videoresnet_ref as vr
    rng = np.random.default_rng(seed)
    W = {}
    for k, v in W_base.items():
        v = np.asarray(v, np.float32)
        if v.ndim == 5:
            W[k] = np.abs(v)
        elif k.endswith(".weight") and v.ndim == 1:
            W[k] = np.ones_like(v)                                   # BatchNorm gamma
        elif k.endswith(".bias") and not k.startswith("fc."):
            W[k] = (rng.standard_normal(v.shape) * 0.3).astype(np.float32)   # BatchNorm beta
        else:
            W[k] = v.copy()
    Wt = {k: torch.from_numpy(v).double() for k, v in W.items()}
    orig_bn = vr.bn

    def calibrating_bn(x, Wd, pre):
        Wd[pre + ".running_mean"] = x.mean(dim=(0, 2, 3, 4))
        Wd[pre + ".running_var"] = x.var(dim=(0, 2, 3, 4), unbiased=False).clamp_min(1e-6)
        return orig_bn(x, Wd, pre)

    vr.bn = calibrating_bn
    try:
        with torch.no_grad():
            vr.videoresnet_logits(x_ncdhw.double(), Wt, arch)
    finally:
        vr.bn = orig_bn
    for k in W:
        if k.endswith("running_mean") or k.endswith("running_var"):
            W[k] = Wt[k].numpy().astype(np.float32)
    C, F_ = W["fc.weight"].shape
    wfc = np.abs(rng.standard_normal((C, F_))).astype(np.float32) * np.float32(0.002)
    wfc[label] += np.float32(0.02)
    W["fc.weight"], W["fc.bias"] = wfc, np.zeros(C, np.float32)
    return W

"""torchvision-0.5.0 VideoResNet (r2plus1d_18 / r3d_18 / mc3_18) layer tables and seeded synthetic weights.

The reference loads pretrained weights through torchvision (utils_cv/action_recognition/model.py:421); neither
torchvision nor the checkpoints are available here, so benchmarks and tests use seeded synthetic weights under the
torchvision ``state_dict`` names -- a real ``state_dict`` converted to ``{name: ndarray}`` drops in.
"""
import numpy as np

DEFAULT_MEAN = (0.43216, 0.394666, 0.37645)      # dataset.py:28
DEFAULT_STD = (0.22803, 0.22145, 0.216989)       # dataset.py:29
ARCHS = ("r2plus1d_18", "r3d_18", "mc3_18")
PLANES = (64, 128, 256, 512)


def midplanes(inplanes, planes):
    return (inplanes * planes * 27) // (inplanes * 9 + 3 * planes)


def _kind(arch, layer):
    if arch == "r2plus1d_18":
        return "2plus1d"
    return "3d" if (arch == "r3d_18" or layer == 1) else "notemporal"


def conv_table(arch):
    """[(weight prefix, cout, cin, (kt,kh,kw), bn prefix)] in forward order"""
    assert arch in ARCHS, arch
    t = [("stem.0", 45, 3, (1, 7, 7), "stem.1"), ("stem.3", 64, 45, (3, 1, 1), "stem.4")] if arch == "r2plus1d_18" else \
        [("stem.0", 64, 3, (3, 7, 7), "stem.1")]
    inpl = 64
    for li, planes in enumerate(PLANES, start=1):
        kind = _kind(arch, li)
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


def synthetic_weights(arch, seed=42, num_classes=400):
    rng = np.random.default_rng(seed)
    W = {}
    for pre, co, ci, k, bnp in conv_table(arch):
        fan_in = ci * k[0] * k[1] * k[2]
        W[pre + ".weight"] = (rng.standard_normal((co, ci, *k), dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in)))
        W[bnp + ".weight"] = rng.uniform(0.8, 1.2, co).astype(np.float32)
        W[bnp + ".bias"] = (rng.standard_normal(co, dtype=np.float32) * 0.1).astype(np.float32)
        W[bnp + ".running_mean"] = (rng.standard_normal(co, dtype=np.float32) * 0.1).astype(np.float32)
        W[bnp + ".running_var"] = rng.uniform(0.5, 1.5, co).astype(np.float32)
    W["fc.weight"] = (rng.standard_normal((num_classes, 512), dtype=np.float32) * np.float32(0.01))   # keeps the synthetic logits O(5)
    W["fc.bias"] = (rng.standard_normal(num_classes, dtype=np.float32) * 0.1).astype(np.float32)
    return W


def synthetic_clip(B, T=16, H=112, W=112, seed=1234):
    """normalised fp32 clip [B,T,H,W,3] channels-last ((u8/255 - mean)/std, dataset.py transforms)"""
    u8 = np.random.default_rng(seed).integers(0, 256, (B, T, H, W, 3)).astype(np.float32)
    return ((u8 / 255.0 - np.array(DEFAULT_MEAN, np.float32)) / np.array(DEFAULT_STD, np.float32)).astype(np.float32)


def load_weights(path, arch=None):
    """Victim weights for FlickerVideoResNet as ``{state_dict name: float32 ndarray}``.

    ``.pth`` / ``.pt``: a torchvision ``state_dict`` as ``torch.save`` writes it -- what the reference obtains through
    ``torchvision.models.video.<arch>(pretrained=True)`` (utils_cv/action_recognition/model.py:421; torchvision 0.5.0 files such as
    ``r2plus1d_18-91a641e6.pth``) -- or a checkpoint dict holding one under ``state_dict`` / ``model``; a ``module.`` prefix
    (``nn.DataParallel``, model.py:576-578) is stripped and ``num_batches_tracked`` counters are dropped.  ``.npz``: the same names.
    With ``arch`` the names and shapes are checked against that architecture's layer table."""
    if str(path).endswith(".npz"):
        W = {k: np.asarray(v, dtype=np.float32) for k, v in np.load(path).items()}
    else:
        import torch
        sd = torch.load(path, map_location="cpu", weights_only=True)
        for key in ("state_dict", "model"):
            if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
                sd = sd[key]
        W = {}
        for k, v in sd.items():
            if k.endswith("num_batches_tracked") or not hasattr(v, "numpy"):
                continue
            W[k[len("module."):] if k.startswith("module.") else k] = v.detach().to(torch.float32).numpy()
    if arch is not None:
        want = {}
        for pre, co, ci, k, bnp in conv_table(arch):
            want[pre + ".weight"] = (co, ci, *k)
            for s in (".weight", ".bias", ".running_mean", ".running_var"):
                want[bnp + s] = (co,)
        want["fc.weight"], want["fc.bias"] = (W.get("fc.bias", np.zeros(400)).shape[0], 512), W.get("fc.bias", np.zeros(400)).shape
        missing = sorted(set(want) - set(W))
        if missing:
            raise KeyError(f"{path}: not a {arch} state_dict, missing {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        bad = [k for k, shp in want.items() if tuple(W[k].shape) != tuple(shp)]
        if bad:
            raise ValueError(f"{path}: shape mismatch for {arch}: {bad[0]} is {W[bad[0]].shape}, expected {want[bad[0]]}")
    return W

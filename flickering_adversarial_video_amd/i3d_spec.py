"""InceptionI3d layer table (reference i3d.py:144-479) as data, and seeded synthetic weights.

The checkpoint (`data/checkpoints/rgb_imagenet/model.ckpt`) is not in the reference tree, so
benchmarks and tests use seeded synthetic weights with the checkpoint's variable names
(kinetics_i3d_utils.py:41-62); a real checkpoint converted to a ``{name: ndarray}`` dict drops in.
"""
import numpy as np

PREFIX = "RGB/inception_i3d/"
NUM_CLASSES = 400

# block -> (b0, b1a, b1b, b2a, b2b, b3) output channels, i3d.py:194-455
MIXED_CHANNELS = {
    "Mixed_3b": (64, 96, 128, 16, 32, 32),
    "Mixed_3c": (128, 128, 192, 32, 96, 64),
    "Mixed_4b": (192, 96, 208, 16, 48, 64),
    "Mixed_4c": (160, 112, 224, 24, 64, 64),
    "Mixed_4d": (128, 128, 256, 24, 64, 64),
    "Mixed_4e": (112, 144, 288, 32, 64, 64),
    "Mixed_4f": (256, 160, 320, 32, 128, 128),
    "Mixed_5b": (256, 160, 320, 32, 128, 128),
    "Mixed_5c": (384, 192, 384, 48, 128, 128),
}
MIXED_ORDER = ["Mixed_3b", "Mixed_3c", "Mixed_4b", "Mixed_4c", "Mixed_4d", "Mixed_4e", "Mixed_4f",
               "Mixed_5b", "Mixed_5c"]


def branch2_3x3_name(block):
    # the reference names this unit Conv3d_0a_3x3 in Mixed_5b only (i3d.py:418)
    return "Conv3d_0a_3x3" if block == "Mixed_5b" else "Conv3d_0b_3x3"


def conv_units():
    """[(unit path, (kt,kh,kw), cin, cout)] in forward order, incl. the logits conv (bias, no BN)."""
    units = [("Conv3d_1a_7x7", (7, 7, 7), 3, 64),
             ("Conv3d_2b_1x1", (1, 1, 1), 64, 64),
             ("Conv3d_2c_3x3", (3, 3, 3), 64, 192)]
    cin = 192
    for blk in MIXED_ORDER:
        c0, c1a, c1b, c2a, c2b, c3 = MIXED_CHANNELS[blk]
        units += [(f"{blk}/Branch_0/Conv3d_0a_1x1", (1, 1, 1), cin, c0),
                  (f"{blk}/Branch_1/Conv3d_0a_1x1", (1, 1, 1), cin, c1a),
                  (f"{blk}/Branch_1/Conv3d_0b_3x3", (3, 3, 3), c1a, c1b),
                  (f"{blk}/Branch_2/Conv3d_0a_1x1", (1, 1, 1), cin, c2a),
                  (f"{blk}/Branch_2/{branch2_3x3_name(blk)}", (3, 3, 3), c2a, c2b),
                  (f"{blk}/Branch_3/Conv3d_0b_1x1", (1, 1, 1), cin, c3)]
        cin = c0 + c1b + c2b + c3
    units.append(("Logits/Conv3d_0c_1x1", (1, 1, 1), 1024, NUM_CLASSES))
    return units


def synthetic_i3d_weights(seed=42):
    """Seeded He-normal conv weights; BN moving_mean~N(0,.1), moving_variance~U(.5,1.5), beta~N(0,.1)
     so activations neither die nor explode.  Shapes follow the TF checkpoint."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, k, cin, cout in conv_units():
        fan_in = k[0] * k[1] * k[2] * cin
        w = rng.standard_normal((*k, cin, cout), dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        W[PREFIX + name + "/conv_3d/w"] = w
        if name.startswith("Logits"):
            W[PREFIX + name + "/conv_3d/b"] = (rng.standard_normal(cout, dtype=np.float32) * 0.1).astype(np.float32)
            W[PREFIX + name + "/conv_3d/w"] = (w * np.float32(0.5)).astype(np.float32)
        else:
            shp = (1, 1, 1, 1, cout)
            W[PREFIX + name + "/batch_norm/beta"] = (rng.standard_normal(shp, dtype=np.float32) * 0.1).astype(np.float32)
            W[PREFIX + name + "/batch_norm/moving_mean"] = (rng.standard_normal(shp, dtype=np.float32) * 0.1).astype(np.float32)
            W[PREFIX + name + "/batch_norm/moving_variance"] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
    return W


def synthetic_clip_u8(B, T, H=224, W=224, seed=1234):
    """uint8 clips as the TFRecord path delivers them (pre_process_rgb_flow.py:226-234): x = u8/128 - 1."""
    return np.random.default_rng(seed).integers(0, 256, (B, T, H, W, 3), dtype=np.uint8)


def load_i3d_weights(model_cfg):
    """Weights for FlickerI3D from the MODEL section of run_config.yml: ``WEIGHTS_NPZ`` ({variable name: array} archive) if
    set, else the TF checkpoint ``CKPT_PATH`` (the reference's ``init_model``, kinetics_i3d_utils.py:41-62, read without
    TensorFlow by tf_checkpoint.py) if its ``.index`` file exists, else seeded synthetic weights."""
    import os
    if model_cfg.get("WEIGHTS_NPZ"):
        return dict(np.load(model_cfg["WEIGHTS_NPZ"])), "npz"
    ckpt = model_cfg.get("CKPT_PATH")
    if ckpt and os.path.exists(ckpt + ".index"):
        from .tf_checkpoint import load_i3d_checkpoint
        return load_i3d_checkpoint(ckpt, "RGB"), "tf-checkpoint"
    return synthetic_i3d_weights(42), "synthetic"

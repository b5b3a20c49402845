"""run_config.yml surface (reference utils/kinetics_i3d_utils.py:22-26: ``edict(yaml.load(f))``).

``yaml.load`` without a Loader fails on PyYAML >= 6: ``safe_load`` is used.  Keys are read verbatim;
[new] optional keys default to the reference's behaviour."""
import os

import yaml


class Cfg(dict):
    """attribute access like EasyDict, recursively"""

    def __init__(self, d=()):
        super().__init__()
        for k, v in dict(d).items():
            self[k] = Cfg(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


DEFAULT_MODEL = {"FRAMES": 90, "DTYPE": "bf16", "WEIGHTS_NPZ": ""}


def load_config(yml_path):
    with open(yml_path, "r") as f:
        cfg = Cfg(yaml.safe_load(f))
    cfg.setdefault("MODEL", Cfg())
    for k, v in DEFAULT_MODEL.items():
        cfg.MODEL.setdefault(k, v)
    return cfg


def load_kinetics_classes(label_map_path):
    """class name per line, index = line number (kinetics_i3d_utils.py:68-74)"""
    with open(label_map_path) as f:
        return [x.strip() for x in f]


def label_from_npy_name(path, classes):
    """``rgb_<id>@<class_with_underscores>.npy`` (pre_process_rgb_flow.py:256; loader i3d_adversarial_main_single_video_npy.py:121-124)"""
    name = os.path.splitext(os.path.basename(path))[0]
    if "@" not in name:
        raise ValueError(f"{path}: expected '<id>@<class_with_underscores>.npy'")
    cls = name.split("@", 1)[1].replace("_", " ")
    if cls not in classes:
        raise ValueError(f"{path}: class {cls!r} is not in the label map")
    return cls, classes.index(cls)


RESULT_KEYS = ("correct_cls_prob", "correct_cls", "correct_cls_id", "softmax_init", "rgb_sample", "total_loss_l", "adv_loss_l",
               "reg_loss_l", "norm_reg_loss_l", "diff_norm_reg_loss_l", "perturbation", "adv_video", "softmax", "total_steps",
               "beta_0", "beta_1", "beta_2", "beta_3", "fatness", "smoothness")


def result_filename(cls, beta_1, thickness_pct, roughness_pct):
    """i3d_adversarial_main_single_video_npy.py:330-331"""
    return "{}_beta1_{}_th_{:.2f}%_rg_{:.2f}%.pkl".format(cls, beta_1, thickness_pct, roughness_pct)

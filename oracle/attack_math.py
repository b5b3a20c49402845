"""CPU restatement of the flickering-attack mathematics (TEST INFRASTRUCTURE, see oracle/__init__.py).

Two dialects, exactly as the reference has them (all citations relative to the reference tree):

* TF dialect    -- utils/kinetics_i3d_utils.py (I3D path): delta is [T,1,1,3] (flicker) or
                   [T,H,W,3] (dense "L12" baseline), clips are NDHWC in [-1,1].
* torch dialect -- utils_cv/action_recognition/model.py (VideoResNet path): delta is [3,T,1,1]
                   (or [3,T,H,W]), clips are NCDHW, normalised with DEFAULT_MEAN/STD.

Everything is fp32 torch-CPU so that autograd supplies the gradients the HIP kernels compute in
closed form.
"""
import math
import numpy as np
import torch

# utils_cv/action_recognition/dataset.py:28-29
DEFAULT_MEAN = (0.43216, 0.394666, 0.37645)
DEFAULT_STD = (0.22803, 0.22145, 0.216989)
# model.py:72-75 -- scalar clamp bounds shared by all channels
TORCH_MAX_VALUE = float(np.min((1 - np.array(DEFAULT_MEAN)) / DEFAULT_STD))
TORCH_MIN_VALUE = float(np.max((0.0 - np.array(DEFAULT_MEAN)) / DEFAULT_STD))

TF_DELTA_CLIP = 0.4  # kinetics_i3d_utils.py:104-105


# ----------------------------------------------------------------------------------------------
# TF dialect
# ----------------------------------------------------------------------------------------------
def tf_frame_mask(T, ind_start=0, ind_end=None):
    """kinetics_i3d_utils.py:107-113: mask[t] = 1 for ind_start <= t <= ind_end, t < T."""
    ind_end = T if ind_end is None else ind_end
    m = torch.zeros(T)
    lo, hi = max(ind_start, 0), min(ind_end, T - 1)
    m[lo:hi + 1] = 1.0
    return m


def tf_apply(x, delta, adv_flag=1.0, shift_x=0, cyclic_flag=0.0, shift_p=0, cyclic_pert_flag=0.0,
             clip_delta=True, ind_start=0, ind_end=None):
    """x_adv = clip(x' + a * p', -1, 1)   (kinetics_i3d_utils.py:100-142; L12 variant :333-366).

    x: [B,T,H,W,3]; delta: [T,1,1,3] or [T,H,W,3]. ``clip_delta=False`` is the dense L12 class
    (no +-0.4 clip, :336). Rolls follow tf.roll (positive shift moves data to higher indices).
    """
    T = x.shape[1]
    d = torch.clamp(delta, -TF_DELTA_CLIP, TF_DELTA_CLIP) if clip_delta else delta
    p = tf_frame_mask(T, ind_start, ind_end).view(T, 1, 1, 1) * d
    p = cyclic_pert_flag * torch.roll(p, shift_p, 0) + (1 - cyclic_pert_flag) * p
    xc = cyclic_flag * torch.roll(x, shift_x, 1) + (1 - cyclic_flag) * x
    return torch.clamp(xc + adv_flag * p, -1.0, 1.0)


def tf_regularizers(delta):
    """norm / diff / laplacian / thickness / roughness on the RAW variable (kinetics_i3d_utils.py:172-200).

    roll is over axis 0 (time)."""
    r = torch.roll(delta, 1, 0)
    l = torch.roll(delta, -1, 0)
    out = {
        "norm": (delta ** 2).mean() + 1e-12,
        "diff": ((delta - r) ** 2).mean() + 1e-12,
        "lap": ((-2 * delta + r + l) ** 2).mean() + 1e-12,
        "thickness": delta.abs().mean(),
        "roughness": (delta - r).abs().mean(),
    }
    out["thickness_pct"] = out["thickness"] / 2.0 * 100
    out["roughness_pct"] = out["roughness"] / 2.0 * 100
    return out


def tf_l12(delta):
    """kinetics_i3d_L12: sum_t sqrt(mean_{hwc} delta_t^2) + 1e-12 (kinetics_i3d_utils.py:409)."""
    return torch.sqrt((delta ** 2).mean(dim=(1, 2, 3))).sum() + 1e-12


def reg_grads_closed_form(delta, time_axis=0):
    """Closed-form d/d(delta) of norm, diff, lap. Returns three tensors."""
    N = delta.numel()
    r = torch.roll(delta, 1, time_axis)
    l = torch.roll(delta, -1, time_axis)
    d = delta - r                      # d_t = delta_t - delta_{t-1}
    lap = -2 * delta + r + l           # l_t
    g_norm = 2 * delta / N
    g_diff = 2 * (d - torch.roll(d, -1, time_axis)) / N
    g_lap = 2 * (-2 * lap + torch.roll(lap, 1, time_axis) + torch.roll(lap, -1, time_axis)) / N
    return g_norm, g_diff, g_lap


def tf_label_stats(logits, labels):
    """kinetics_i3d_utils.py:152-169. NOTE max_non_label_logits subtracts one_hot (does NOT exclude
    the label) -- reproduced."""
    p = torch.softmax(logits, -1)
    oh = torch.nn.functional.one_hot(labels, logits.shape[-1]).to(logits.dtype)
    idx = labels.view(-1, 1)
    return {
        "softmax": p,
        "label_prob": p.gather(1, idx)[:, 0],
        "label_logits": logits.gather(1, idx)[:, 0],
        "max_non_label_prob": (p - oh).max(-1)[0],
        "max_non_label_logits": (logits - oh).max(-1)[0],
    }


def tf_improve_adversarial_loss(logits, labels, margin=0.05, targeted=False, use_logits=False):
    """kinetics_i3d_utils.py:253-288.  labels = true label (untargeted) or target class (targeted).
    Returns (loss_total, to_min_prob, to_max_prob)."""
    s = tf_label_stats(logits, labels)
    if targeted:
        if use_logits:
            to_min, to_max = s["max_non_label_logits"], s["label_logits"]
            m = torch.log(1.0 + margin * (1.0 / s["label_prob"]))
        else:
            to_min, to_max, m = s["max_non_label_prob"], s["label_prob"], margin
        to_min_prob, to_max_prob = s["max_non_label_prob"], s["label_prob"]
    else:
        if use_logits:
            to_min, to_max = s["label_logits"], s["max_non_label_logits"]
            m = torch.log(1.0 + margin * (1.0 / (0.00001 + s["max_non_label_prob"])))
        else:
            to_min, to_max, m = s["label_prob"], s["max_non_label_prob"], margin
        to_min_prob, to_max_prob = s["label_prob"], s["max_non_label_prob"]
    u = to_min - (to_max - m)
    l2 = u ** 2 / m
    adv = torch.maximum(torch.zeros_like(u), torch.minimum(l2, u))
    return adv.sum(), to_min_prob, to_max_prob


def tf_ce_adversarial_loss(logits, labels, targeted=False):
    """kinetics_i3d_utils.py:290-307."""
    s = tf_label_stats(logits, labels)
    if targeted:
        ce = torch.nn.functional.cross_entropy(logits, labels, reduction="none")
        return ce.mean(), s["max_non_label_prob"], s["label_prob"]
    ce = -torch.log(1 - s["label_prob"] + 1e-6)
    return ce.mean(), s["label_prob"], s["max_non_label_prob"]


def tf_total_loss(adv, delta, beta0, beta1, beta2, beta3):
    """i3d_adversarial_main_single_video_npy.py:56-59: adv + b0*(b1*norm + b2*diff + b3*lap)."""
    r = tf_regularizers(delta)
    reg = beta1 * r["norm"] + beta2 * r["diff"] + beta3 * r["lap"]
    return adv + beta0 * reg, reg


def tf_adam_step(delta, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """TF-1.15 AdamOptimizer (non-Keras): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); eps OUTSIDE the
    bias correction. t is the 1-based step counter of THIS update."""
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return delta - lr_t * m / (torch.sqrt(v) + eps), m, v


def is_adversarial(softmax, labels, targeted=False):
    """i3d_adversarial_main_single_video_npy.py:155-161."""
    am = softmax.argmax(-1)
    return bool((am == labels).all()) if targeted else bool((am != labels).all())


def fooling_counts(adv_logits, clean_logits, labels, targeted=False, target=None, exclude_misclassify=True):
    """kinetics_i3d_utils.py:217-250 / model.py:293-323 -> (miss, valid)."""
    a, c = adv_logits.argmax(-1), clean_logits.argmax(-1)
    miss_cond = (a == target) if targeted else (a != labels)
    if not exclude_misclassify:                                  # kinetics_i3d_utils.py:244-246
        return int(miss_cond.sum()), int(miss_cond.numel())
    valid = c == labels
    return int((miss_cond & valid).sum()), int(valid.sum())


# ----------------------------------------------------------------------------------------------
# torch dialect (model.py:58-330)
# ----------------------------------------------------------------------------------------------
def torch_apply(x, delta, dyn_max_norm, adversarial=True, shift=0, cyclic=False):
    """Perturbation.forward (model.py:80-101). x: [B,3,T,H,W] normalised; delta: [3,T,1,1]|[3,T,H,W]."""
    if not adversarial:
        return x
    dc = delta.clamp(-dyn_max_norm, dyn_max_norm)
    std = torch.tensor(DEFAULT_STD, dtype=x.dtype).view(3, 1, 1, 1)
    dn = dc / std
    if cyclic:
        dn = torch.roll(dn, shift, 1)
    return (x + dn).clamp(TORCH_MIN_VALUE, TORCH_MAX_VALUE)


def torch_flicker_reg(dc, beta_1):
    """Losses.flickering_regularization_loss on the CLAMPED delta, roll on dim 1 (model.py:198-209)."""
    r, l = torch.roll(dc, 1, 1), torch.roll(dc, -1, 1)
    norm = (dc ** 2).mean() + 1e-12
    diff = ((dc - r) ** 2).mean() + 1e-12
    lap = ((-2 * dc + r + l) ** 2).mean() + 1e-12
    return beta_1 * norm + (1 - beta_1) * (diff + lap)


def torch_l12_reg(dc):
    """model.py:211-214."""
    return torch.sqrt((dc ** 2).mean(dim=(0, 2, 3))).sum() + 1e-12


def torch_improve_loss(logits, prob, labels, margin=0.05, use_logits=False):
    """Losses.improve_adversarial_loss, untargeted (model.py:216-250).  True exclusion of the label
    for max-non-label; logits-mode margin uses LABEL prob (:236).  Targeted is non-functional in the
    reference (undefined names, :223-225) -> raise."""
    C = logits.shape[1]
    idx = labels.view(-1, 1)
    label_prob = prob.gather(1, idx)
    non = torch.nn.functional.one_hot(labels, C) == 0
    if use_logits:
        to_min = logits.gather(1, idx)
        to_max = logits.masked_fill(~non, -float("inf")).max(1, keepdim=True)[0]
        m = torch.log(1.0 + margin * (1.0 / (0.00001 + label_prob)))
    else:
        to_min = label_prob
        to_max = prob.masked_fill(~non, -float("inf")).max(1, keepdim=True)[0]
        m = torch.full_like(to_min, margin)
    u = to_min - (to_max - m)
    return torch.maximum(torch.zeros_like(u), torch.minimum(u ** 2 / m, u)).sum()


def torch_ce_loss(prob, labels, targeted=False, target_class=None):
    """Losses.ce_adversarial_loss (model.py:177-196)."""
    if targeted:
        return (-torch.log(prob[:, target_class] + 1e-6)).mean()
    return (-torch.log(1 - prob.gather(1, labels.view(-1, 1)) + 1e-6)).mean()


def torch_losses(labels, logits, prob, dc, beta_1=0.5, lambda_=1.0, margin=0.05, improve_loss=True,
                 use_logits=False, attack_type="flickering", targeted=False, target_class=None):
    """Losses.__call__ -> [loss, adv, reg] (model.py:169-175)."""
    reg = torch_flicker_reg(dc, beta_1) if attack_type == "flickering" else torch_l12_reg(dc)
    if improve_loss:
        if targeted:
            raise NotImplementedError("reference targeted improve-loss is non-functional (model.py:223-225)")
        adv = torch_improve_loss(logits, prob, labels, margin, use_logits)
    else:
        adv = torch_ce_loss(prob, labels, targeted, target_class)
    return adv + lambda_ * reg, adv, reg


def torch_adam_step(delta, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch-1.4 Adam: denom = sqrt(v)/sqrt(1-b2^t) + eps; step = lr/(1-b1^t)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = torch.sqrt(v) / math.sqrt(1 - b2 ** t) + eps
    return delta - (lr / (1 - b1 ** t)) * m / denom, m, v


def torch_metrics(delta):
    """Adversarial_metrics.adversarial_metric / Perturbation.metric_calc (model.py:114-119,325-330)."""
    return delta.abs().mean() * 100.0, (torch.roll(delta, 1, 1) - delta).abs().mean() * 100.0

"""I3D flickering-attack engine: the host-side mirror of the reference's ``kinetics_i3d`` object
(utils/kinetics_i3d_utils.py:76-307) on top of libflicker_hip.so.

The reference drives a TF-1 graph with 3-4 ``sess.run`` calls per iteration (3 forwards + 1 backward,
each re-feeding the 54 MB clip: i3d_adversarial_main_single_video_npy.py:213-217,305-308).  Here ONE
device pass per iteration does  apply(delta) -> I3D forward -> loss -> backward-to-delta -> Adam  and
returns every scalar the scripts fetch; nothing syncs with the host unless a value is read.

Data-parallel (class-generalisation / universal attacks): every rank holds the frozen packed weights and
an identical replica of (delta, Adam m, v, t); the only exchange is ONE sum all-reduce of the (T x 3)
adversarial gradient plus the loss scalars; the regulariser gradient is added once, after it.
"""
import numpy as np
import torch

from . import ops, parallel
from ._lib import FLK_NET_I3D

import os

_APPLY_INSIDE = os.environ.get("FLK_APPLY_INSIDE", "1") != "0"
NUM_CLASSES = 400          # kinetics_i3d_utils.py:19
SAMPLE_VIDEO_FRAMES = 90   # kinetics_i3d_utils.py:12 (reference default; the benchmark shape is 64)
IMAGE_SIZE = 224           # kinetics_i3d_utils.py:9


class StepResult(dict):
    """Scalars of one attack iteration as device tensors; ``.host()`` fetches them (one sync).

    The kernels write straight into one of ``RESULT_SLOTS`` result slots of the engine, so the tensors stay valid for
    the next ``RESULT_SLOTS - 1`` iterations (copy them to keep them longer).  Quantities that are pure functions of
    the stored ones (``is_adversarial``, ``argmax``, ``total_loss``, the ``*_relative`` percentages) are computed on
    first access -- an iteration that nobody inspects launches no bookkeeping kernels."""

    _DERIVED = ("argmax", "is_adversarial", "total_loss", "loss", "thickness_relative", "roughness_relative", "thickness", "roughness")

    def __missing__(self, key):
        if key == "argmax":
            v = self["_argmax_f"].to(torch.int64)
        elif key == "is_adversarial":
            v = (self["argmax"] == self["_labels"]).all() if self["_targeted"] else (self["argmax"] != self["_labels"]).all()
        elif key in ("total_loss", "loss") and "reg_loss" in self:      # "loss": the torch engine's name (model.py:1079)
            v = self["adv_loss"] + self["_reg_weight"] * self["reg_loss"]
        elif key == "thickness" and "_thickness" in self:                # torch engine: percentages (model.py:326-329)
            v = self["_thickness"] * 100
        elif key == "roughness" and "_roughness" in self:
            v = self["_roughness"] * 100
        elif key == "thickness_relative" and "thickness" in self:
            v = self["thickness"] / 2 * 100
        elif key == "roughness_relative" and "roughness" in self:
            v = self["roughness"] / 2 * 100
        else:
            raise KeyError(key)
        self[key] = v
        return v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def materialize(self):
        for k in self._DERIVED:
            self.get(k)
        return self

    def host(self):
        self.materialize()
        return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in self.items() if not k.startswith("_")}


RESULT_SLOTS = 4


class FlickerI3D:
    """Attack engine for InceptionI3d.

    ctor kwargs follow kinetics_i3d.__init__ (kinetics_i3d_utils.py:79-80) where they still mean something:
    ``batch_size``, ``cyclic_flag_default_c``, ``cyclic_pert_flag_default_c``, ``default_adv_flag_c``;
    ``weights`` replaces ``ckpt_path`` (a {checkpoint variable name: ndarray} dict, see i3d_spec.py) and
    ``frames`` / ``dtype`` are new optional knobs (defaults reproduce the reference: 90 frames).
    """

    def __init__(self, weights, batch_size=1, frames=SAMPLE_VIDEO_FRAMES, dtype="bf16", device=0, dense_delta=False,
                 cyclic_flag_default_c=0.0, cyclic_pert_flag_default_c=0.0, default_adv_flag_c=1.0, process_group=None,
                 seed=0, kinetics_classes=None, per_clip_delta=False):
        if not torch.cuda.is_available():
            raise RuntimeError("FlickerI3D needs an MI355X (HIP) device; there is no CPU fallback")
        torch.cuda.set_device(device)
        self.B, self.T, self.H, self.W = batch_size, frames, IMAGE_SIZE, IMAGE_SIZE
        self.dtype = dtype
        self.dense = dense_delta
        # per_clip_delta: B INDEPENDENT single-video attacks in one batch (the reference runs them one after another,
        # i3d_adversarial_main_single_video_npy.py:103-337): eps_rgb, the Adam state, the step counter and the "still attacking" flag
        # are per clip; clip b follows exactly the trajectory it would follow alone (bit for bit in fp32).  Replicas only: no collective.
        self.per_clip = bool(per_clip_delta)
        if self.per_clip and (dense_delta or parallel.world_size(process_group) > 1 or cyclic_flag_default_c or cyclic_pert_flag_default_c):
            raise ValueError("per_clip_delta: flicker perturbation, one rank, no cyclic rolls")
        self.cyclic_flag, self.cyclic_pert_flag, self.adv_flag = cyclic_flag_default_c, cyclic_pert_flag_default_c, default_adv_flag_c
        self.pg = process_group
        self.world = parallel.world_size(process_group)
        self.net = ops.Net(FLK_NET_I3D, dtype, self.B, self.T, self.H, self.W, weights, device)
        dev = torch.device("cuda", device)
        dshape = (self.T, self.H, self.W, 3) if dense_delta else (self.B, self.T, 3) if self.per_clip else (self.T, 3)
        # eps_rgb: zeros [T,1,1,3] (kinetics_i3d_utils.py:100); dense L12 variant: 1e-8 (:333)
        self.eps_rgb = torch.full(dshape, 1e-8 if dense_delta else 0.0, dtype=torch.float32, device=dev)
        self.adam_m, self.adam_v = torch.zeros_like(self.eps_rgb), torch.zeros_like(self.eps_rgb)
        self.adam_t = 0
        if self.per_clip:
            self.adam_steps = torch.zeros(self.B, dtype=torch.int32, device=dev)      # Adam step counters, advanced by the kernel
            self.active = torch.ones(self.B, dtype=torch.int32, device=dev)           # 0: this clip's attack has ended (frozen)
        self._xs2d = torch.empty((self.B, self.T // 2, self.H // 2, self.W // 2, 32), dtype=self.net_torch_dtype, device=dev)
        self._gx = torch.empty_like(self._xs2d)
        self._logits = torch.empty((self.B, NUM_CLASSES), dtype=torch.float32, device=dev)
        # [g_adv (T*3) | adv loss sum | sum to_min_prob | sum to_max_prob]: ONE all-reduce payload
        self._red = torch.zeros(parallel.payload_size(self.T), dtype=torch.float32, device=dev)
        self._scratch = torch.empty(max(1, ops.load().flk_perturb_grad_scratch_bytes(self.B, self.T, self.H, self.W) // 4,
                                        ops.load().flk_stem_delta_grad_scratch_bytes(self.B, self.T, self.H) // 4),
                                    dtype=torch.float32, device=dev)
        # flicker perturbation in bf16: the stem's data-gradient and the (b,h,w) reduction run as ONE MFMA kernel (csrc/stem_grad.hip)
        self.fused_delta_grad = (not dense_delta) and self.net.has_backward_delta
        # ... and the perturbation reaches the stem in fp32 (centred clip + position-class bias) instead of being rounded with the bf16 input
        self.exact_delta_forward = (not dense_delta) and self.net.has_forward_flicker
        self._scalars = torch.empty(8, dtype=torch.float32, device=dev)
        # result slots: [payload | softmax | per-clip table | scalars] written by the kernels, rotated per iteration
        self._slots = [dict(payload=torch.zeros(parallel.payload_size(self.T), dtype=torch.float32, device=dev),
                            sm=torch.empty((self.B, NUM_CLASSES), dtype=torch.float32, device=dev),
                            pc=torch.empty((self.B, 4), dtype=torch.float32, device=dev),
                            scalars=torch.zeros((self.B, 8) if self.per_clip else 8, dtype=torch.float32, device=dev),
                            gclip=torch.zeros((self.B, self.T, 3), dtype=torch.float32, device=dev) if self.per_clip else None)
                       for _ in range(RESULT_SLOTS)]
        self._dl = torch.empty((self.B, NUM_CLASSES), dtype=torch.float32, device=dev)
        self._it = 0
        self._rng = np.random.default_rng(seed)
        # class names, index = label id (kinetics_i3d_utils.py:68-74 reads data/label_map.txt in the constructor; here the caller passes
        # the list or a path -- config.load_kinetics_classes -- because the engine has no file dependencies)
        if isinstance(kinetics_classes, (str, bytes)):
            from .config import load_kinetics_classes
            kinetics_classes = load_kinetics_classes(kinetics_classes)
        self.kinetics_classes = list(kinetics_classes) if kinetics_classes is not None else [str(i) for i in range(NUM_CLASSES)]

    @property
    def net_torch_dtype(self):
        return torch.bfloat16 if self.dtype in ("bf16", torch.bfloat16) else torch.float32

    # ---- state -------------------------------------------------------------------------------------
    def reset_clip(self, b, delta=None):
        """per-clip mode: slot b starts a NEW video -- its perturbation, Adam moments and step counter are re-initialised
        (i3d_adversarial_main_single_video_npy.py:205-206) and it is active again; the other clips are untouched"""
        if not self.per_clip:
            raise ValueError("reset_clip: the engine was not built with per_clip_delta=True")
        if delta is None:
            self.eps_rgb[b].zero_()
        else:
            self.eps_rgb[b].copy_(torch.as_tensor(delta, dtype=torch.float32).reshape(self.T, 3))
        self.adam_m[b].zero_()
        self.adam_v[b].zero_()
        self.adam_steps[b] = 0
        self.active[b] = 1

    def reset_perturbation(self, delta=None):
        """sess.run(eps_rgb.initializer) + Adam slot re-init (i3d_adversarial_main_single_video_npy.py:205-206)"""
        if self.per_clip:
            self.adam_steps.zero_()
            self.active.fill_(1)
        if delta is None:
            self.eps_rgb.fill_(1e-8 if self.dense else 0.0)
        else:
            self.eps_rgb.copy_(torch.as_tensor(delta, dtype=torch.float32).reshape(self.eps_rgb.shape))
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.adam_t = 0

    @property
    def perturbation(self):
        """[T,1,1,3] like the reference variable ([B,T,1,1,3] in per-clip mode: one such variable per clip)"""
        if self.per_clip:
            return self.eps_rgb.view(self.B, self.T, 1, 1, 3)
        return self.eps_rgb if self.dense else self.eps_rgb.view(self.T, 1, 1, 3)

    def _apply_args(self, x, adv_flag, cyclic, cyclic_pert):
        sx = int(self._rng.integers(0, self.T)) if cyclic else 0          # one shift per step for the whole batch
        sp = int(self._rng.integers(0, self.T)) if cyclic_pert else 0     # (kinetics_i3d_utils.py:115,130)
        self._last_x, self._last_shifts = x, (sx, sp)
        return ops.make_apply_args(x, self.eps_rgb, dialect="tf", dclip=0.0 if self.dense else 0.4, adv_flag=adv_flag,
                                   shift_x=sx, shift_p=sp, fold_t=ops.I3D_FOLD, center=self.exact_delta_forward)

    def _forward(self, a):
        """apply + network forward into self._logits: one call -- the plan applies each half of the batch on the stream that half's
        stem convolution runs on, so the second half's apply overlaps the first half's stem (FLK_APPLY_INSIDE=0: apply first, then
        forward -- the round-2 order; same results)"""
        if _APPLY_INSIDE:
            return self.net.forward_apply(a, self._xs2d, self._logits)
        ops.perturb_apply_s2d(a, self.dtype, self._xs2d)
        if a.center:
            return self.net.forward_flicker(self._xs2d, a, self._logits)
        return self.net.forward(self._xs2d, self._logits)

    def _check_x(self, x):
        if tuple(x.shape) != (self.B, self.T, self.H, self.W, 3) or x.dtype not in (torch.uint8, torch.float32) or not x.is_cuda:
            raise ValueError(f"clip must be a CUDA uint8/float32 tensor of shape {(self.B, self.T, self.H, self.W, 3)}, "
                             f"got {tuple(x.shape)} {x.dtype} {x.device}")
        return x.contiguous()

    def autotune(self, x):
        """Pick the fastest launch layout for every convolution on this clip's activations (like cudnn.benchmark: speed
        only -- the arithmetic per output does not depend on the layout).  Takes a fraction of a second; call once."""
        x = self._check_x(x)
        ops.perturb_apply_s2d(self._apply_args(x, 0.0, 0, 0), self.dtype, self._xs2d)
        dl = torch.randn_like(self._logits) * 1e-3
        self.net.autotune(self._xs2d, self._logits, dl, self._gx)

    # ---- inference ---------------------------------------------------------------------------------
    def logits(self, x, adv_flag=None, cyclic=None):
        x = self._check_x(x)
        a = self._apply_args(x, self.adv_flag if adv_flag is None else adv_flag, self.cyclic_flag if cyclic is None else cyclic, 0)
        return self._forward(a)

    def __call__(self, inputs, adv_flag=0, cyclic=None):
        """softmax of the (clean by default) clip: kinetics_i3d.__call__ (kinetics_i3d_utils.py:210-212)"""
        return torch.softmax(self.logits(inputs, adv_flag, cyclic), -1)

    predict = __call__                                   # predict(x, adv_flag)

    def get_kinetics_classes(self):
        """kinetics_i3d.get_kinetics_classes (kinetics_i3d_utils.py:214-215)"""
        return self.kinetics_classes

    # The reference's loss methods return the TF loss node that the script adds to its train_op
    # (i3d_adversarial_main_single_video_npy.py:46-59).  The eager engine has no graph: the same calls return the loss CHOICE as
    # keyword arguments of step(), so a script reads  loss = k_i3d.improve_adversarial_loss(margin, targeted, logits);
    # res = k_i3d.step(x, labels, lr=..., beta0=..., **loss).
    @staticmethod
    def improve_adversarial_loss(margin=0.05, targeted=False, logits=False):
        """kinetics_i3d.improve_adversarial_loss(margin, targeted, logits) (kinetics_i3d_utils.py:253-288)"""
        return dict(improve_loss=True, margin=float(margin), targeted=bool(targeted), use_logits=bool(logits))

    @staticmethod
    def ce_adversarial_loss(targeted=False):
        """kinetics_i3d.ce_adversarial_loss(targeted) (kinetics_i3d_utils.py:290-307)"""
        return dict(improve_loss=False, targeted=bool(targeted))

    # ---- one attack iteration ----------------------------------------------------------------------
    def step(self, x, labels, lr=1e-3, beta0=1.0, beta1=0.5, beta2=0.5, beta3=0.5, margin=0.05, targeted=False,
             use_logits=False, improve_loss=True, cyclic=None, cyclic_pert=None, update=True):
        """apply -> forward -> loss -> backward-to-delta -> (all-reduce) -> Adam, in one device pass.

        loss = adv + beta0*(beta1*norm + beta2*diff + beta3*lap)  (i3d_adversarial_main_single_video_npy.py:56-59);
        labels = true labels (untargeted) or the target class per clip (targeted).  Returned scalars are the
        PRE-update values, as the reference fetches them together with train_op."""
        if self.dense:
            return self._step_dense(x, labels, lr, beta0, beta1, margin, targeted, use_logits, improve_loss, cyclic, update)
        if self.per_clip:
            return self._step_per_clip(x, labels, lr, beta0, beta1, beta2, beta3, margin, targeted, use_logits, improve_loss, update)
        x = self._check_x(x)
        cyclic = self.cyclic_flag if cyclic is None else cyclic
        cyclic_pert = self.cyclic_pert_flag if cyclic_pert is None else cyclic_pert
        a = self._apply_args(x, 1.0, cyclic, cyclic_pert)
        slot = self._slots[self._it % RESULT_SLOTS]
        self._it += 1
        red, sm, pc = slot["payload"], slot["sm"], slot["pc"]
        self._red = red
        # (labels are validated BEFORE the mask pre-pass starts: a raise between prepare_backward_delta and backward_delta would leave
        #  the pre-pass running on the net's mask stream, unjoined, while torch frees or reuses the clip and the scratch buffer)
        ops.check_labels(labels, self.B, self.net.num_classes)
        if self.fused_delta_grad:
            self.net.prepare_backward_delta(a, self._scratch)      # the clip mask of this iteration's backward pass: beside the stem
        self._forward(a)
        gbatch = self.B * self.world
        ops.softmax_adv_loss(self._logits, labels, dialect="tf", improve_loss=improve_loss, use_logits=use_logits,
                             targeted=targeted, margin=margin, mean_scale=1.0 / gbatch, out=(sm, self._dl, pc))
        n = self.T * 3
        if self.fused_delta_grad:
            self.net.backward_delta(self._dl, a, red[:n], self._scratch)
        else:
            self.net.backward(self._dl, self._gx)
            ops.perturb_grad_reduce(a, self._gx, red[:n].view(self.T, 3), self._scratch)
        ops.pack_batch_sums(pc, 1.0 / gbatch, red[n:])                 # [sum adv | mean label prob | mean max-other prob]
        parallel.allreduce_sum_(red, self.pg)                          # RCCL over xGMI: (T*3+3) floats
        p_lab, p_non = red[n + 1], red[n + 2]
        # to_min / to_max probabilities swap roles for targeted attacks (kinetics_i3d_utils.py:265-278)
        res = StepResult(adv_loss=red[n], prob_to_min=p_non if targeted else p_lab, prob_to_max=p_lab if targeted else p_non,
                         softmax=sm, label_prob=pc[:, 1], _argmax_f=pc[:, 3], _labels=labels, _targeted=bool(targeted))
        if update:
            self.adam_t += 1
            sc = slot["scalars"]
            ops.perturb_reg_adam(red[:n], self.eps_rgb, self.adam_m, self.adam_v, self.adam_t, dialect="tf", beta0=beta0,
                                 beta1=beta1, beta2=beta2, beta3=beta3, lr=lr, scalars=sc)
            res.update(reg_loss=sc[0], norm_reg=sc[1], diff_norm_reg=sc[2], laplacian_norm_reg=sc[3], thickness=sc[4],
                       roughness=sc[5], pert_max=sc[6], pert_min=sc[7], _reg_weight=beta0)
        self.last_result = res
        return res

    def _step_per_clip(self, x, labels, lr, beta0, beta1, beta2, beta3, margin, targeted, use_logits, improve_loss, update):
        """one iteration of B independent single-video attacks: every quantity of step() per clip ([B] / [B,...] tensors).  The loss of
        clip b is its own adversarial term + the regulariser of ITS perturbation (a batch of one in the reference: the CE variants'
        mean is over that one clip); d(loss_b)/d(delta_b) is the per-clip slice of the delta-gradient reduction.  Clips whose
        ``active`` flag is 0 still run through the network (the batch is dense) but are not updated."""
        x = self._check_x(x)
        a = self._apply_args(x, 1.0, 0, 0)
        slot = self._slots[self._it % RESULT_SLOTS]
        self._it += 1
        sm, pc, g = slot["sm"], slot["pc"], slot["gclip"]
        ops.check_labels(labels, self.B, self.net.num_classes)
        if self.fused_delta_grad:
            self.net.prepare_backward_delta(a, self._scratch)
        self._forward(a)
        ops.softmax_adv_loss(self._logits, labels, dialect="tf", improve_loss=improve_loss, use_logits=use_logits,
                             targeted=targeted, margin=margin, mean_scale=1.0, out=(sm, self._dl, pc))
        if self.fused_delta_grad:
            self.net.backward_delta(self._dl, a, g, self._scratch)
        else:
            self.net.backward(self._dl, self._gx)
            ops.perturb_grad_reduce(a, self._gx, g, self._scratch)
        self._gclip = g
        p_lab, p_non = pc[:, 1], pc[:, 2]
        res = StepResult(adv_loss=pc[:, 0], prob_to_min=p_non if targeted else p_lab, prob_to_max=p_lab if targeted else p_non,
                         softmax=sm, label_prob=pc[:, 1], _argmax_f=pc[:, 3], _labels=labels, _targeted=bool(targeted))
        am = pc[:, 3].to(torch.int64)
        res["argmax"] = am
        res["is_adversarial"] = (am == labels) if targeted else (am != labels)          # per clip
        if update:
            sc = slot["scalars"]
            ops.perturb_reg_adam_batched(g, self.eps_rgb, self.adam_m, self.adam_v, self.adam_steps, self.active, dialect="tf", beta0=beta0,
                                         beta1=beta1, beta2=beta2, beta3=beta3, lr=lr, scalars=sc)
            res.update(reg_loss=sc[:, 0], norm_reg=sc[:, 1], diff_norm_reg=sc[:, 2], laplacian_norm_reg=sc[:, 3], thickness=sc[:, 4],
                       roughness=sc[:, 5], pert_max=sc[:, 6], pert_min=sc[:, 7], _reg_weight=beta0)
        self.last_result = res
        return res

    def _step_dense(self, x, labels, lr, beta0, beta1, margin, targeted, use_logits, improve_loss, cyclic, update):
        """kinetics_i3d_L12 (kinetics_i3d_utils.py:308-521) with loss = adv + beta0 * (beta1 * L12) (i3d_adversarial_main_universal.py:
        129-133): dense delta [T,224,224,3]; the all-reduce payload is the full dense gradient (38.5 MB at T=64: bandwidth-bound)."""
        x = self._check_x(x)
        a = self._apply_args(x, 1.0, self.cyclic_flag if cyclic is None else cyclic, 0)
        ops.perturb_apply_s2d(a, self.dtype, self._xs2d)
        self.net.forward(self._xs2d, self._logits)
        gbatch = self.B * self.world
        sm, dl, pc = ops.softmax_adv_loss(self._logits, labels, dialect="tf", improve_loss=improve_loss, use_logits=use_logits,
                                          targeted=targeted, margin=margin, mean_scale=1.0 / gbatch)
        self.net.backward(dl, self._gx)
        if not hasattr(self, "_gdense"):
            self._gdense = torch.empty_like(self.eps_rgb)
        ops.perturb_grad_reduce(a, self._gx, self._gdense)
        tail = pc[:, :3].sum(0)
        parallel.allreduce_sum_(self._gdense, self.pg)            # RCCL over xGMI: the dense gradient (38.5 MB at 64 x 224 x 224)
        parallel.allreduce_sum_(tail, self.pg)
        res = StepResult(adv_loss=tail[0].clone(), softmax=sm, label_prob=pc[:, 1], argmax=pc[:, 3].to(torch.int64),
                         prob_to_min=(tail[2] if targeted else tail[1]) / gbatch, prob_to_max=(tail[1] if targeted else tail[2]) / gbatch)
        res["is_adversarial"] = (res["argmax"] == labels).all() if targeted else (res["argmax"] != labels).all()
        if update:
            self.adam_t += 1
            sc = ops.perturb_dense_l12_adam(self._gdense, self.eps_rgb, self.adam_m, self.adam_v, self.adam_t, dialect="tf", beta=beta0 * beta1,
                                            lr=lr).clone()
            res.update(reg_loss=beta1 * sc[0], L12=sc[0], thickness=sc[1], roughness=sc[2], pert_max=sc[3],
                       total_loss=res["adv_loss"] + beta0 * beta1 * sc[0], thickness_relative=sc[1] / 2 * 100, roughness_relative=sc[2] / 2 * 100)
        return res

    # ---- the attribute surface of the reference object (kinetics_i3d_utils.py:100-200, read by the scripts through sess.run:
    # i3d_adversarial_main_single_video_npy.py:61-77) -- values of the LAST step() / logits() call, as device tensors ----------------
    def _last(self, key):
        if getattr(self, "last_result", None) is None or key not in self.last_result:
            raise AttributeError(f"{key}: no attack iteration has run yet (call step() first)")
        return self.last_result[key]

    softmax = property(lambda self: self._last("softmax"))
    model_logits = property(lambda self: self._logits)
    labels = property(lambda self: self._last("_labels"))
    norm_reg = property(lambda self: self._last("norm_reg"))
    diff_norm_reg = property(lambda self: self._last("diff_norm_reg"))
    laplacian_norm_reg = property(lambda self: self._last("laplacian_norm_reg"))
    thickness = property(lambda self: self._last("thickness"))
    roughness = property(lambda self: self._last("roughness"))
    to_min_prob = property(lambda self: self._last("prob_to_min"))
    to_max_prob = property(lambda self: self._last("prob_to_max"))
    rgb_input = property(lambda self: self._last_x)

    @property
    def adversarial_inputs_rgb(self):
        """the perturbed clip ``clip(x + a * clip(eps_rgb, +-0.4), -1, 1)`` [B,T,224,224,3] fp32 (kinetics_i3d_utils.py:104-142) of the
        last clip seen, under the CURRENT perturbation and the last call's rolls: the apply kernel's fp32 output, unfolded from its
        space-to-depth layout (data movement only)"""
        if getattr(self, "_last_x", None) is None:
            raise AttributeError("adversarial_inputs_rgb: no clip has been seen yet")
        sx, sp = self._last_shifts
        a = ops.make_apply_args(self._last_x, self.eps_rgb, dialect="tf", dclip=0.0 if self.dense else 0.4, adv_flag=1.0, shift_x=sx, shift_p=sp, fold_t=2)
        f = ops.perturb_apply_s2d(a, "f32")                                     # [B,T/2,H/2,W/2,32], channel (qt*4+qh*2+qw)*3+c
        B, T2, H2, W2 = f.shape[:4]
        return f[..., :24].reshape(B, T2, H2, W2, 2, 2, 2, 3).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, 2 * T2, 2 * H2, 2 * W2, 3).contiguous()

    def delta_gradient(self):
        """last all-reduced adversarial gradient d(adv)/d(delta), [T,3] ([B,T,3] in per-clip mode)"""
        if self.per_clip:
            return self._gclip
        return self._red[:self.T * 3].view(self.T, 3)

    # ---- fooling rate --------------------------------------------------------------------------------
    def evaluate(self, batches, targeted_attack=False, target_class_id=None, cyclic=0, exclude_misclassify=True):
        """kinetics_i3d.evaluate (kinetics_i3d_utils.py:217-250) over an iterable of (clip, labels) batches.
        Returns (miss_rate, total_valid) aggregated over all ranks."""
        cnt = parallel.FoolingCounter(self.eps_rgb.device)
        for x, y in batches:
            adv = self.logits(x, 1.0, cyclic).argmax(-1)
            clean = self.logits(x, 0.0, 0).argmax(-1) if exclude_misclassify else None
            cnt.update(adv, clean, y, targeted_attack, target_class_id, exclude_misclassify)
        return cnt.result(self.pg)


class FlickerI3DInference(FlickerI3D):
    """``kinetics_i3d_inference`` (kinetics_i3d_utils.py:574-647): softmax of a clip under a FIXED perturbation with
    independent random rolls of the clip (``cyclic_input_flag``) and of the perturbation (``cyclic_eps_flag``).  Unlike the
    training graph the perturbation is not clipped to +-0.4 here (kinetics_i3d_utils.py:619-624); only the perturbed clip
    is clipped to [-1, 1]."""

    def set_perturbation(self, delta):
        """load eps_rgb ([T,1,1,3] or [T,3]), e.g. ``tf_checkpoint.read_bundle(ckpt)['RGB/eps']``"""
        self.reset_perturbation(delta)

    def __call__(self, inputs, adv_flag=0, cyclic_input_flag=0, cyclic_eps_flag=0):
        x = self._check_x(inputs)
        self.last_shift_x = int(self._rng.integers(0, self.T)) if cyclic_input_flag else 0
        self.last_shift_p = int(self._rng.integers(0, self.T)) if cyclic_eps_flag else 0
        a = ops.make_apply_args(x, self.eps_rgb, dialect="tf", dclip=0.0, adv_flag=float(adv_flag), shift_x=self.last_shift_x,
                                shift_p=self.last_shift_p, fold_t=ops.I3D_FOLD, center=self.exact_delta_forward)
        return torch.softmax(self._forward(a), -1)

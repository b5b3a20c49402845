"""Host-side mirror of the reference's torch attack surface (utils_cv/action_recognition/model.py:58-330) on top of
libflicker_hip.so: ``Perturbation``, ``Losses``, ``Adversarial_metrics`` and a VideoResNet attack engine
(``FlickerVideoResNet``, the hot loop of ``VideoLearnerAdversarial.fit_single_video_attack``, model.py:984-1205).

Same names, argument meaning and error behaviour as the reference classes; the arithmetic runs in the HIP kernels
(torch dialect flags), pinned by the reference's own golden vectors (tests/golden/torch_attack_golden.npz).
Clips are channels-last ``[B,T,H,W,3]`` on the device (the reference's NCDHW ``[B,3,T,H,W]`` permuted once at load)."""
import numpy as np
import torch

from . import ops, parallel
from ._lib import FLK_NET_MC3_18, FLK_NET_R2PLUS1D_18, FLK_NET_R3D_18
from .videoresnet_spec import DEFAULT_MEAN, DEFAULT_STD

ARCH_CODES = {"r2plus1d_18": FLK_NET_R2PLUS1D_18, "r3d_18": FLK_NET_R3D_18, "mc3_18": FLK_NET_MC3_18}


class Perturbation:
    """model.py:58-129.  size = [3,T,1,1] (flickering) or [3,T,H,W] (the dense "L12" attack, model.py:380-384); the parameter
    is stored time-major / channels-last on the device: [T,3] or [T,H,W,3]."""

    def __init__(self, size, requires_grad=True, device="cuda", max_value=None, min_value=None, max_norm=1.0, cyclic_pert=False, batch=None):
        if len(size) != 4 or size[0] != 3:
            raise ValueError(f"perturbation size must be [3,T,1,1] or [3,T,H,W], got {tuple(size)}")
        self.size, self.device, self.requires_grad = tuple(size), device, requires_grad
        self.T = size[1]
        self.dense = not (size[2] == 1 and size[3] == 1)
        # batch = B (flickering only): B INDEPENDENT perturbations [B,T,3], one per clip of the batch, each with its own clamp bound
        # (``dyn_max_norm_dev`` [B]) -- single-video attacks advancing together (fit_many_videos, model.py:791-982)
        self.batch = batch
        if batch is not None and (self.dense or cyclic_pert):
            raise ValueError("batch: flickering perturbations without cyclic roll only")
        # model.py:72-75: attributes are only set when the argument is None -- reproduced
        if max_value is None:
            self.max_value = float(np.min((1 - np.array(DEFAULT_MEAN)) / DEFAULT_STD))
        if min_value is None:
            self.min_value = float(np.max((0.0 - np.array(DEFAULT_MEAN)) / DEFAULT_STD))
        self.max_norm = self.dynamic_max_norm = max_norm
        self.cyclic_pert = cyclic_pert
        self._rng = np.random.default_rng(0)
        self.perturbation = None
        if batch is not None:
            self.perturbation = torch.zeros((batch, self.T, 3), dtype=torch.float32, device="cuda")
            self.dyn_max_norm_dev = torch.full((batch,), float(max_norm), dtype=torch.float32, device="cuda")
            for b in range(batch):
                self.init_clip(b)
        else:
            self.init_perturbation()

    @property
    def _dev_shape(self):
        if self.batch is not None:
            return (self.batch, self.T, 3)
        return (self.T, self.size[2], self.size[3], 3) if self.dense else (self.T, 3)

    def init_clip(self, b, perturbation=(), max_norm=None):
        """batch mode: (re)start slot b -- ``init_perturbation`` for that clip alone, clamp bound back to ``max_norm`` (model.py:938-947)"""
        p = (self._rng.random(self.size, dtype=np.float32) * 2 - 1) * 1e-6 if len(perturbation) == 0 else perturbation
        p = np.asarray(p, dtype=np.float32).reshape(self.size)
        self.perturbation[b].copy_(torch.from_numpy(np.ascontiguousarray(np.transpose(p, (1, 2, 3, 0)).reshape(self.T, 3))))
        self.dyn_max_norm_dev[b] = float(self.max_norm if max_norm is None else max_norm)

    def _to_dev(self, p_cthw):
        """[3,T,H,W] (reference layout) -> device layout"""
        p = np.asarray(p_cthw, dtype=np.float32).reshape(self.size)
        return np.ascontiguousarray(np.transpose(p, (1, 2, 3, 0)).reshape(self._dev_shape))

    def _to_ref(self, t):
        """device layout -> [3,T,H,W] like the reference tensors"""
        return t.reshape(self.T, self.size[2], self.size[3], 3).permute(3, 0, 1, 2)

    def init_perturbation(self, perturbation=(), requires_grad=True, device="cuda"):
        """model.py:121-126: U(-1,1)*1e-6 or the given numpy array [3,T,1,1] / [3,T,H,W] (batch mode: every slot gets it)"""
        if self.batch is not None:
            for b in range(self.batch):
                self.init_clip(b, perturbation, max_norm=float(self.dyn_max_norm_dev[b]))
            return
        if len(perturbation) == 0:
            p = (self._rng.random(self.size, dtype=np.float32) * 2 - 1) * 1e-6
        else:
            p = perturbation
        self.perturbation = torch.from_numpy(self._to_dev(p)).cuda()

    def apply_args(self, x, adversarial=True, fold_t=1):
        """fold_t: 1 = the (h,w)-folded 16-channel tensor; 4 = the same as two bf16 numbers per value (the input of bf16 plans)"""
        shift = int(self._rng.integers(0, self.T)) if (self.cyclic_pert and adversarial) else 0   # model.py:91-92
        inf = float("inf")
        return ops.make_apply_args(x, self.perturbation, dialect="torch", dclip=self.dynamic_max_norm,
                                   adv_flag=1.0 if adversarial else 0.0, shift_p=shift,
                                   inv_std=tuple(1.0 / s for s in DEFAULT_STD),
                                   lo=self.min_value if adversarial else -inf, hi=self.max_value if adversarial else inf, fold_t=fold_t,
                                   dclip_dev=self.dyn_max_norm_dev if self.batch is not None else None)

    def forward(self, input):
        """model.py:80-101: ``input = [x, adversarial]`` -> the perturbed (or, with adversarial False, the untouched) clip.  ``x`` is the
        reference's NCDHW tensor [B,3,T,H,W] or the channels-last [B,T,H,W,3] the engine works on; the result has x's layout.  The
        kernel writes the (h,w)-folded tensor the network consumes (flk_perturb_apply_s2d); it is unfolded here."""
        x, adversarial = input
        ncdhw = x.dim() == 5 and x.shape[1] == 3 and x.shape[-1] != 3
        xcl = (x.permute(0, 2, 3, 4, 1) if ncdhw else x).contiguous().float().cuda()
        B, T, H, W, _ = xcl.shape
        if T != self.T:
            raise ValueError(f"clip has {T} frames, the perturbation {self.T}")
        folded = ops.perturb_apply_s2d(self.apply_args(xcl, bool(adversarial)), torch.float32)      # [B,T,H/2,W/2,16]
        out = folded[..., :12].reshape(B, T, H // 2, W // 2, 2, 2, 3).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, T, H, W, 3)
        return out.permute(0, 4, 1, 2, 3).contiguous() if ncdhw else out.contiguous()

    __call__ = forward

    @staticmethod
    def convert_adversarial_video_zero_one(adv_vid):
        """model.py:107-112: the normalised clip back to pixel range, as a NUMPY array ``[B,T,H,W,3]`` in [0,1]:
        ``(x^T + mean/std) * std`` (float64, like the reference's numpy arithmetic).  ``adv_vid`` is the reference's NCDHW
        tensor [B,3,T,H,W] or the engine's channels-last [B,T,H,W,3]."""
        x = adv_vid.detach().cpu().numpy()
        if x.ndim == 5 and x.shape[1] == 3 and x.shape[-1] != 3:
            x = x.transpose([0, 2, 3, 4, 1])
        return (x + np.array(DEFAULT_MEAN) / np.array(DEFAULT_STD)) * np.array(DEFAULT_STD)

    def apply_perturbation(self, x):
        """model.py:103-105: the perturbed clip de-normalised to [0,1], numpy [B,T,H,W,3] (what the scripts save / plot)"""
        return self.convert_adversarial_video_zero_one(self.forward([x, True]))

    def clamp_perturbation(self):
        if self.batch is not None:
            bound = self.dyn_max_norm_dev.view(-1, 1, 1)
            return torch.minimum(torch.maximum(self.perturbation, -bound), bound)
        return self.perturbation.clamp(-self.dynamic_max_norm, self.dynamic_max_norm)

    def clip_perturbation_ref(self, b):
        """batch mode: (clamped) perturbation of slot b in the reference layout [3,T,1,1]"""
        return self.clamp_perturbation()[b].t().reshape(3, self.T, 1, 1)

    def get_perturbation(self):
        """(clamped, raw) as [3,T,1,1] / [3,T,H,W] like the reference (model.py:128-129)"""
        return self._to_ref(self.clamp_perturbation()), self._to_ref(self.perturbation)

    def metric_calc(self):
        """model.py:113-118 on the raw parameter: roll over the time axis"""
        p = self.perturbation
        return p.abs().mean() * 100.0, (torch.roll(p, 1, 0) - p).abs().mean() * 100.0


class Losses:
    """model.py:131-250: __call__(labels, logits, prob, perturbation) -> [loss, adv, reg]; here the adversarial part and
    d(adv)/d(logits) come from flk_softmax_adv_loss (torch dialect), the regulariser from flk_perturb_reg_adam."""

    def __init__(self, beta_1=0.5, lambda_=1.0, targeted=False, target_class=None, margin=0.05, improve_loss=False, logits=False,
                 attack_type="flickering"):
        if attack_type not in ("flickering", "L12"):
            raise ValueError(f"attack_type must be 'flickering' or 'L12' (model.py:165-168), got {attack_type!r}")
        if targeted and improve_loss:
            # model.py:223-225 references undefined names: the reference crashes here; refuse instead of guessing
            raise NotImplementedError("the reference's targeted improve-loss is non-functional (model.py:223-225)")
        self.beta_1, self.lambda_, self.targeted, self.target_class = beta_1, lambda_, targeted, target_class
        self.margin, self.improve_loss, self.logits, self.attack_type = margin, improve_loss, logits, attack_type
        self.label_prob = None

    def __call__(self, labels, model_logits, prob, perturbation):
        """model.py:169-175: ``[loss, adv_loss, reg_loss]`` (values; the engine's step takes the gradients from the kernels).
        ``prob`` is accepted for signature parity -- the kernel recomputes the softmax; ``perturbation`` is the clamped delta in the
        reference layout [3,T,1,1] / [3,T,H,W] (model.py:1078).  Sets ``label_prob`` like the reference (:232)."""
        _, _, pc = self.adv(labels, model_logits.contiguous().float(), model_logits.shape[0])
        adv_loss = pc[:, 0].sum()
        reg_loss = self.regularization_loss(perturbation)
        return [adv_loss + self.lambda_ * reg_loss, adv_loss, reg_loss]

    def adv(self, labels, model_logits, global_batch, out=None):
        lab = labels if not self.targeted else torch.full_like(labels, self.target_class)
        sm, dl, pc = ops.softmax_adv_loss(model_logits, lab, dialect="torch", improve_loss=self.improve_loss, use_logits=self.logits,
                                          targeted=self.targeted, margin=self.margin, mean_scale=1.0 / global_batch, out=out)
        self.label_prob = pc[:, 1]
        return sm, dl, pc

    def flickering_regularization_loss(self, perturbation):
        """model.py:198-209 on the CLAMPED perturbation [3,T,1,1] (model.py:1078): value only -- the gradient and the same
        value during an update come from flk_perturb_reg_adam.  Used for evaluation passes (no Adam step)."""
        p = perturbation
        right, left = torch.roll(p, 1, dims=1), torch.roll(p, -1, dims=1)
        norm_reg = torch.mean(p ** 2) + 1e-12
        diff_norm_reg = torch.mean((p - right) ** 2) + 1e-12
        laplacian_norm_reg = torch.mean((-2 * p + right + left) ** 2) + 1e-12
        return self.beta_1 * norm_reg + (1 - self.beta_1) * (diff_norm_reg + laplacian_norm_reg)

    def L12_regularization_loss(self, perturbation):
        """model.py:211-214 on the clamped perturbation [3,T,H,W]: value only (update passes take it from flk_perturb_dense_l12_adam)"""
        return torch.sum(torch.sqrt(torch.mean(perturbation ** 2, [0, 2, 3]))) + 1e-12

    def regularization_loss(self, perturbation):
        return self.L12_regularization_loss(perturbation) if self.attack_type == "L12" else self.flickering_regularization_loss(perturbation)


class Adversarial_metrics:
    """model.py:253-330"""

    def __init__(self, targeted=False, target_class=None):
        self.targeted, self.target_class = targeted, target_class

    def accuracy(self, output, ground_truth, topk=(1,), clean_pred=None):
        """model.py:262-291: fooling percentages as a list of 1-element tensors.  Targeted: ONE entry, 100 * #(top-maxk predictions
        equal to the target class) / batch.  Untargeted: per k, 100 * (1 - #(adv top-k hit AND clean top-k hit, at the SAME rank
        position) / #(clean top-k hits)) -- the rank-wise product is the reference's (model.py:287), reproduced as is."""
        with torch.no_grad():
            res = []
            batch_size = ground_truth.size(0)
            maxk = max(topk)
            pred = output.topk(maxk, 1, True, True)[1].t()
            if self.targeted:
                correct = pred.eq(self.target_class).reshape(-1).float().sum(0, keepdim=True)
                res.append(correct[0] * (100.0 / batch_size))
                return res
            correct = pred.eq(ground_truth.view(1, -1).expand_as(pred))
            pred_no_adv = clean_pred.topk(maxk, 1, True, True)[1].t()
            correct_no_adv = pred_no_adv.eq(ground_truth.view(1, -1).expand_as(pred_no_adv))
            for k in topk:
                correct_k = (correct[:k] * correct_no_adv[:k]).reshape(-1).float().sum(0, keepdim=True)
                res.append((1 - correct_k * (1.0 / correct_no_adv[:k].reshape(-1).float().sum())) * 100.0)
            return res

    def accuracy_for_eval(self, output, ground_truth, topk=(1,), clean_pred=None):
        adv, clean = output.argmax(1), clean_pred.argmax(1)
        correct_clean = clean == ground_truth
        if self.targeted:
            return (adv == self.target_class).float().sum() * (100.0 / ground_truth.numel())
        return ((adv != ground_truth) & correct_clean).float().sum(), correct_clean.float().sum()

    def adversarial_metric(self, perturbation):
        return perturbation.abs().mean() * 100.0, (torch.roll(perturbation, 1, dims=1) - perturbation).abs().mean() * 100.0


class FlickerVideoResNet:
    """Attack engine for torchvision-0.5.0 r2plus1d_18 / r3d_18 / mc3_18 (model.py:337-399,984-1205)."""

    def __init__(self, base_model, weights, batch_size=1, sample_length=16, image_size=112, dtype="bf16", device=0, l_inf_pert_norm=0.2,
                 cyclic_pert=False, num_classes=400, process_group=None, attack_type="flickering", per_clip=False):
        if base_model not in ARCH_CODES:
            raise ValueError(f"base_model must be one of {sorted(ARCH_CODES)} (model.py:47-56), got {base_model!r}")
        if not torch.cuda.is_available():
            raise RuntimeError("FlickerVideoResNet needs an MI355X (HIP) device; there is no CPU fallback")
        torch.cuda.set_device(device)
        self.model_name, self.B, self.T, self.H, self.W, self.dtype = base_model, batch_size, sample_length, image_size, image_size, dtype
        self.pg, self.world = process_group, parallel.world_size(process_group)
        self.net = ops.Net(ARCH_CODES[base_model], dtype, self.B, self.T, self.H, self.W, weights, device)
        if attack_type not in ("flickering", "L12"):
            raise ValueError(f"attack_type must be 'flickering' or 'L12', got {attack_type!r}")
        self.attack_type = attack_type
        # model.py:380-384: [3,T,1,1] for the flickering attack, a dense [3,T,H,W] perturbation otherwise
        # per_clip: batch_size INDEPENDENT single-video attacks in one batch (fit_many_videos(batch=...)): a perturbation, Adam state,
        # step counter, clamp bound and "still attacking" flag per clip; replicas only (no collective)
        self.per_clip = bool(per_clip)
        if self.per_clip and (attack_type != "flickering" or cyclic_pert or self.world > 1):
            raise ValueError("per_clip: flickering attack, no cyclic roll, one rank")
        self.pert_model = Perturbation((3, self.T, 1, 1) if attack_type == "flickering" else (3, self.T, self.H, self.W),
                                       max_norm=l_inf_pert_norm, cyclic_pert=cyclic_pert, batch=self.B if self.per_clip else None)
        dev = torch.device("cuda", device)
        tdt = torch.bfloat16 if dtype in ("bf16", torch.bfloat16) else torch.float32
        # bf16 plans take the clip as TWO bf16 numbers per value (32 channels, fold_t = 4: the stem sees x + delta/std to ~16 bits -- one
        # bf16 per value swallowed |delta| = 1e-4 outright; the reference STARTS at U(+-1e-6), model.py:71); gradients: 16 channels
        self._xs = torch.empty((self.B, self.T, self.H // 2, self.W // 2, self.net.input_channels), dtype=tdt, device=dev)
        self._gx = torch.empty((self.B, self.T, self.H // 2, self.W // 2, 16), dtype=tdt, device=dev)
        self._logits = torch.empty((self.B, num_classes), dtype=torch.float32, device=dev)
        self._red = torch.zeros(parallel.payload_size(self.T), dtype=torch.float32, device=dev)
        self._scratch = torch.empty(max(1, ops.load().flk_perturb_grad_scratch_bytes(self.B, self.T, self.H, self.W) // 4), dtype=torch.float32, device=dev)
        self._scalars = torch.empty(8, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.pert_model._dev_shape, device=dev)
        self.adam_v = torch.zeros(self.pert_model._dev_shape, device=dev)
        self.adam_t = 0      # the reference keeps ONE Adam instance across videos: not reset by init_perturbation
        if self.per_clip:
            self.adam_steps = torch.zeros(self.B, dtype=torch.int32, device=dev)
            self.active = torch.ones(self.B, dtype=torch.int32, device=dev)

    def _forward(self, x, adversarial):
        """Perturbation -> network: the apply arguments of this call (the backward pass masks with the same) ; logits in self._logits"""
        a = self.pert_model.apply_args(self._check_x(x), adversarial, fold_t=self.net.input_fold)
        self.net.forward_apply(a, self._xs, self._logits)           # the plan applies the perturbation in front of its stem
        return a

    def _check_x(self, x):
        if tuple(x.shape) != (self.B, self.T, self.H, self.W, 3) or x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError(f"clip must be a CUDA float32 channels-last tensor {(self.B, self.T, self.H, self.W, 3)}, got {tuple(x.shape)} {x.dtype}")
        return x.contiguous()

    def logits(self, x, adversarial=False):
        """model([x, adversarial]) (model.py:1028,1073)"""
        self._forward(x, adversarial)
        return self._logits

    def step(self, x, labels, criterion, lr=1e-3, update=True):
        """one iteration of fit_single_video_attack (model.py:1073-1101): forward, Losses, backward, torch-Adam step.
        The kernels write into one of ``RESULT_SLOTS`` result slots (valid for the next RESULT_SLOTS - 1 iterations);
        ``loss`` / ``argmax`` are derived on first access (i3d_engine.StepResult)."""
        from .i3d_engine import RESULT_SLOTS, StepResult
        if criterion.attack_type != self.attack_type:
            raise ValueError(f"criterion.attack_type {criterion.attack_type!r} != engine attack_type {self.attack_type!r}")
        if self.attack_type == "L12":
            return self._step_dense(x, labels, criterion, lr, update)
        if self.per_clip:
            return self._step_per_clip(x, labels, criterion, lr, update)
        if not hasattr(self, "_slots"):
            dev = self._logits.device
            self._slots = [dict(payload=torch.zeros(parallel.payload_size(self.T), dtype=torch.float32, device=dev),
                                sm=torch.empty_like(self._logits), pc=torch.empty((self.B, 4), dtype=torch.float32, device=dev),
                                scalars=torch.zeros(8, dtype=torch.float32, device=dev)) for _ in range(RESULT_SLOTS)]
            self._dl = torch.empty_like(self._logits)
            self._it = 0
        slot = self._slots[self._it % RESULT_SLOTS]
        self._it += 1
        red, sm, pc = slot["payload"], slot["sm"], slot["pc"]
        self._red = red
        a = self._forward(x, True)
        gbatch = self.B * self.world
        criterion.adv(labels, self._logits, gbatch, out=(sm, self._dl, pc))
        self.net.backward(self._dl, self._gx)
        n = 3 * self.T
        ops.perturb_grad_reduce(a, self._gx, red[:n].view(self.T, 3), self._scratch)
        ops.pack_batch_sums(pc, 1.0 / gbatch, red[n:])
        parallel.allreduce_sum_(red, self.pg)
        res = StepResult(adv_loss=red[n], softmax=sm, label_prob=pc[:, 1], _argmax_f=pc[:, 3], _labels=labels, _targeted=bool(criterion.targeted))
        if update:
            self.adam_t += 1
            b1 = criterion.beta_1
            sc = slot["scalars"]
            ops.perturb_reg_adam(red[:n], self.pert_model.perturbation, self.adam_m, self.adam_v, self.adam_t, dialect="torch",
                                 beta0=criterion.lambda_, beta1=b1, beta2=1 - b1, beta3=1 - b1,
                                 dyn_max_norm=self.pert_model.dynamic_max_norm, lr=lr, scalars=sc)
            res.update(reg_loss=sc[0], _reg_weight=criterion.lambda_, _thickness=sc[4], _roughness=sc[5])
        else:
            res.update(reg_loss=criterion.regularization_loss(self.pert_model.get_perturbation()[0]), _reg_weight=criterion.lambda_)
        return res

    def _step_per_clip(self, x, labels, criterion, lr, update):
        """one iteration of B independent single-video attacks (torch dialect): per-clip loss / gradient / Adam, everything [B]-shaped.
        Per clip the arithmetic is that of ``step`` on a batch of one (bitwise in fp32)."""
        from .i3d_engine import RESULT_SLOTS, StepResult
        if not hasattr(self, "_slots"):
            dev = self._logits.device
            self._slots = [dict(sm=torch.empty_like(self._logits), pc=torch.empty((self.B, 4), dtype=torch.float32, device=dev),
                                scalars=torch.zeros((self.B, 8), dtype=torch.float32, device=dev),
                                g=torch.zeros((self.B, self.T, 3), dtype=torch.float32, device=dev)) for _ in range(RESULT_SLOTS)]
            self._dl = torch.empty_like(self._logits)
            self._it = 0
        slot = self._slots[self._it % RESULT_SLOTS]
        self._it += 1
        sm, pc, g = slot["sm"], slot["pc"], slot["g"]
        a = self._forward(x, True)
        criterion.adv(labels, self._logits, 1, out=(sm, self._dl, pc))            # every clip is its own batch of one
        self.net.backward(self._dl, self._gx)
        ops.perturb_grad_reduce(a, self._gx, g, self._scratch)
        self._gclip = g
        am = pc[:, 3].to(torch.int64)
        res = StepResult(adv_loss=pc[:, 0], softmax=sm, label_prob=pc[:, 1], argmax=am, _labels=labels, _targeted=bool(criterion.targeted),
                         _reg_weight=criterion.lambda_)
        if update:
            b1 = criterion.beta_1
            sc = slot["scalars"]
            ops.perturb_reg_adam_batched(g, self.pert_model.perturbation, self.adam_m, self.adam_v, self.adam_steps, self.active, dialect="torch",
                                         beta0=criterion.lambda_, beta1=b1, beta2=1 - b1, beta3=1 - b1, lr=lr, scalars=sc,
                                         dyn_max_norm_dev=self.pert_model.dyn_max_norm_dev)
            res.update(reg_loss=sc[:, 0], _thickness=sc[:, 4], _roughness=sc[:, 5])
        else:
            pc_ = self.pert_model.clamp_perturbation()
            res.update(reg_loss=torch.stack([criterion.regularization_loss(pc_[b].t().reshape(3, self.T, 1, 1)) for b in range(self.B)]))
        return res

    def _step_dense(self, x, labels, criterion, lr, update):
        """the dense "L12" attack (model.py:211-214,380-384): loss = adv + lambda * L12(clamped delta); the data-parallel payload
        is the dense gradient [T,H,W,3] (2.4 MB at 16 x 112 x 112)"""
        from .i3d_engine import StepResult
        a = self._forward(x, True)
        gbatch = self.B * self.world
        sm, dl, pc = criterion.adv(labels, self._logits, gbatch)
        self._dl = dl
        self.net.backward(dl, self._gx)
        if not hasattr(self, "_gdense"):
            self._gdense = torch.empty_like(self.pert_model.perturbation)
        ops.perturb_grad_reduce(a, self._gx, self._gdense)
        tail = pc[:, :3].sum(0)
        parallel.allreduce_sum_(self._gdense, self.pg)            # RCCL over xGMI: the dense gradient (2.4 MB at 16 x 112 x 112)
        parallel.allreduce_sum_(tail, self.pg)
        res = StepResult(adv_loss=tail[0].clone(), softmax=sm, label_prob=pc[:, 1], _argmax_f=pc[:, 3], _labels=labels,
                         _targeted=bool(criterion.targeted), _reg_weight=criterion.lambda_)
        if update:
            self.adam_t += 1
            sc = ops.perturb_dense_l12_adam(self._gdense, self.pert_model.perturbation, self.adam_m, self.adam_v, self.adam_t, dialect="torch",
                                            beta=criterion.lambda_, lr=lr, dyn_max_norm=self.pert_model.dynamic_max_norm).clone()
            res.update(reg_loss=sc[0], _thickness=sc[1], _roughness=sc[2])
        else:
            res.update(reg_loss=criterion.L12_regularization_loss(self.pert_model.get_perturbation()[0]))
        return res

    # ---- drivers around step(): VideoLearnerAdversarial's loops without the plotting ------------------------------------
    def fit_single_video_attack(self, inputs, target, criterion, lr=1e-3, n_iter=3000, targeted_attack=False, target_class_id=None,
                                restart_after=3000, norm_growth=1.3, max_restarts=4, log_every=0):
        """``VideoLearnerAdversarial.fit_single_video_attack`` (model.py:984-1205).

        Returns None when the clean clip is misclassified (model.py:1030-1032).  Otherwise iterates
        ``while step < n_iter or not is_adversarial`` (model.py:1056); whenever ``step > restart_after`` the clamp norm
        grows by ``norm_growth`` and the step counter restarts, giving up after ``max_restarts`` (model.py:1061-1066:
        3000 / 1.3 / 4).  The result dict has the reference's keys (model.py:1193-1203); per-iteration values are host
        floats (the reference syncs every iteration as well: ``loss.item()``)."""
        outputs_no_adv = self.logits(inputs, False).clone()
        if not bool((outputs_no_adv.argmax(1) == target).all()):
            return None
        tot, adv_l, reg_l, thick_l, rough_l, maxp_l, corr_l, isadv_l, pert_l = [], [], [], [], [], [], [], [], []
        step, new_chance, is_adversarial = 0, 0, False
        while step < n_iter or not is_adversarial:
            if step > restart_after:
                new_chance += 1
                self.pert_model.dynamic_max_norm *= norm_growth
                step = 0
            if new_chance == max_restarts:
                break
            r = self.step(inputs, target, criterion, lr=lr)
            adv_class = r["argmax"]
            is_adversarial = bool((adv_class == target_class_id).all()) if targeted_attack else not bool(adv_class.equal(target))
            isadv_l.append(is_adversarial)
            tot.append(float(r["loss"])); adv_l.append(float(r["adv_loss"])); reg_l.append(float(r["reg_loss"]))
            p = self.pert_model.get_perturbation()[0].cpu().numpy()          # after the update, like model.py:1110-1112
            pert_l.append(p)
            thick_l.append(float(np.abs(p).mean())); rough_l.append(float(np.abs(np.roll(p, 1, 1) - p).mean()))
            maxp_l.append(float(r["softmax"].max())); corr_l.append(float(r["label_prob"][0]))
            if log_every and step % log_every == 0:
                print(f"batch {step} of {n_iter} | loss = {tot[-1]:.4f} | adv loss = {adv_l[-1]:.4f} | reg loss = {reg_l[-1]:.4f} | "
                      f"pert_thickness = {thick_l[-1]:.4f} | pert_roughness = {rough_l[-1]:.4f}", flush=True)
            step += 1
        p = self.pert_model.get_perturbation()[0].cpu().numpy()
        return {"loss/total": tot, "loss/adv_loss": adv_l, "loss/reg_loss": reg_l, "perturbation/thickness": thick_l,
                "perturbation/roughness": rough_l, "perturbation/inf_norm": float(np.abs(p).max()), "perturbation": pert_l,
                "prob_clean_input": outputs_no_adv, "label": target.cpu().numpy(), "is_adversarial": isadv_l,
                "max_prob": maxp_l, "correct_cls_prob": corr_l, "restarts": new_chance}

    def train_an_epoch(self, data_loaders, criterion, metric, lr):
        """``train_an_epoch`` (model.py:627-789): 'train' then 'valid' over iterables of (inputs, target, _); the valid phase
        evaluates the same loss without an update.  Result keys as model.py:780-786."""
        import time
        result = {}
        for phase in ("train", "valid"):
            t0 = time.time()
            n, loss_sum, miss, valid = 0, 0.0, 0.0, 0.0
            for inputs, target, *_ in data_loaders[phase]:
                clean = self.logits(inputs, False).clone()
                r = self.step(inputs, target, criterion, lr=lr, update=(phase == "train"))
                adv_logits = self._logits
                m = metric.accuracy_for_eval(adv_logits, target, topk=(1,), clean_pred=clean)
                if isinstance(m, tuple):
                    miss += float(m[0]); valid += float(m[1])
                else:                                           # targeted: a percentage (model.py:300-302)
                    miss += float(m) / 100.0 * target.numel(); valid += target.numel()
                bs = inputs.shape[0]
                loss_sum += float(r["loss"]) * bs
                n += bs
            p = self.pert_model.get_perturbation()[0].cpu().numpy()
            result[f"{phase}/time"] = time.time() - t0
            result[f"{phase}/loss"] = loss_sum / max(n, 1)
            result[f"{phase}/fooling_ratio"] = miss / valid if valid else float("nan")
            result[f"{phase}/pert_thickness"] = float(np.abs(p).mean())
            result[f"{phase}/pert_roughness"] = float(np.abs(np.roll(p, 1, 1) - p).mean())
            result[f"{phase}/inf_norm"] = float(np.abs(p).max())
            result[f"{phase}/perturbation"] = p
        return result

    def fit(self, data_loaders, criterion, metric, lr=1e-3, epochs=1, lr_gamma=0.1, lr_step_size=None, model_dir=None,
            model_name=None, save_model=False, start_epoch=1):
        """``VideoLearnerAdversarial.fit`` (model.py:460-625) with the step-decay schedule (StepLR, model.py:571-573:
        lr_e = lr * gamma ** floor((e - start_epoch) / lr_step_size), default step = ceil(2/3 * epochs), model.py:496-497).
        Returns the list of per-epoch result dicts; ``save_model`` writes ``{model_name}_{epoch:03d}.npy`` (model.py:613-619)."""
        import os
        if lr_step_size is None:
            lr_step_size = int(np.ceil(2 / 3 * epochs))
        results = []
        for e in range(start_epoch, epochs + 1):
            lr_e = lr * lr_gamma ** ((e - start_epoch) // max(int(lr_step_size), 1))   # a fresh StepLR on every (re)start, like the reference
            res = self.train_an_epoch(data_loaders, criterion, metric, lr_e)
            res["lr"] = lr_e
            results.append(res)
            if save_model and model_dir:
                os.makedirs(model_dir, exist_ok=True)
                np.save(os.path.join(model_dir, f"{model_name or self.model_name}_{str(e).zfill(3)}.npy"), np.array(results, dtype=object),
                        allow_pickle=True)
        return results

    def _fit_many_videos_batched(self, videos, criterion, lr, model_dir, label_id_to_text, save_model, n_iter, targeted_attack, target_class_id,
                                 restart_after=3000, norm_growth=1.3, max_restarts=4, reset_optimizer_per_video=False, log_every=0):
        """``fit_many_videos`` with B = batch_size videos attacked AT ONCE (engine built with ``per_clip=True``): every slot runs the loop of
        ``fit_single_video_attack`` (model.py:1056-1101: while step < n_iter or not adversarial; restart with 1.3x clamp bound, give up
        after 4) on its own video with its own perturbation / clamp bound / step counters; a finished slot takes the next video.  Per
        video the iterations are those of the one-by-one loop (bitwise in fp32 when the optimiser state is reset per video; the
        reference's single Adam instance carried from video to video, SURVEY D.5, becomes one carried state PER SLOT here)."""
        import os
        assert self.per_clip
        B, T = self.B, self.T
        dev = self._logits.device
        x = torch.zeros((B, T, self.H, self.W, 3), dtype=torch.float32, device=dev)
        labels = torch.zeros(B, dtype=torch.int64, device=dev)
        rng = np.random.default_rng(0)
        it = iter(videos)
        slots, out = [None] * B, {}

        def refill(b):
            while True:
                nxt = next(it, None)
                if nxt is None:
                    slots[b] = None
                    self.active[b] = 0
                    return
                inputs, target, name = nxt
                cls = (label_id_to_text[int(target[0])] if label_id_to_text is not None else str(int(target[0]))).replace(" ", "_")
                dest = os.path.join(model_dir, f"{os.path.basename(str(name))}_@{cls}.npy") if model_dir else None
                if dest and os.path.exists(dest):
                    prev = np.load(dest, allow_pickle=True).tolist()
                    if prev is None or np.array(prev["is_adversarial"]).any():
                        continue
                elif dest and save_model:
                    os.makedirs(model_dir, exist_ok=True)
                    np.save(dest, None)
                self.pert_model.init_clip(b, (rng.random(self.pert_model.size, dtype=np.float32) * 2 - 1) * 0.005)     # model.py:938-947
                if reset_optimizer_per_video:
                    self.adam_m[b].zero_(); self.adam_v[b].zero_(); self.adam_steps[b] = 0
                self.active[b] = 1
                x[b].copy_(inputs[0])
                labels[b] = int(target[0])
                clean = self.logits(x, False)[b:b + 1].clone()
                if int(clean.argmax(1)) != int(target[0]):
                    out[str(name)] = None                                  # model.py:1030-1032
                    continue
                slots[b] = dict(name=str(name), dest=dest, target=target.clone(), clean=clean, step=0, new_chance=0, is_adv=False,
                                tot=[], adv=[], reg=[], thick=[], rough=[], maxp=[], corr=[], isadv=[], pert=[])
                return

        def finish(b):
            st = slots[b]
            p = self.pert_model.clip_perturbation_ref(b).cpu().numpy()
            res = {"loss/total": st["tot"], "loss/adv_loss": st["adv"], "loss/reg_loss": st["reg"], "perturbation/thickness": st["thick"],
                   "perturbation/roughness": st["rough"], "perturbation/inf_norm": float(np.abs(p).max()), "perturbation": st["pert"],
                   "prob_clean_input": st["clean"], "label": st["target"].cpu().numpy(), "is_adversarial": st["isadv"],
                   "max_prob": st["maxp"], "correct_cls_prob": st["corr"], "restarts": st["new_chance"]}
            out[st["name"]] = res
            if st["dest"] and save_model:
                np.save(st["dest"], dict(res, prob_clean_input=res["prob_clean_input"].cpu().numpy()), allow_pickle=True)
            refill(b)

        for b in range(B):
            refill(b)
        while any(st is not None for st in slots):
            # the loop head of fit_single_video_attack, per slot (a refilled slot is checked again: its fresh state passes trivially)
            for b in range(B):
                while slots[b] is not None:
                    st = slots[b]
                    if not (st["step"] < n_iter or not st["is_adv"]):
                        finish(b)
                        continue
                    if st["step"] > restart_after:
                        st["new_chance"] += 1
                        self.pert_model.dyn_max_norm_dev[b] *= norm_growth
                        st["step"] = 0
                    if st["new_chance"] == max_restarts:
                        finish(b)
                        continue
                    break
            if not any(st is not None for st in slots):
                break
            r = self.step(x, labels, criterion, lr=lr)
            h = r.host()
            pcl = self.pert_model.clamp_perturbation().cpu().numpy()               # [B,T,3] after the update (model.py:1110-1112)
            for b, st in enumerate(slots):
                if st is None:
                    continue
                adv_class = int(h["argmax"][b])
                st["is_adv"] = (adv_class == target_class_id) if targeted_attack else (adv_class != int(st["target"][0]))
                st["isadv"].append(st["is_adv"])
                st["tot"].append(float(h["loss"][b])); st["adv"].append(float(h["adv_loss"][b])); st["reg"].append(float(h["reg_loss"][b]))
                p = pcl[b].reshape(T, 1, 1, 3).transpose(3, 0, 1, 2)      # [3,T,1,1], the memory layout get_perturbation() hands the one-by-one loop
                                                                          # (numpy's float32 mean depends on it in the last bit)
                st["pert"].append(p)
                st["thick"].append(float(np.abs(p).mean())); st["rough"].append(float(np.abs(np.roll(p, 1, 1) - p).mean()))
                st["maxp"].append(float(h["softmax"][b].max())); st["corr"].append(float(h["label_prob"][b]))
                if log_every and st["step"] % log_every == 0:
                    print(f"[{st['name']}] batch {st['step']} of {n_iter} | loss = {st['tot'][-1]:.4f} | adv loss = {st['adv'][-1]:.4f} | "
                          f"reg loss = {st['reg'][-1]:.4f}", flush=True)
                st["step"] += 1
        return out

    def fit_many_videos(self, videos, criterion, lr=1e-3, model_dir=None, label_id_to_text=None, save_model=True, n_iter=3000,
                        targeted_attack=False, target_class_id=None, reset_optimizer_per_video=False, **kw):
        """``VideoLearnerAdversarial.fit_many_videos`` (model.py:791-982): one single-video attack per (inputs, target, name);
        a video whose result file already shows a success is skipped, a placeholder (None) is written before the attack, the
        perturbation restarts from U(-1,1) * 0.005 and the clamp norm from ``max_norm`` (model.py:938-947).  Result files are
        ``<name>_@<class>.npy`` (model.py:917-921).  Returns {name: result dict or None}."""
        import os
        if self.per_clip:
            return self._fit_many_videos_batched(videos, criterion, lr, model_dir, label_id_to_text, save_model, n_iter, targeted_attack,
                                                 target_class_id, reset_optimizer_per_video=reset_optimizer_per_video, **kw)
        out = {}
        rng = np.random.default_rng(0)
        for inputs, target, name in videos:
            cls = (label_id_to_text[int(target[0])] if label_id_to_text is not None else str(int(target[0]))).replace(" ", "_")
            dest = os.path.join(model_dir, f"{os.path.basename(str(name))}_@{cls}.npy") if model_dir else None
            if dest and os.path.exists(dest):
                prev = np.load(dest, allow_pickle=True).tolist()
                if prev is None or np.array(prev["is_adversarial"]).any():
                    continue                                  # attacked before (None: the clean clip was misclassified / aborted run)
            elif dest and save_model:
                os.makedirs(model_dir, exist_ok=True)
                np.save(dest, None)
            self.pert_model.init_perturbation(((rng.random(self.pert_model.size, dtype=np.float32) * 2 - 1) * 0.005))
            self.pert_model.dynamic_max_norm = self.pert_model.max_norm
            if reset_optimizer_per_video:                       # (the reference carries ONE Adam state from video to video, SURVEY D.5: default)
                self.adam_m.zero_(); self.adam_v.zero_(); self.adam_t = 0
            res = self.fit_single_video_attack(inputs, target, criterion, lr=lr, n_iter=n_iter, targeted_attack=targeted_attack,
                                               target_class_id=target_class_id, **kw)
            out[str(name)] = res
            if res is not None and dest and save_model:
                res = dict(res, prob_clean_input=res["prob_clean_input"].cpu().numpy())
                np.save(dest, res, allow_pickle=True)
        return out


class VideoLearnerAdversarial(FlickerVideoResNet):
    """The reference's class name and constructor keywords (model.py:337-347: ``dataset, num_classes, base_model, sample_length,
    cyclic_pert, l_inf_pert_norm, attack_type, labaels_id_to_text`` [sic]) over the HIP engine.  ``dataset`` only supplies
    ``sample_length`` / batch size when it has them (the decord mp4 loader stays out of scope); ``weights`` = torchvision
    ``state_dict`` arrays or a ``.pth`` / ``.npz`` path (videoresnet_spec.load_weights) replaces ``pretrained=True``.
    ``.pert_model``, ``.model_name``, ``.results``, ``.fit``, ``.fit_many_videos``, ``.fit_single_video_attack`` as in the reference."""

    def __init__(self, dataset=None, num_classes=400, base_model="r2plus1d_18", sample_length=None, cyclic_pert=False, l_inf_pert_norm=0.1,
                 attack_type="flickering", labaels_id_to_text=None, weights=None, batch_size=None, image_size=112, dtype="bf16", device=0,
                 process_group=None):
        from . import videoresnet_spec as vs
        if weights is None:
            raise ValueError("weights: a torchvision state_dict ({name: array}) or a .pth / .npz path -- there is no network to download "
                             "the pretrained checkpoint the reference uses (model.py:421)")
        if isinstance(weights, (str, bytes)):
            weights = vs.load_weights(weights, base_model)
        if sample_length is None:
            sample_length = getattr(dataset, "sample_length", 16)
        if batch_size is None:
            batch_size = getattr(dataset, "batch_size", 1)
        super().__init__(base_model, weights, batch_size=batch_size, sample_length=sample_length, image_size=image_size, dtype=dtype,
                         device=device, l_inf_pert_norm=l_inf_pert_norm, cyclic_pert=cyclic_pert, num_classes=num_classes,
                         process_group=process_group, attack_type=attack_type)
        self.dataset, self.labaels_id_to_text = dataset, labaels_id_to_text
        self.results = {}

#!/usr/bin/env python3
"""Single-video flickering attack on I3D over a folder of .npy clips -- the MI355X counterpart of the reference's
i3d_adversarial_main_single_video_npy.py (same run_config.yml section, same result pickle).

    python scripts/i3d_adversarial_main_single_video_npy.py [run_config.yml] [--max-steps N] [--frames T]

Per clip (reference :115-337): clean prediction (skip if misclassified) -> delta, Adam re-initialised -> iterate until
step > MAX_NUM_STEP and the clip is adversarial -> pickle {keys of config.RESULT_KEYS}.  One device pass per iteration
(FlickerI3D.step) replaces the reference's four sess.run calls; past MAX_NUM_STEP the updated perturbation is verified with one extra
forward before the loop stops, so the saved perturbation / adv_video / last softmax describe the same, checked-adversarial state.  Weights: MODEL.WEIGHTS_NPZ ({variable name: array}); without
it seeded synthetic weights are used (the checkpoint is not distributed with the reference).
"""
import argparse
import glob
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import config as cfgmod, i3d_spec  # noqa: E402
from flickering_adversarial_video_amd.i3d_engine import FlickerI3D  # noqa: E402


def attack_clip(eng, x, label_id, c, max_steps, target_id=None, log_every=100):
    """returns the result dict (without file-level fields) or None if the clean clip is misclassified"""
    clean = eng(x, adv_flag=0)[0]
    if int(clean.argmax()) != label_id:
        return None
    targeted = bool(c.TARGETED_ATTACK)
    lab = torch.tensor([target_id if targeted else label_id], device=x.device)
    eng.reset_perturbation()
    beta3 = c.BETA_2                       # reference :98: beta_3 := BETA_2
    res = {k: [] for k in ("total_loss_l", "adv_loss_l", "reg_loss_l", "norm_reg_loss_l", "diff_norm_reg_loss_l", "perturbation", "softmax")}
    cyc = float(bool(c.CYCLIC_ATTACK))

    def is_adv(sm):
        am = int(sm.argmax())
        return am == target_id if targeted else am != label_id

    # reference :37-44: the loss node is chosen once, from the same three config switches
    if c.IMPROVE_ADV_LOSS:
        adversarial_loss = eng.improve_adversarial_loss(margin=c.PROB_MARGIN, targeted=targeted, logits=bool(c.USE_LOGITS))
    else:
        adversarial_loss = eng.ce_adversarial_loss(targeted=targeted)
    step, last, pending = 0, None, False
    while True:
        # one device pass: scalars of the CURRENT delta (what the reference fetches together with train_op, :213-215), then the update
        r = eng.step(x, lab, lr=1e-3, beta0=c.LAMBDA, beta1=c.BETA_1, beta2=c.BETA_2, beta3=beta3, cyclic=cyc, **adversarial_loss)
        h = r.host()                       # one host sync per step, like the reference's fetches
        # the reference's second sess.run (:217) evaluates is_adversarial / softmax AFTER the update: the forward of step k+1 is
        # that evaluation for step k (same delta), so the softmax list is filled with one step of lag ...
        if pending:
            res["softmax"].append(h["softmax"].copy())
        pending = True
        for k, src in (("total_loss_l", "total_loss"), ("adv_loss_l", "adv_loss"), ("reg_loss_l", "reg_loss"),
                       ("norm_reg_loss_l", "norm_reg"), ("diff_norm_reg_loss_l", "diff_norm_reg")):
            res[k].append(float(h[src]))
        res["perturbation"].append(eng.perturbation.cpu().numpy().copy())      # sess.run(perturbation) after the update, :305
        last = h
        step += 1
        if log_every and step % log_every == 0:
            print(f"  step {step}: total {h['total_loss']:.5f} adv {h['adv_loss']:.5f} thick {h['thickness_relative']:.3f}% "
                  f"rough {h['roughness_relative']:.3f}% adversarial (before this update) {bool(h['is_adversarial'])}", flush=True)
        give_up = step > 20 * max_steps + 100       # the reference loops forever on a robust clip; bound it
        if step > max_steps or give_up:
            # ... and past MAX_NUM_STEP the updated delta is verified with its own forward before the loop may stop (:313): the
            # perturbation that is saved is the one that was checked
            sm = eng(x, adv_flag=1, cyclic=cyc)[0].cpu().numpy()
            res["softmax"].append(sm)
            pending = False
            if is_adv(sm):
                break
            if give_up:
                print("  giving up: not adversarial", flush=True)
                break
    # thickness / roughness of the SAVED (verified) perturbation, kinetics_i3d_utils.py:196-200 on the raw variable
    d = res["perturbation"][-1].astype(np.float64)
    res.update(correct_cls_prob=float(clean[label_id]), softmax_init=clean.cpu().numpy(), total_steps=step,
               fatness=float(np.abs(d).mean() / 2 * 100), smoothness=float(np.abs(d - np.roll(d, 1, 0)).mean() / 2 * 100),
               adv_video=None, beta_0=c.LAMBDA, beta_1=c.BETA_1, beta_2=c.BETA_2, beta_3=beta3)
    return res


def attack_clips_batched(eng, videos, c, max_steps, target_id=None, log_every=100):
    """The same per-video procedure as ``attack_clip`` for B videos AT ONCE (``eng`` built with ``per_clip_delta=True``): every slot of
    the batch attacks its own video with its own perturbation, Adam state and step counter; a slot whose video is done (past
    MAX_NUM_STEP and verified adversarial, reference :211-337) is handed the next video, so the batch stays full until the list
    runs out.  One device pass per iteration serves all B videos -- 2-3x the clip-iterations per second of the one-by-one loop
    (the bs-1 plan leaves most CUs idle) -- and each video's trajectory is the one it has alone (bitwise in fp32).

    ``videos``: iterable of (tag, clip float32 [1,T,224,224,3], label_id).  Yields (tag, result dict | None) in completion order."""
    B, T = eng.B, eng.T
    targeted = bool(c.TARGETED_ATTACK)
    if bool(c.CYCLIC_ATTACK):
        raise ValueError("CYCLIC_ATTACK draws one roll per run in the reference: use the one-by-one loop (--batch 1)")
    beta3 = c.BETA_2
    dev = eng.eps_rgb.device
    x = torch.zeros((B, T, 224, 224, 3), dtype=torch.float32, device=dev)
    labels = torch.zeros(B, dtype=torch.int64, device=dev)
    if c.IMPROVE_ADV_LOSS:
        adversarial_loss = eng.improve_adversarial_loss(margin=c.PROB_MARGIN, targeted=targeted, logits=bool(c.USE_LOGITS))
    else:
        adversarial_loss = eng.ce_adversarial_loss(targeted=targeted)
    it = iter(videos)
    slots = [None] * B
    keys = ("total_loss_l", "adv_loss_l", "reg_loss_l", "norm_reg_loss_l", "diff_norm_reg_loss_l", "perturbation", "softmax")

    def refill(b):
        """next correctly classified video into slot b; misclassified ones are reported as None (reference :139-141); False: list exhausted"""
        skipped = []
        while True:
            nxt = next(it, None)
            if nxt is None:
                slots[b] = None
                eng.active[b] = 0
                return skipped
            tag, clip, label_id = nxt
            x[b].copy_(torch.from_numpy(np.ascontiguousarray(clip[0])))
            eng.reset_clip(b)
            clean = eng(x, adv_flag=0)[b]
            if int(clean.argmax()) != label_id:
                skipped.append((tag, None))
                continue
            labels[b] = target_id if targeted else label_id
            slots[b] = dict(tag=tag, label=label_id, clean=clean.clone(), step=0, pending=False, res={k: [] for k in keys})
            return skipped

    def is_adv(b, sm):
        am = int(sm.argmax())
        return am == target_id if targeted else am != slots[b]["label"]

    for b in range(B):
        yield from refill(b)
    while any(s is not None for s in slots):
        r = eng.step(x, labels, lr=1e-3, beta0=c.LAMBDA, beta1=c.BETA_1, beta2=c.BETA_2, beta3=beta3, **adversarial_loss)
        h = r.host()                                        # one host sync per step for the whole batch
        pert = eng.perturbation.cpu().numpy()               # [B,T,1,1,3] after the update
        verify = []
        for b, st in enumerate(slots):
            if st is None:
                continue
            res = st["res"]
            if st["pending"]:
                res["softmax"].append(h["softmax"][b].copy())
            st["pending"] = True
            adv_b, reg_b = float(h["adv_loss"][b]), float(h["reg_loss"][b])
            res["total_loss_l"].append(adv_b + c.LAMBDA * reg_b); res["adv_loss_l"].append(adv_b); res["reg_loss_l"].append(reg_b)
            res["norm_reg_loss_l"].append(float(h["norm_reg"][b])); res["diff_norm_reg_loss_l"].append(float(h["diff_norm_reg"][b]))
            res["perturbation"].append(pert[b].copy())
            st["step"] += 1
            if log_every and st["step"] % log_every == 0:
                print(f"  [{st['tag']}] step {st['step']}: total {res['total_loss_l'][-1]:.5f} adv {adv_b:.5f} thick {float(h['thickness'][b]) / 2 * 100:.3f}% "
                      f"rough {float(h['roughness'][b]) / 2 * 100:.3f}%", flush=True)
            st["give_up"] = st["step"] > 20 * max_steps + 100
            if st["step"] > max_steps or st["give_up"]:
                verify.append(b)
        if verify:
            sm_all = eng(x, adv_flag=1).cpu().numpy()      # the updated perturbations are verified with their own forward (:313)
            for b in verify:
                st = slots[b]
                st["res"]["softmax"].append(sm_all[b])
                st["pending"] = False
                if not is_adv(b, sm_all[b]) and not st["give_up"]:
                    continue
                if not is_adv(b, sm_all[b]):
                    print(f"  [{st['tag']}] giving up: not adversarial", flush=True)
                res = st["res"]
                d = res["perturbation"][-1].astype(np.float64)
                res.update(correct_cls_prob=float(st["clean"][st["label"]]), softmax_init=st["clean"].cpu().numpy(), total_steps=st["step"],
                           fatness=float(np.abs(d).mean() / 2 * 100), smoothness=float(np.abs(d - np.roll(d, 1, 0)).mean() / 2 * 100),
                           adv_video=eng.adversarial_inputs_rgb[b:b + 1].cpu().numpy(), beta_0=c.LAMBDA, beta_1=c.BETA_1, beta_2=c.BETA_2, beta_3=beta3)
                yield st["tag"], res
                yield from refill(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="run_config.yml")
    ap.add_argument("--batch", type=int, default=1, help="videos attacked at once, each with its own perturbation (independent single-video "
                    "attacks batched; 1 = the reference's one-by-one loop)")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--dtype", default=None)
    a = ap.parse_args()
    cfg = cfgmod.load_config(a.config)
    c = cfg.SINGLE_VIDEO_ATTACK
    T = a.frames or cfg.MODEL.FRAMES
    classes = cfgmod.load_kinetics_classes(cfg.DATA.LABEL_MAP_PATH)
    W, wsrc = i3d_spec.load_i3d_weights(cfg.MODEL)
    print(f"I3D weights: {wsrc}", flush=True)
    os.makedirs(c.PKL_RESULT_PATH, exist_ok=True)
    target_id = classes.index(c.TARGETED_CLASS) if c.TARGETED_ATTACK else None
    max_steps = a.max_steps if a.max_steps is not None else c.MAX_NUM_STEP
    if a.batch > 1:
        eng = FlickerI3D(W, batch_size=a.batch, frames=T, dtype=a.dtype or cfg.MODEL.DTYPE, per_clip_delta=True)
        meta = {}

        def videos():
            for path in sorted(glob.glob(os.path.join(c.NPY_PATH, "*.npy"))):
                cls, label_id = cfgmod.label_from_npy_name(path, classes)
                clip = np.load(path)[0, -T:][None].astype(np.float32)
                meta[path] = (cls, label_id, clip)
                yield path, clip, label_id
        for path, res in attack_clips_batched(eng, videos(), c, max_steps, target_id):
            cls, label_id, clip = meta.pop(path)
            if res is None:
                print(f"{path}: class {cls!r} ({label_id}): clean clip is misclassified: skipped", flush=True)
                continue
            res.update(correct_cls=cls, correct_cls_id=label_id, rgb_sample=clip)
            out = os.path.join(c.PKL_RESULT_PATH, cfgmod.result_filename(cls, c.BETA_1, res["fatness"], res["smoothness"]))
            with open(out, "wb") as f:
                pickle.dump(res, f)
            print(f"{path}: class {cls!r} ({label_id}) -> {out}  ({res['total_steps']} steps, thickness {res['fatness']:.2f}% roughness {res['smoothness']:.2f}%)", flush=True)
        return
    eng = FlickerI3D(W, batch_size=1, frames=T, dtype=a.dtype or cfg.MODEL.DTYPE)
    for path in sorted(glob.glob(os.path.join(c.NPY_PATH, "*.npy"))):
        cls, label_id = cfgmod.label_from_npy_name(path, classes)
        clip = np.load(path)[0, -T:][None].astype(np.float32)          # reference :121
        x = torch.from_numpy(np.ascontiguousarray(clip)).cuda()
        print(f"{path}: class {cls!r} ({label_id})", flush=True)
        res = attack_clip(eng, x, label_id, c, max_steps, target_id)
        if res is None:
            print("  clean clip is misclassified: skipped", flush=True)
            continue
        res.update(correct_cls=cls, correct_cls_id=label_id, rgb_sample=clip)
        res["adv_video"] = eng.adversarial_inputs_rgb.cpu().numpy()        # sess.run(adversarial_inputs_rgb), :61,325 -- the apply kernel's output
        out = os.path.join(c.PKL_RESULT_PATH, cfgmod.result_filename(cls, c.BETA_1, res["fatness"], res["smoothness"]))
        with open(out, "wb") as f:
            pickle.dump(res, f)
        print(f"  -> {out}  ({res['total_steps']} steps, thickness {res['fatness']:.2f}% roughness {res['smoothness']:.2f}%)", flush=True)


if __name__ == "__main__":
    main()

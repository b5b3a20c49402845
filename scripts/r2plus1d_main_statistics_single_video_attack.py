#!/usr/bin/env python3
"""Single-video flickering attacks on a list of clips with per-video result files -- the MI355X counterpart of the reference's
r2plus1d_main_statistics_single_video_attack.py (knobs :28-48, `learner.fit_many_videos` :190-200, result files
model.py:917-921).  Clips come pre-decoded (no mp4 decoder here): `--videos-npz` holds `clips` [N,T,112,112,3] (uint8 or
normalised float32), `labels` [N] and optionally `names` [N]; class names from `--label-map` (one per line)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import videoresnet_spec as vs  # noqa: E402
from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet, Losses  # noqa: E402

BASE_MODEL = "r2plus1d_18"       # "mc3_18", "r2plus1d_18", "r3d_18"
USE_LOGITS = True
IMPROVE_LOSS = True
CYCLIC_PERT = False
ATTACK_TYPE = "flickering"
L_INF_PERT_NORM = 0.2
TARGETED_ATTACK = False
LR = 0.001
LAMBDA = 1.0
BETA_1 = 0.5
N_ITER = 3000                    # fit_single_video_attack(n_iter=3000), model.py:962


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos-npz", required=True)
    ap.add_argument("--label-map", default="")
    ap.add_argument("--weights-npz", "--weights", dest="weights_npz", default="",
                    help="victim weights: a torchvision state_dict (.pth / .pt, what model.py:421 downloads) or an .npz of the same names; "
                         "'' = seeded synthetic weights")
    ap.add_argument("--attack-type", default=ATTACK_TYPE, choices=["flickering", "L12"], help="L12: dense [3,T,H,W] perturbation (model.py:380-384)")
    ap.add_argument("--results-root", default=os.path.join(os.getcwd(), "results"))
    ap.add_argument("--base-model", default=BASE_MODEL)
    ap.add_argument("--n-iter", type=int, default=N_ITER)
    ap.add_argument("--restart-after", type=int, default=3000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=1, help="videos attacked at once, each with its own perturbation / clamp bound / Adam state "
                    "(flickering attack; 1 = the reference's one-by-one loop)")
    ap.add_argument("--reset-optimizer-per-video", action="store_true", help="fresh Adam state for every video (the reference carries one "
                    "state from video to video, model.py:946; with --batch > 1 the carried state is per batch slot)")
    a = ap.parse_args()
    z = np.load(a.videos_npz, allow_pickle=True)
    clips, labels = z["clips"], z["labels"].astype(np.int64)
    names = [str(n) for n in z["names"]] if "names" in z else [f"video_{i:05d}" for i in range(len(clips))]
    if clips.dtype == np.uint8:
        clips = (clips.astype(np.float32) / 255.0 - np.array(vs.DEFAULT_MEAN, np.float32)) / np.array(vs.DEFAULT_STD, np.float32)
    clips = np.ascontiguousarray(clips, dtype=np.float32)
    classes = [l.strip() for l in open(a.label_map)] if a.label_map else None
    W = vs.load_weights(a.weights_npz) if a.weights_npz else vs.synthetic_weights(a.base_model, 42)
    learner = FlickerVideoResNet(a.base_model, W, batch_size=a.batch, sample_length=clips.shape[1], image_size=clips.shape[2], dtype=a.dtype,
                                 l_inf_pert_norm=L_INF_PERT_NORM, cyclic_pert=CYCLIC_PERT, attack_type=a.attack_type, per_clip=a.batch > 1)
    dest = os.path.join(a.results_root, learner.model_name, "single_video_attack", a.attack_type,
                        f"linf_{L_INF_PERT_NORM}_lambda_{LAMBDA}_beta1_{BETA_1}_")
    crit = Losses(beta_1=BETA_1, lambda_=LAMBDA, targeted=TARGETED_ATTACK, improve_loss=IMPROVE_LOSS, logits=USE_LOGITS, attack_type=a.attack_type)
    videos = ((torch.from_numpy(clips[i:i + 1]).cuda(), torch.from_numpy(labels[i:i + 1]).cuda(), names[i]) for i in range(len(clips)))
    out = learner.fit_many_videos(videos, crit, lr=LR, model_dir=dest, label_id_to_text=classes, n_iter=a.n_iter, restart_after=a.restart_after,
                                  reset_optimizer_per_video=a.reset_optimizer_per_video)
    for name, r in out.items():
        if r is None:
            print(f"{name}: clean clip misclassified, skipped")
        else:
            print(f"{name}: {len(r['loss/total'])} iterations, adversarial {bool(r['is_adversarial'][-1])}, thickness "
                  f"{r['perturbation/thickness'][-1]:.4f}, roughness {r['perturbation/roughness'][-1]:.4f}, restarts {r['restarts']}", flush=True)


if __name__ == "__main__":
    main()

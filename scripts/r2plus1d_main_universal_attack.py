#!/usr/bin/env python3
"""Universal flickering attack on torchvision VideoResNets -- the MI355X counterpart of the reference's
r2plus1d_main_universal_attack.py (constants, destination folder, resume rules and the `learner.fit` call follow it:
r2plus1d_main_universal_attack.py:33-58,176-236).

What differs: clips come PRE-DECODED (the reference decodes mp4 with decord, which this image lacks): `--train-npz` /
`--val-npz` hold `clips` ([N,T,112,112,3]; uint8 frames or float32 already normalised with dataset.py:28-29 mean / std)
and `labels` ([N] int).  Weights: `--weights-npz` with torchvision state_dict names, else seeded synthetic weights.
One process per GPU (torch.distributed.run); every rank takes its shard of the training clips, the perturbation is
replicated (parallel.py)."""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import parallel, videoresnet_spec as vs  # noqa: E402
from flickering_adversarial_video_amd.torch_attack import Adversarial_metrics, FlickerVideoResNet, Losses  # noqa: E402

# ---- the reference's knobs (r2plus1d_main_universal_attack.py:33-58) ----
EPOCHS = 22
LR = 0.001
LAMBDA = 1.0
BETA_1 = 0.5
L_INF_PERT_NORM = 0.1
BASE_MODEL = "mc3_18"            # "mc3_18", "r2plus1d_18", "r3d_18"
USE_LOGITS = False
IMPROVE_LOSS = True
CYCLIC_PERT = False
ATTACK_TYPE = "flickering"
TARGETED_ATTACK = False
INIT_PERT_FROM_LAST_CKPT = True
CONTINUE_TRAIN = True
MODEL_INPUT_SIZE = 16            # frames per clip for the three VideoResNets
BATCH_SIZE = 8                   # BATCH_SIZE_ARRAY[1] (one device per process here)


def load_clips(path):
    z = np.load(path)
    clips, labels = z["clips"], z["labels"].astype(np.int64)
    if clips.dtype == np.uint8:      # get_normalize_transforms (dataset.py:212-243): /255, mean / std
        clips = (clips.astype(np.float32) / 255.0 - np.array(vs.DEFAULT_MEAN, np.float32)) / np.array(vs.DEFAULT_STD, np.float32)
    return np.ascontiguousarray(clips, dtype=np.float32), labels


def loader(clips, labels, batch_size, rank=0, world=1):
    """batches of (clip [B,T,H,W,3] on the GPU, label, None), dropping the ragged tail; ranks take alternating batches"""
    nb = len(clips) // batch_size
    for i in range(rank, nb - nb % world if world > 1 else nb, world):
        sl = slice(i * batch_size, (i + 1) * batch_size)
        yield torch.from_numpy(clips[sl]).cuda(), torch.from_numpy(labels[sl]).cuda(), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-npz", required=True)
    ap.add_argument("--val-npz", required=True)
    ap.add_argument("--weights-npz", "--weights", dest="weights_npz", default="",
                    help="victim weights: a torchvision state_dict (.pth / .pt, what model.py:421 downloads) or an .npz of the same names; "
                         "'' = seeded synthetic weights")
    ap.add_argument("--attack-type", default=ATTACK_TYPE, choices=["flickering", "L12"], help="L12: dense [3,T,H,W] perturbation (model.py:380-384)")
    ap.add_argument("--results-root", default=os.path.join(os.getcwd(), "results"))
    ap.add_argument("--base-model", default=BASE_MODEL)
    ap.add_argument("--epochs", type=int, default=EPOCHS)
    ap.add_argument("--batch-size", type=int, default=BATCH_SIZE)
    ap.add_argument("--lr", type=float, default=LR)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--gpus", type=int, default=None, help="data-parallel ranks, one process per GPU (the reference's DEVICES_IDS, "
                    "r2plus1d_main_universal_attack.py:30-33); without a launcher in the environment the script starts them itself")
    a = ap.parse_args()
    if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # before anything touches the GPU
        sys.exit(parallel.launch_ranks(a.gpus, __file__, sys.argv[1:]))
    world, rank, local_rank = parallel.ranks_from_env(a.gpus)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    xtr, ytr = load_clips(a.train_npz)
    xva, yva = load_clips(a.val_npz)
    T, HW = xtr.shape[1], xtr.shape[2]
    W = vs.load_weights(a.weights_npz) if a.weights_npz else vs.synthetic_weights(a.base_model, 42)
    learner = FlickerVideoResNet(a.base_model, W, batch_size=a.batch_size, sample_length=T, image_size=HW, dtype=a.dtype,
                                 device=local_rank, l_inf_pert_norm=L_INF_PERT_NORM, cyclic_pert=CYCLIC_PERT, attack_type=a.attack_type)
    dest = os.path.join(a.results_root, learner.model_name, "generalization", "universal", "val_test", f"all_cls_shuffle_{a.attack_type}",
                        f"t_{len(xtr)}_v_{len(xva)}_linf_{L_INF_PERT_NORM}_lambda_{LAMBDA}_beta1_{BETA_1}_")
    start_epoch = 1
    ckpts = sorted(glob.glob(os.path.join(dest, "*.npy")), key=os.path.getmtime)
    if INIT_PERT_FROM_LAST_CKPT and ckpts:        # r2plus1d_main_universal_attack.py:199-208
        learner.pert_model.init_perturbation(np.load(ckpts[-1], allow_pickle=True)[-1]["valid/perturbation"])
        print("Success! init from last ckpt")
    if CONTINUE_TRAIN and ckpts:                  # :210-219
        start_epoch = int(ckpts[-1].split("_")[-1].split(".")[0]) + 1
        print(f"Success! to continue from last epoch. init with {start_epoch}")
    crit = Losses(beta_1=BETA_1, lambda_=LAMBDA, targeted=TARGETED_ATTACK, improve_loss=IMPROVE_LOSS, logits=USE_LOGITS, attack_type=a.attack_type)

    class Loaders(dict):                          # fresh iterators every epoch
        def __getitem__(self, phase):
            x, y = (xtr, ytr) if phase == "train" else (xva, yva)
            return loader(x, y, a.batch_size, rank if phase == "train" else 0, world if phase == "train" else 1)
    results = learner.fit(Loaders(), crit, Adversarial_metrics(targeted=TARGETED_ATTACK), lr=a.lr, epochs=a.epochs, model_dir=dest if rank == 0 else None,
                          model_name=learner.model_name, save_model=rank == 0, start_epoch=start_epoch)
    if rank == 0:
        for e, r in enumerate(results, start_epoch):
            print(f"epoch {e}: train loss {r['train/loss']:.5f} fooling {r['train/fooling_ratio']:.4f} | valid loss {r['valid/loss']:.5f} "
                  f"fooling {r['valid/fooling_ratio']:.4f} | thickness {r['valid/pert_thickness']:.5f} roughness {r['valid/pert_roughness']:.5f}", flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Universal / single-class-generalisation flickering attack on I3D over uint8 TFRecords, data-parallel over the GPUs of one
node -- the MI355X counterpart of the reference's i3d_adversarial_main_universal.py and
i3d_adversarial_main_single_class_gen.py (sections UNIVERSAL_ATTACK / CLASS_GEN_ATTACK of run_config.yml).

    python scripts/i3d_adversarial_main_universal.py [run_config.yml] [--section CLASS_GEN_ATTACK]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/i3d_adversarial_main_universal.py ...

Every rank reads its share of the records (round-robin), keeps the frozen weights and an identical replica of
(delta, Adam state); per step ONE RCCL all-reduce of the (T x 3) delta-gradient (+ 3 loss scalars).  Per epoch: fooling-rate
evaluation on the validation records (kinetics_i3d.evaluate), `res.pkl` and a resumable `model_step_XXXXX.npz` checkpoint
(delta, Adam m, v, t -- the reference's TF Saver format cannot be reproduced without TensorFlow).
The reference's class-gen loop never terminates and its universal script ignores BATCH_SIZE: MAX_NUM_STEP and
BATCH_SIZE are honoured here.
"""
import argparse
import glob
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd import config as cfgmod, i3d_spec, parallel, tb_events, tf_checkpoint, tfrecord_io as tio  # noqa: E402
from flickering_adversarial_video_amd.i3d_engine import FlickerI3D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="run_config.yml")
    ap.add_argument("--section", default="UNIVERSAL_ATTACK", choices=["UNIVERSAL_ATTACK", "CLASS_GEN_ATTACK"])
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--summary-steps", type=int, default=50, help="TensorBoard scalars every N steps (reference: save_steps=50)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--dtype", default=None)
    a = ap.parse_args()
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cfg = cfgmod.load_config(a.config)
    c = cfg[a.section]
    dense = not c.get("FLICKERING_ATTACK", True)      # False: dense "sparse perturbations" baseline, regulariser beta1 * L12
    T = a.frames or cfg.MODEL.FRAMES
    B = int(c.BATCH_SIZE)
    classes = cfgmod.load_kinetics_classes(cfg.DATA.LABEL_MAP_PATH)
    target_id = classes.index(c.TARGETED_CLASS) if c.TARGETED_ATTACK else None
    W, wsrc = i3d_spec.load_i3d_weights(cfg.MODEL)
    if rank == 0:
        print(f"I3D weights: {wsrc}", flush=True)
    eng = FlickerI3D(W, batch_size=B, frames=T, dtype=a.dtype or cfg.MODEL.DTYPE, device=local_rank, dense_delta=dense,
                     cyclic_flag_default_c=float(bool(c.CYCLIC_ATTACK)), cyclic_pert_flag_default_c=float(bool(c.get("CYCLIC_PERTURBATION_ATTACK", False))))
    train_files = tio.list_tfrecords(c.TF_RECORDS_TRAIN_PATH, c.get("NUM_OF_TRAIN_TF_RECORDS"))
    val_files = tio.list_tfrecords(c.TF_RECORDS_VAL_PATH, c.get("NUM_OF_VAL_TF_RECORDS"))
    if not train_files:
        raise FileNotFoundError(f"no *.tfrecords under {c.TF_RECORDS_TRAIN_PATH}")
    out_dir = c.PKL_RESULT_PATH
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    # resume: newest model_step_XXXXX.npz (the reference parses the step from the checkpoint name, single_class_gen.py:192-197)
    step = 0
    ckpts = sorted(glob.glob(os.path.join(out_dir, "model_step_*.npz")))
    if ckpts:
        ck = np.load(ckpts[-1])
        eng.reset_perturbation(ck["delta"])
        eng.adam_m.copy_(torch.from_numpy(ck["m"])); eng.adam_v.copy_(torch.from_numpy(ck["v"])); eng.adam_t = int(ck["t"])
        step = int(ck["step"])
        if rank == 0:
            print(f"resumed from {ckpts[-1]} at step {step}", flush=True)
    max_steps = a.max_steps if a.max_steps is not None else int(c.MAX_NUM_STEP)
    beta3 = c.BETA_2

    def to_dev(batch):
        x, y = batch
        y = torch.from_numpy(y).cuda()
        return torch.from_numpy(x).cuda(), (torch.full_like(y, target_id) if c.TARGETED_ATTACK else y)

    def evaluate():
        it = (to_dev(b) for b in tio.batches(val_files, B, T, rank, world))
        return eng.evaluate(it, bool(c.TARGETED_ATTACK), target_id, cyclic=float(bool(c.CYCLIC_ATTACK)))

    hist = {k: [] for k in ("total_loss_l", "adv_loss_l", "reg_loss_l", "thickness_l", "roughness_l", "fool_rate")}
    if val_files:
        rate, nval = evaluate()
        hist["fool_rate"].append(rate)
        if rank == 0:
            print(f"initial fooling rate {rate:.4f} over {nval} correctly classified validation clips", flush=True)
    epoch = 0
    # TensorBoard scalars under <out>/train every 50 steps (SummarySaverHook(save_steps=50), i3d_adversarial_main_universal.py:198-201)
    tb = tb_events.SummaryWriter(os.path.join(out_dir, "train")) if rank == 0 and not dense else None
    while step < max_steps:
        t0, nb = time.time(), 0
        for batch in tio.batches(train_files, B, T, rank, world):
            x, y = to_dev(batch)
            r = eng.step(x, y, lr=1e-3, beta0=c.LAMBDA, beta1=c.BETA_1, beta2=c.BETA_2, beta3=beta3, margin=c.PROB_MARGIN,
                         targeted=bool(c.TARGETED_ATTACK), use_logits=bool(c.USE_LOGITS), improve_loss=bool(c.IMPROVE_ADV_LOSS))
            step += 1; nb += 1
            if tb is not None and step % a.summary_steps == 0:
                tb.add_step_result(step, r.host(), beta0=c.LAMBDA)
                tb.flush()
            if step % 10 == 0 or step == max_steps:
                h = r.host()
                for k, s_ in (("total_loss_l", "total_loss"), ("adv_loss_l", "adv_loss"), ("reg_loss_l", "reg_loss"),
                              ("thickness_l", "thickness_relative"), ("roughness_l", "roughness_relative")):
                    hist[k].append(float(h[s_]))
                if rank == 0:
                    print(f"step {step}: total {h['total_loss']:.5f} adv {h['adv_loss']:.5f} reg {h['reg_loss']:.6f} thickness "
                          f"{h['thickness_relative']:.3f}% roughness {h['roughness_relative']:.3f}% prob_to_min {h['prob_to_min']:.4f}", flush=True)
            if step >= max_steps:
                break
        epoch += 1
        if nb == 0:
            raise RuntimeError("the training records do not fill one batch per rank")
        if val_files:
            rate, nval = evaluate()
            hist["fool_rate"].append(rate)
        if rank == 0:
            dt = time.time() - t0
            print(f"epoch {epoch}: {nb} steps in {dt:.1f}s ({nb * B * world / dt:.1f} clips/s incl. input)"
                  + (f"; fooling rate {hist['fool_rate'][-1]:.4f}" if val_files else ""), flush=True)
            np.savez(os.path.join(out_dir, f"model_step_{step:05d}.npz"), delta=eng.perturbation.cpu().numpy(), m=eng.adam_m.cpu().numpy(),
                     v=eng.adam_v.cpu().numpy(), t=eng.adam_t, step=step)
            # the same state as a TensorFlow checkpoint (saver.save(sess, 'model_step_%05d'), the reference's format)
            tf_checkpoint.write_bundle(os.path.join(out_dir, f"model_step_{step:05d}"),
                                       {"eps": eng.perturbation.cpu().numpy(), "eps/Adam": eng.adam_m.cpu().numpy().reshape(eng.perturbation.shape),
                                        "eps/Adam_1": eng.adam_v.cpu().numpy().reshape(eng.perturbation.shape), "global_step": np.array(step, np.int64)})
            with open(os.path.join(out_dir, "res.pkl"), "wb") as f:
                pickle.dump(dict(hist, perturbation=eng.perturbation.cpu().numpy(), total_steps=step, beta_0=c.LAMBDA, beta_1=c.BETA_1,
                                 beta_2=c.BETA_2, beta_3=beta3), f)
        if world > 1:
            torch.distributed.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

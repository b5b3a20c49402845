"""Shared driver of the two data-set attacks on I3D -- the MI355X counterpart of the hot loops of the reference's
``i3d_adversarial_main_single_class_gen.py`` (:218-373, section CLASS_GEN_ATTACK) and ``i3d_adversarial_main_universal.py``
(:45-203,300-380, section UNIVERSAL_ATTACK) -- over uint8 TFRecords, data-parallel over the GPUs of one node.

Every rank keeps the frozen weights and an identical replica of (delta, Adam state) and reads its slice of every global batch
(tfrecord_io.batches: equal batch counts per rank by construction); per step ONE RCCL all-reduce of the (T x 3) delta-gradient
(+ 3 loss scalars) -- the dense baseline (FLICKERING_ATTACK: False) all-reduces the dense gradient.

What each mode keeps of its reference script:

* class-gen: per-step lists and the ``res.pkl`` keys of :353-367, fooling rate before the first step and after every pass over the
  records (:200-208,338-344), a TensorFlow ``Saver`` checkpoint ``model_step_%05d`` at the start and after every pass (:214,373),
  resume from the newest one with the step parsed from its name (:192-197).
* universal: Estimator layout -- ``<PKL_RESULT_PATH>/<attack_type>/<class>_t<N>_v<M>_`` model directory (:300-302),
  ``model.ckpt-<step>`` checkpoints every 100 steps keeping 5 (:314-321), TensorBoard scalars under ``train`` every 50 steps
  (:198-201) and the ``ACC: 1- FOOLING_RATIO`` evaluation metric (:139-161) under ``eval``.

Checkpoints are written in the reference's own format and names (variables live in ``tf.variable_scope('RGB')``,
kinetics_i3d_utils.py:87,100; Adam slots ``<var>/Adam``, ``<var>/Adam_1`` and the optimizer's ``beta1_power`` / ``beta2_power``),
so a TF-1.15 ``Saver`` / Estimator restore of the perturbation and its optimizer state is complete.  The reference's class-gen
loop never terminates and its universal script ignores BATCH_SIZE: MAX_NUM_STEP and BATCH_SIZE are honoured here.
"""
import argparse
import glob
import os
import pickle
import re
import sys
import time

import numpy as np
import torch

from . import config as cfgmod, i3d_spec, ops, parallel, prefetch, tb_events, tf_checkpoint, tfrecord_io as tio
from .i3d_engine import FlickerI3D

ADAM_B1, ADAM_B2 = 0.9, 0.999        # tf.train.AdamOptimizer defaults (i3d_adversarial_main_single_class_gen.py:83)


def checkpoint_tensors(eng, step, weights=None, with_global_step=False):
    """{variable name: array} of one reference-format checkpoint"""
    shp = tuple(eng.perturbation.shape)
    t = {"RGB/eps": eng.perturbation.cpu().numpy().reshape(shp),
         "RGB/eps/Adam": eng.adam_m.cpu().numpy().reshape(shp),
         "RGB/eps/Adam_1": eng.adam_v.cpu().numpy().reshape(shp),
         "beta1_power": np.array(ADAM_B1 ** eng.adam_t, np.float32),
         "beta2_power": np.array(ADAM_B2 ** eng.adam_t, np.float32)}
    if with_global_step:
        t["global_step"] = np.array(step, np.int64)
    if weights:
        t.update({k: np.asarray(v, np.float32) for k, v in weights.items()})
    return t


def restore(eng, prefix):
    """load (delta, Adam m, v, t) from a bundle written by ``checkpoint_tensors`` -- or by the reference itself.
    The Adam step count is recovered from beta1_power = 0.9 ** t."""
    ck = tf_checkpoint.read_bundle(prefix, names=lambda n: n.startswith("RGB/eps") or n in ("beta1_power", "beta2_power", "global_step"))
    eng.reset_perturbation(ck["RGB/eps"])
    if "RGB/eps/Adam" in ck:
        eng.adam_m.copy_(torch.from_numpy(np.ascontiguousarray(ck["RGB/eps/Adam"])).reshape(eng.adam_m.shape))
        eng.adam_v.copy_(torch.from_numpy(np.ascontiguousarray(ck["RGB/eps/Adam_1"])).reshape(eng.adam_v.shape))
    if "beta1_power" in ck and 0.0 < float(ck["beta1_power"]) < 1.0:
        eng.adam_t = int(round(np.log(float(ck["beta1_power"])) / np.log(ADAM_B1)))
    return ck


def latest_checkpoint(out_dir, pattern):
    """tf.train.latest_checkpoint: the prefix with the highest step among ``<out_dir>/<pattern><step>.index``"""
    best = None
    for f in glob.glob(os.path.join(out_dir, pattern + "*.index")):
        m = re.search(r"(\d+)\.index$", f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f[:-len(".index")])
    return best


def main(default_section, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="run_config.yml")
    ap.add_argument("--section", default=default_section, choices=["UNIVERSAL_ATTACK", "CLASS_GEN_ATTACK"])
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--summary-steps", type=int, default=50, help="TensorBoard scalars every N steps (reference: save_steps=50)")
    ap.add_argument("--checkpoint-steps", type=int, default=100, help="universal: checkpoint every N steps (reference: save_checkpoints_steps=100)")
    ap.add_argument("--log-every", type=int, default=None, help="steps between host read-backs of the scalars (class-gen: 1 like the "
                    "reference's per-step lists; universal: 10)")
    ap.add_argument("--no-weights-in-checkpoint", action="store_true", help="checkpoint only the perturbation and its Adam state "
                    "(the reference's Saver() also saves the 12.7 M frozen I3D weights every time)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--gpus", type=int, default=None, help="data-parallel ranks (one process per GPU, RCCL all-reduce of the delta-gradient); "
                    "without a launcher in the environment the script starts them itself")
    a = ap.parse_args(argv)
    universal = a.section == "UNIVERSAL_ATTACK"
    if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # before anything touches the GPU
        import __main__
        sys.exit(parallel.launch_ranks(a.gpus, __main__.__file__, sys.argv[1:] if argv is None else argv))
    world, rank, local_rank = parallel.ranks_from_env(a.gpus)
    backend = os.environ.get("FLK_DIST_BACKEND", "nccl")          # gloo: rehearse the multi-rank plumbing on fewer GPUs than ranks
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(backend)
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
                     cyclic_flag_default_c=float(bool(c.CYCLIC_ATTACK)),
                     cyclic_pert_flag_default_c=float(bool(c.get("CYCLIC_PERTURBATION_ATTACK", False))))
    train_files = tio.list_tfrecords(c.TF_RECORDS_TRAIN_PATH, c.get("NUM_OF_TRAIN_TF_RECORDS"))
    val_files = tio.list_tfrecords(c.TF_RECORDS_VAL_PATH, c.get("NUM_OF_VAL_TF_RECORDS"))
    if not train_files:
        raise FileNotFoundError(f"no *.tfrecords under {c.TF_RECORDS_TRAIN_PATH}")
    if tio.count_batches(train_files, B, world) == 0:      # identical on every rank: nobody is left waiting in a collective
        raise RuntimeError(f"the training records do not fill one global batch of {world} x {B} clips")
    out_dir = c.PKL_RESULT_PATH
    ck_pattern = "model_step_"
    if universal:
        # model_dir of the Estimator (i3d_adversarial_main_universal.py:285-302): <attack type>/<source class>_t<N>_v<M>_
        attack_type = "FLICKERING_ATTACK" if c.get("FLICKERING_ATTACK", True) else "SUP_ATTACK"
        tr = c.TF_RECORDS_TRAIN_PATH
        parts = (tr[-1] if isinstance(tr, (list, tuple)) else tr).split("/")
        source = parts[-2] if len(parts) >= 2 else parts[-1]
        nvid = int(c.get("NUM_OF_VID_EACH_TF_RECORDS") or 0)
        out_dir = os.path.join(out_dir, "{}/{}_t{}_v{}_".format(attack_type, source, nvid * int(c.get("NUM_OF_TRAIN_TF_RECORDS") or 0),
                                                                 nvid * int(c.get("NUM_OF_VAL_TF_RECORDS") or 0)))
        ck_pattern = "model.ckpt-"
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    if world > 1:
        torch.distributed.barrier()
    step = 0
    # WHAT to resume from is decided on rank 0 and broadcast: every rank restores the same state (or none does), and no rank can
    # glob a bundle that rank 0 is about to write
    last = latest_checkpoint(out_dir, ck_pattern) if rank == 0 else None
    old = sorted(glob.glob(os.path.join(out_dir, "model_step_*.npz"))) if rank == 0 and not last else []      # round-1 format, still accepted
    if world > 1:
        decided = [(last, old)]
        torch.distributed.broadcast_object_list(decided, src=0)
        last, old = decided[0]
    if last:
        step = last[0]
        restore(eng, last[1])
        if rank == 0:
            print(f"resumed from {last[1]} at step {step}", flush=True)
    elif old:
        ck = np.load(old[-1])
        eng.reset_perturbation(ck["delta"])
        eng.adam_m.copy_(torch.from_numpy(ck["m"]).reshape(eng.adam_m.shape)); eng.adam_v.copy_(torch.from_numpy(ck["v"]).reshape(eng.adam_v.shape))
        eng.adam_t, step = int(ck["t"]), int(ck["step"])
        if rank == 0:
            print(f"resumed from {old[-1]} at step {step}", flush=True)
    if world > 1:
        torch.distributed.barrier()                    # every rank has finished reading before rank 0 writes its first checkpoint
    max_steps = a.max_steps if a.max_steps is not None else int(c.MAX_NUM_STEP)
    beta3 = c.BETA_2                                   # reference: beta_3 := BETA_2 (single_class_gen.py:98; universal.py:131)
    log_every = a.log_every if a.log_every is not None else (10 if universal else 1)
    ck_weights = None if a.no_weights_in_checkpoint else W
    written = []

    def to_dev(batch, true_labels=False):
        x, y = batch
        if (y < 0).any() or (y >= len(classes)).any():
            raise ValueError(f"record labels outside [0, {len(classes)}): {y}")
        y = ops.mark_labels_validated(torch.from_numpy(y).cuda(), len(classes))      # range-checked above, on the host
        # training feeds the target class as the label of a targeted attack (single_class_gen.py:226-229); evaluation counts a
        # clip as valid when its CLEAN prediction equals the TRUE label (kinetics_i3d_utils.py:241-243)
        x = x if torch.is_tensor(x) else torch.from_numpy(x).cuda()        # (prefetch.DeviceBatches yields device tensors)
        if c.TARGETED_ATTACK and not true_labels:
            return x, ops.mark_labels_validated(torch.full_like(y, target_id), len(classes))      # target_id = classes.index(...)
        return x, y

    def evaluate():
        it = (to_dev(b, true_labels=True) for b in prefetch.DeviceBatches(val_files, B, T, rank, world))
        # the reference evaluates with cyclic=0 (single_class_gen.py:202,340)
        return eng.evaluate(it, bool(c.TARGETED_ATTACK), target_id, cyclic=0.0)

    def save_checkpoint():
        if rank != 0:
            return
        prefix = os.path.join(out_dir, f"{ck_pattern}{step:05d}" if not universal else f"{ck_pattern}{step}")
        tf_checkpoint.write_bundle(prefix, checkpoint_tensors(eng, step, ck_weights, with_global_step=universal))
        written.append(prefix)
        if universal:                                  # keep_checkpoint_max=5
            while len(written) > 5:
                for f in glob.glob(written.pop(0) + ".*"):
                    os.remove(f)

    lists = {k: [] for k in ("total_loss_l", "adv_loss_l", "reg_loss_l", "norm_reg_loss_l", "diff_norm_reg_loss_l", "laplacian_norm_reg_l",
                             "fatness", "smoothness", "prob_to_min_l", "prob_to_max_l", "perturbation", "fool_rate", "fool_rate_step")}
    tb = tb_events.SummaryWriter(os.path.join(out_dir, "train")) if rank == 0 and universal and not dense else None
    tb_eval = tb_events.SummaryWriter(os.path.join(out_dir, "eval")) if rank == 0 and universal and val_files else None

    def run_eval():
        rate, nval = evaluate()
        lists["fool_rate"].append(rate); lists["fool_rate_step"].append(step)
        if rank == 0:
            print("step: {:05d} ,fool_rate: {:.5f} ({} correctly classified validation clips)".format(step, rate, nval), flush=True)
            if tb_eval is not None:
                tb_eval.add_scalars(step, {"ACC: 1- FOOLING_RATIO": 1.0 - rate})
                tb_eval.flush()
        return rate

    if val_files and not universal:
        run_eval()                                     # single_class_gen.py:200-208
    if not universal and not last:
        save_checkpoint()                              # :214
    # i3d_adversarial_main_single_class_gen.py:39-46 / _universal.py:121-127: the loss node is chosen once from the config switches
    if c.IMPROVE_ADV_LOSS:
        adversarial_loss = eng.improve_adversarial_loss(margin=c.PROB_MARGIN, targeted=bool(c.TARGETED_ATTACK), logits=bool(c.USE_LOGITS))
    else:
        adversarial_loss = eng.ce_adversarial_loss(targeted=bool(c.TARGETED_ATTACK))
    train_batches = prefetch.DeviceBatches(train_files, B, T, rank, world)
    epoch = 0
    while step < max_steps:
        t0, nb = time.time(), 0
        for batch in train_batches:          # background reader -> pinned ring -> asynchronous H2D copy (prefetch.py)
            x, y = to_dev(batch)
            r = eng.step(x, y, lr=1e-3, beta0=c.LAMBDA, beta1=c.BETA_1, beta2=c.BETA_2, beta3=beta3, **adversarial_loss)
            step += 1; nb += 1
            if tb is not None and step % a.summary_steps == 0:
                tb.add_step_result(step, r.host(), beta0=c.LAMBDA)
                tb.flush()
            if step % log_every == 0 or step == max_steps:
                h = r.host()
                for k, s_ in (("total_loss_l", "total_loss"), ("adv_loss_l", "adv_loss"), ("reg_loss_l", "reg_loss"), ("norm_reg_loss_l", "norm_reg"),
                              ("diff_norm_reg_loss_l", "diff_norm_reg"), ("laplacian_norm_reg_l", "laplacian_norm_reg"),
                              ("fatness", "thickness_relative"), ("smoothness", "roughness_relative"), ("prob_to_min_l", "prob_to_min"),
                              ("prob_to_max_l", "prob_to_max")):
                    if s_ in h:
                        lists[k].append(float(h[s_]))
                if not dense and not universal:
                    lists["perturbation"].append(eng.perturbation.cpu().numpy().copy())      # sess.run(perturbation), :283
                if rank == 0:
                    print("Step: {:05d}, Total Loss: {:.5f}, Cls Loss: {:.5f}, Total Reg Loss: {:.5f}, prob_to_min: {:.6f}, prob_to_max: {:.6f}, "
                          "thickness: {:.2f} %, roughness: {:.2f} %".format(step, float(h["total_loss"]), float(h["adv_loss"]), float(h["reg_loss"]),
                                                                             float(h["prob_to_min"]), float(h["prob_to_max"]),
                                                                             float(h["thickness_relative"]), float(h["roughness_relative"])), flush=True)
            if universal and step % a.checkpoint_steps == 0:
                save_checkpoint()
            if step >= max_steps:
                break
        epoch += 1
        if val_files:
            run_eval()
        if rank == 0:
            dt = time.time() - t0
            print(f"epoch {epoch}: {nb} steps in {dt:.1f}s ({nb * B * world / dt:.1f} clips/s incl. input)", flush=True)
        save_checkpoint()
        if rank == 0 and not universal:
            res = {k: lists[k] for k in ("total_loss_l", "adv_loss_l", "reg_loss_l", "norm_reg_loss_l", "diff_norm_reg_loss_l", "perturbation",
                                         "fatness", "smoothness", "fool_rate")}
            res.update(total_steps=step, beta_1=c.BETA_1, beta_2=c.BETA_2)            # the keys of single_class_gen.py:353-367
            with open(os.path.join(out_dir, "res.pkl"), "wb") as f:
                pickle.dump(res, f)
        if world > 1:
            torch.distributed.barrier()
    if rank == 0 and universal:
        # Estimator PREDICT mode returns the perturbation (universal.py:112-117): keep it beside the checkpoints
        np.save(os.path.join(out_dir, "perturbation.npy"), eng.perturbation.cpu().numpy())
    if world > 1:
        torch.distributed.destroy_process_group()
    return eng

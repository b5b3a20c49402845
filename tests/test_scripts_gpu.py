"""BASELINE config 1 plumbing: the single-video entry point on a synthetic .npy clip (the reference's bartending.npy is not
distributed): clip container format, label-from-filename, skip rule, loop termination and the result pickle schema."""
import glob
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_video_script_end_to_end(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import config as cfgmod, i3d_spec
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    T = 16
    u8 = i3d_spec.synthetic_clip_u8(1, T + 4, seed=77)                       # longer than T: the script takes the LAST T frames
    clip = (u8.astype(np.float32) / 128 - 1)
    eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=T, dtype="f32")
    cls_id = int(eng(torch.from_numpy(clip[:, -T:]).cuda(), adv_flag=0).argmax())
    del eng
    classes = [f"class {i}" for i in range(400)]
    (tmp_path / "npy").mkdir()
    (tmp_path / "labels.txt").write_text("\n".join(classes))
    np.save(tmp_path / "npy" / f"rgb_0001@class_{cls_id}.npy", clip)
    np.save(tmp_path / "npy" / f"rgb_0002@class_{(cls_id + 1) % 400}.npy", clip)     # wrong label -> clean-misclassified -> skipped
    cfg = open(os.path.join(ROOT, "run_config.yml")).read()
    cfg = cfg.replace("'data/label_map.txt'", f"'{tmp_path}/labels.txt'").replace("NPY_PATH: 'data/videos_for_tests/npy/'", f"NPY_PATH: '{tmp_path}/npy/'", 1)
    cfg = cfg.replace("PKL_RESULT_PATH: 'result/videos_for_tests/npy/'", f"PKL_RESULT_PATH: '{tmp_path}/out/'")
    (tmp_path / "cfg.yml").write_text(cfg)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "i3d_adversarial_main_single_video_npy.py"), str(tmp_path / "cfg.yml"),
                        "--max-steps", "3", "--frames", str(T), "--dtype", "f32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "skipped" in r.stdout
    outs = os.listdir(tmp_path / "out")
    assert len(outs) == 1 and outs[0].startswith(f"class {cls_id}_beta1_0.5_th_") and outs[0].endswith("%.pkl")
    res = pickle.load(open(tmp_path / "out" / outs[0], "rb"))
    assert set(res) == set(cfgmod.RESULT_KEYS)
    n = res["total_steps"]
    assert n >= 4 and len(res["perturbation"]) == n and res["perturbation"][0].shape == (T, 1, 1, 3) and len(res["softmax"]) == n
    assert res["rgb_sample"].shape == (1, T, 224, 224, 3) and res["adv_video"].shape == (1, T, 224, 224, 3)
    assert res["correct_cls_id"] == cls_id and res["softmax_init"].shape == (400,)
    assert res["total_loss_l"][0] >= res["adv_loss_l"][0] and np.isfinite(res["total_loss_l"]).all()
    # adv_video = the engine's adversarial_inputs_rgb attribute (kinetics_i3d_utils.py:104-142: the apply kernel's own output) under the
    # final perturbation -- against the formula in numpy
    want = np.clip(clip[:, -T:] + np.clip(res["perturbation"][-1], -0.4, 0.4)[None], -1, 1)
    np.testing.assert_allclose(res["adv_video"], want, rtol=0, atol=1e-7)


def test_universal_script_on_tfrecords(tmp_path):
    """uint8 TFRecords -> class-generalisation loop (BASELINE config 4 plumbing, 1 GPU): steps, evaluation, checkpoint, resume"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec, tfrecord_io as tio
    T, B = 16, 2
    (tmp_path / "rec").mkdir()
    u8 = i3d_spec.synthetic_clip_u8(5, T + 2, seed=9)
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=T, dtype="f32")
    labels = [int(eng(torch.from_numpy(u8[i:i + 1, -T:]).cuda(), adv_flag=0).argmax()) for i in range(5)]   # correctly classified clips
    del eng
    tio.write_records(str(tmp_path / "rec" / "a.tfrecords"), [tio.make_example(u8[i], labels[i]) for i in range(5)], with_payload_crc=False)
    (tmp_path / "labels.txt").write_text("\n".join(f"class {i}" for i in range(400)))
    cfg = open(os.path.join(ROOT, "run_config.yml")).read().replace("'data/label_map.txt'", f"'{tmp_path}/labels.txt'")
    cfg = cfg.replace("['data/kinetics/database/tfrecord/test/hula hooping']", f"['{tmp_path}/rec']")
    cfg = cfg.replace("PKL_RESULT_PATH: 'result/generalization/model_gen_one_class/'", f"PKL_RESULT_PATH: '{tmp_path}/out/'")
    cfg = cfg.replace("BATCH_SIZE: 8\n    MAX_NUM_STEP: 10000\n    TARGETED_ATTACK: False\n    TARGETED_CLASS: 'javelin throw'", f"BATCH_SIZE: {B}\n    MAX_NUM_STEP: 10000\n    TARGETED_ATTACK: False\n    TARGETED_CLASS: 'javelin throw'", 1)
    # the victim's weights arrive as a TensorFlow checkpoint (MODEL.CKPT_PATH), read without TensorFlow
    from flickering_adversarial_video_amd import tf_checkpoint as tfc
    tfc.write_bundle(str(tmp_path / "ckpt" / "model.ckpt"), i3d_spec.synthetic_i3d_weights(42), with_crc=False)
    cfg = cfg.replace("'data/checkpoints/rgb_imagenet/model.ckpt'", f"'{tmp_path}/ckpt/model.ckpt'")
    (tmp_path / "cfg.yml").write_text(cfg)
    # the class-generalisation entry point (reference i3d_adversarial_main_single_class_gen.py)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "i3d_adversarial_main_single_class_gen.py"), str(tmp_path / "cfg.yml"),
           "--frames", str(T), "--dtype", "f32", "--no-weights-in-checkpoint"]
    r = subprocess.run(cmd + ["--max-steps", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Step: 00003" in r.stdout and "fool_rate" in r.stdout and "I3D weights: tf-checkpoint" in r.stdout
    # TensorFlow Saver checkpoints under the reference's names (variables live in tf.variable_scope('RGB'), kinetics_i3d_utils.py:87,100)
    from flickering_adversarial_video_amd import tf_checkpoint
    assert os.path.exists(tmp_path / "out" / "model_step_00000.index")                  # saved before the first step (:214)
    tfck = tf_checkpoint.read_bundle(str(tmp_path / "out" / "model_step_00003"), verify_crc=True)
    assert set(tfck) == {"RGB/eps", "RGB/eps/Adam", "RGB/eps/Adam_1", "beta1_power", "beta2_power"}
    assert tfck["RGB/eps"].shape == (T, 1, 1, 3) and np.abs(tfck["RGB/eps"]).max() > 0 and np.abs(tfck["RGB/eps/Adam_1"]).max() > 0
    assert float(tfck["beta1_power"]) == pytest.approx(0.9 ** 3, rel=1e-6) and float(tfck["beta2_power"]) == pytest.approx(0.999 ** 3, rel=1e-6)
    res = pickle.load(open(tmp_path / "out" / "res.pkl", "rb"))
    assert set(res) == {"total_loss_l", "adv_loss_l", "reg_loss_l", "norm_reg_loss_l", "diff_norm_reg_loss_l", "perturbation", "total_steps",
                        "beta_1", "beta_2", "fatness", "smoothness", "fool_rate"}      # i3d_adversarial_main_single_class_gen.py:353-367
    assert res["total_steps"] == 3 and len(res["total_loss_l"]) == 3 and len(res["perturbation"]) == 3 and res["perturbation"][0].shape == (T, 1, 1, 3)
    assert len(res["fool_rate"]) == 3           # before the first step + after each of the two passes (2 global batches per pass, 3 steps)
    # resume: delta, Adam m / v and the step count t = log(beta1_power) / log(0.9) come back from the newest bundle
    r = subprocess.run(cmd + ["--max-steps", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "resumed from" in r.stdout and "at step 3" in r.stdout and os.path.exists(tmp_path / "out" / "model_step_00005.index")
    tf5 = tf_checkpoint.read_bundle(str(tmp_path / "out" / "model_step_00005"))
    assert float(tf5["beta1_power"]) == pytest.approx(0.9 ** 5, rel=1e-6)
    # the checkpoint carries the whole optimiser state: write -> restore into a fresh engine -> identical (delta, m, v, t)
    from flickering_adversarial_video_amd import i3d_dataset_attack as ida
    e1 = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=T, dtype="f32")
    x1, y1 = torch.from_numpy(u8[:1, -T:]).cuda(), torch.tensor(labels[:1]).cuda()
    for _ in range(2):
        e1.step(x1, y1)
    tf_checkpoint.write_bundle(str(tmp_path / "rt" / "model_step_00002"), ida.checkpoint_tensors(e1, 2))
    assert ida.latest_checkpoint(str(tmp_path / "rt"), "model_step_") == (2, str(tmp_path / "rt" / "model_step_00002"))
    e2 = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=T, dtype="f32")
    ida.restore(e2, str(tmp_path / "rt" / "model_step_00002"))
    assert e2.adam_t == 2 and torch.equal(e2.eps_rgb, e1.eps_rgb) and torch.equal(e2.adam_m, e1.adam_m) and torch.equal(e2.adam_v, e1.adam_v)
    e1.step(x1, y1); e2.step(x1, y1)
    assert torch.equal(e2.eps_rgb, e1.eps_rgb)            # fp32 runs are bitwise reproducible (gather-form pool backward in fp32)


@pytest.mark.parametrize("variant", ["cyclic_pert", "dense"])
def test_universal_section(tmp_path, variant):
    """BASELINE config 5's entry: ``UNIVERSAL_ATTACK`` (Estimator layout) with CYCLIC_PERTURBATION_ATTACK: True, and with
    FLICKERING_ATTACK: False (the dense L12 baseline) -- model_dir naming, model.ckpt-<step> checkpoints with global_step and the
    frozen weights, TensorBoard train / eval scalars, perturbation.npy."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import i3d_spec, tb_events, tf_checkpoint, tfrecord_io as tio
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    T, B = 16, 2
    (tmp_path / "all_cls" ).mkdir()
    u8 = i3d_spec.synthetic_clip_u8(4, T, seed=19)
    eng = FlickerI3D(i3d_spec.synthetic_i3d_weights(42), batch_size=1, frames=T, dtype="f32")
    labels = [int(eng(torch.from_numpy(u8[i:i + 1]).cuda(), adv_flag=0).argmax()) for i in range(4)]
    del eng
    tio.write_records(str(tmp_path / "all_cls" / "a.tfrecords"), [tio.make_example(u8[i], labels[i]) for i in range(4)], with_payload_crc=False)
    (tmp_path / "labels.txt").write_text("\n".join(f"class {i}" for i in range(400)))
    cfg = open(os.path.join(ROOT, "run_config.yml")).read().replace("'data/label_map.txt'", f"'{tmp_path}/labels.txt'")
    cfg = cfg.replace("['data/kinetics/database/tfrecord/test_all_cls/']", f"['{tmp_path}/all_cls/']")
    cfg = cfg.replace("PKL_RESULT_PATH: 'result/generalization/universal_untargeted/'", f"PKL_RESULT_PATH: '{tmp_path}/out/'")
    cfg = cfg.replace("NUM_OF_VID_EACH_TF_RECORDS: 50\n    BATCH_SIZE: 8", f"NUM_OF_VID_EACH_TF_RECORDS: 50\n    BATCH_SIZE: {B}")
    if variant == "cyclic_pert":
        cfg = cfg.replace("CYCLIC_PERTURBATION_ATTACK: False", "CYCLIC_PERTURBATION_ATTACK: True")
    else:
        cfg = cfg.replace("FLICKERING_ATTACK: True ", "FLICKERING_ATTACK: False")
    assert f"BATCH_SIZE: {B}" in cfg
    (tmp_path / "cfg.yml").write_text(cfg)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "i3d_adversarial_main_universal.py"), str(tmp_path / "cfg.yml"), "--frames", str(T),
           "--dtype", "f32", "--max-steps", "4", "--summary-steps", "2", "--checkpoint-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    kind = "FLICKERING_ATTACK" if variant == "cyclic_pert" else "SUP_ATTACK"
    mdir = tmp_path / "out" / kind / "all_cls_t1050_v2000_"                 # <attack type>/<source class>_t<21*50>_v<40*50>_ (universal.py:300-302)
    assert mdir.is_dir(), os.listdir(tmp_path / "out")
    ck = tf_checkpoint.read_bundle(str(mdir / "model.ckpt-4"), verify_crc=False)
    shape = (T, 1, 1, 3) if variant == "cyclic_pert" else (T, 224, 224, 3)
    assert ck["RGB/eps"].shape == shape and int(ck["global_step"]) == 4 and ck["RGB/eps/Adam"].shape == shape
    assert "RGB/inception_i3d/Conv3d_1a_7x7/conv_3d/w" in ck           # Saver() of the reference saves every variable, the frozen net included
    assert os.path.exists(mdir / "model.ckpt-2.index") and np.load(mdir / "perturbation.npy").shape == shape
    assert np.isfinite(ck["RGB/eps"]).all() and np.abs(ck["RGB/eps"]).max() > 0
    ev = tb_events.read_scalars(glob.glob(str(mdir / "eval" / "events.out.tfevents.*"))[0])
    assert ev and "ACC: 1- FOOLING_RATIO" in ev[-1][1] and 0.0 <= ev[-1][1]["ACC: 1- FOOLING_RATIO"] <= 1.0
    if variant == "cyclic_pert":
        tr = tb_events.read_scalars(glob.glob(str(mdir / "train" / "events.out.tfevents.*"))[0])
        assert [s_ for s_, _ in tr] == [2, 4] and set(tb_events.SCALAR_TAGS) <= set(tr[0][1]) and np.isfinite(list(tr[0][1].values())).all()


def test_r2plus1d_universal_script(tmp_path):
    """pre-decoded clips -> VideoResNet universal attack epochs: result files named like the reference's, resume from the last"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet
    T, N = 8, 4
    u8 = np.random.default_rng(3).integers(0, 256, (N, T, 112, 112, 3), dtype=np.uint8)
    eng = FlickerVideoResNet("r3d_18", vs.synthetic_weights("r3d_18", 42), batch_size=2, sample_length=T, dtype="f32")
    norm = ((u8.astype(np.float32) / 255.0 - np.array(vs.DEFAULT_MEAN, np.float32)) / np.array(vs.DEFAULT_STD, np.float32)).astype(np.float32)
    labels = np.concatenate([eng.logits(torch.from_numpy(norm[i:i + 2]).cuda(), False).argmax(1).cpu().numpy() for i in (0, 2)])
    del eng
    np.savez(tmp_path / "train.npz", clips=u8, labels=labels)
    np.savez(tmp_path / "val.npz", clips=norm[:2], labels=labels[:2])            # float32 clips are taken as normalised
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "r2plus1d_main_universal_attack.py"), "--train-npz", str(tmp_path / "train.npz"),
           "--val-npz", str(tmp_path / "val.npz"), "--results-root", str(tmp_path / "results"), "--base-model", "r3d_18", "--batch-size", "2",
           "--dtype", "f32"]
    r = subprocess.run(cmd + ["--epochs", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    dest = glob.glob(str(tmp_path / "results" / "r3d_18" / "generalization" / "universal" / "val_test" / "all_cls_shuffle_flickering" / "t_4_v_2_*"))
    assert len(dest) == 1 and sorted(os.path.basename(f) for f in glob.glob(dest[0] + "/*.npy")) == ["r3d_18_001.npy", "r3d_18_002.npy"]
    res = np.load(os.path.join(dest[0], "r3d_18_002.npy"), allow_pickle=True)
    assert len(res) == 2 and res[-1]["valid/perturbation"].shape == (3, T, 1, 1) and 0.0 <= res[-1]["valid/fooling_ratio"] <= 1.0
    r = subprocess.run(cmd + ["--epochs", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "init from last ckpt" in r.stdout and "continue from last epoch. init with 3" in r.stdout
    assert os.path.exists(os.path.join(dest[0], "r3d_18_003.npy"))


def test_r2plus1d_single_video_statistics_script(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flickering_adversarial_video_amd import videoresnet_spec as vs
    from flickering_adversarial_video_amd.torch_attack import FlickerVideoResNet
    T = 8
    norm = vs.synthetic_clip(2, T, seed=6)
    eng = FlickerVideoResNet("r3d_18", vs.synthetic_weights("r3d_18", 42), batch_size=1, sample_length=T, dtype="f32")
    lab = [int(eng.logits(torch.from_numpy(norm[i:i + 1]).cuda(), False).argmax()) for i in range(2)]
    del eng
    lab[1] = (lab[1] + 1) % 400                                               # second clip "misclassified": no attack, None result
    np.savez(tmp_path / "v.npz", clips=norm, labels=np.array(lab), names=np.array(["clipA", "clipB"]))
    (tmp_path / "labels.txt").write_text("\n".join(f"class {i}" for i in range(400)))
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "r2plus1d_main_statistics_single_video_attack.py"), "--videos-npz", str(tmp_path / "v.npz"),
           "--label-map", str(tmp_path / "labels.txt"), "--results-root", str(tmp_path / "res"), "--base-model", "r3d_18", "--dtype", "f32",
           "--n-iter", "3", "--restart-after", "40"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "clipA:" in r.stdout and "clipB: clean clip misclassified" in r.stdout
    files = sorted(glob.glob(str(tmp_path / "res" / "r3d_18" / "single_video_attack" / "flickering" / "*" / "*.npy")))
    assert [os.path.basename(f) for f in files] == [f"clipA_@class_{lab[0]}.npy", f"clipB_@class_{lab[1]}.npy"]
    ra = np.load(files[0], allow_pickle=True).tolist()
    assert len(ra["loss/total"]) >= 3 and ra["perturbation"][0].shape == (3, T, 1, 1) and ra["prob_clean_input"].shape == (1, 400)
    assert np.load(files[1], allow_pickle=True).tolist() is None
    # the same videos two at a time (--batch 2: per-clip perturbations / clamp bounds / Adam states, model.py:791-982 batched): the
    # same files with bitwise-equal trajectories (fp32; fresh optimiser state per video in both runs)
    for tag, extra in (("one", []), ("two", ["--batch", "2"])):
        cmd2 = cmd[:cmd.index("--results-root") + 1] + [str(tmp_path / tag)] + cmd[cmd.index("--results-root") + 2:] + ["--reset-optimizer-per-video"] + extra
        r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=600)
        assert r2.returncode == 0, r2.stdout + r2.stderr
    fa = sorted(glob.glob(str(tmp_path / "one" / "r3d_18" / "single_video_attack" / "flickering" / "*" / "*.npy")))
    fb = sorted(glob.glob(str(tmp_path / "two" / "r3d_18" / "single_video_attack" / "flickering" / "*" / "*.npy")))
    assert [os.path.basename(f) for f in fa] == [os.path.basename(f) for f in fb] == [os.path.basename(f) for f in files]
    a, b = np.load(fa[0], allow_pickle=True).tolist(), np.load(fb[0], allow_pickle=True).tolist()
    assert a["loss/total"] == b["loss/total"] and a["is_adversarial"] == b["is_adversarial"]
    assert all(np.array_equal(p_, q_) for p_, q_ in zip(a["perturbation"], b["perturbation"]))
    assert np.load(fb[1], allow_pickle=True).tolist() is None

"""run_config.yml surface and result-file schema (host logic, no GPU)."""
import os

import pytest

from flickering_adversarial_video_amd import config as cfgmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_config_has_the_reference_surface():
    cfg = cfgmod.load_config(os.path.join(ROOT, "run_config.yml"))
    assert set(cfg) >= {"DATA", "MODEL", "SINGLE_VIDEO_ATTACK", "CLASS_GEN_ATTACK", "UNIVERSAL_ATTACK"}
    common = {"TARGETED_ATTACK", "TARGETED_CLASS", "IMPROVE_ADV_LOSS", "PROB_MARGIN", "USE_LOGITS", "MAX_NUM_STEP", "LAMBDA", "BETA_1",
              "BETA_2", "BATCH_SIZE", "CYCLIC_ATTACK", "PKL_RESULT_PATH", "NPY_PATH", "TF_RECORDS_TRAIN_PATH", "TF_RECORDS_VAL_PATH"}
    for sec in ("SINGLE_VIDEO_ATTACK", "CLASS_GEN_ATTACK", "UNIVERSAL_ATTACK"):
        assert common <= set(cfg[sec]), (sec, common - set(cfg[sec]))
    assert {"NUM_OF_TRAIN_TF_RECORDS", "NUM_OF_VAL_TF_RECORDS", "NUM_OF_VID_EACH_TF_RECORDS"} <= set(cfg.CLASS_GEN_ATTACK)
    assert {"FLICKERING_ATTACK", "CYCLIC_PERTURBATION_ATTACK"} <= set(cfg.UNIVERSAL_ATTACK)
    # reference defaults (run_config.yml:18-24,41-46,70-75)
    assert cfg.SINGLE_VIDEO_ATTACK.MAX_NUM_STEP == 2500 and cfg.SINGLE_VIDEO_ATTACK.LAMBDA == 1.0 and cfg.CLASS_GEN_ATTACK.LAMBDA == 10.0
    assert cfg.UNIVERSAL_ATTACK.BATCH_SIZE == 8 and cfg.SINGLE_VIDEO_ATTACK.PROB_MARGIN == 0.05
    assert cfg.MODEL.FRAMES == 90 and cfg.MODEL.DTYPE in ("bf16", "f32")


def test_reference_style_config_without_new_keys(tmp_path):
    p = tmp_path / "c.yml"
    p.write_text("DATA:\n    LABEL_MAP_PATH: 'x'\nMODEL:\n    CKPT_PATH: 'y'\nSINGLE_VIDEO_ATTACK:\n    BETA_1: 0.1\n")
    cfg = cfgmod.load_config(str(p))
    assert cfg.MODEL.FRAMES == 90 and cfg.MODEL.CKPT_PATH == "y" and cfg.SINGLE_VIDEO_ATTACK.BETA_1 == 0.1


def test_label_and_result_names(tmp_path):
    classes = ["abseiling", "bartending", "triple jump"]
    assert cfgmod.label_from_npy_name("/a/rgb_12@triple_jump.npy", classes) == ("triple jump", 2)
    with pytest.raises(ValueError):
        cfgmod.label_from_npy_name("/a/clip.npy", classes)
    with pytest.raises(ValueError):
        cfgmod.label_from_npy_name("/a/rgb_1@unknown_class.npy", classes)
    # README.md:71 sample result name of the reference
    assert cfgmod.result_filename("bartending", 0.1, 1.67, 1.19) == "bartending_beta1_0.1_th_1.67%_rg_1.19%.pkl"
    assert len(cfgmod.RESULT_KEYS) == 20


def test_engine_surface_mirrors_the_reference_loss_methods():
    """kinetics_i3d.improve_adversarial_loss(margin, targeted, logits) / ce_adversarial_loss(targeted) /
    get_kinetics_classes / predict exist on the engine under the reference's names and argument meaning
    (kinetics_i3d_utils.py:214-215,253-307); the loss methods return the choice as keyword arguments of step()."""
    import inspect
    from flickering_adversarial_video_amd.i3d_engine import FlickerI3D
    assert FlickerI3D.improve_adversarial_loss(margin=0.1, targeted=True, logits=True) == dict(improve_loss=True, margin=0.1, targeted=True,
                                                                                            use_logits=True)
    assert FlickerI3D.improve_adversarial_loss() == dict(improve_loss=True, margin=0.05, targeted=False, use_logits=False)
    assert FlickerI3D.ce_adversarial_loss(targeted=True) == dict(improve_loss=False, targeted=True)
    step_params = inspect.signature(FlickerI3D.step).parameters
    for spec in (FlickerI3D.improve_adversarial_loss(), FlickerI3D.ce_adversarial_loss()):
        assert set(spec) <= set(step_params)
    for name in ("get_kinetics_classes", "predict", "evaluate", "__call__", "reset_perturbation"):
        assert callable(getattr(FlickerI3D, name))
    assert "kinetics_classes" in inspect.signature(FlickerI3D.__init__).parameters

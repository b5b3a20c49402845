"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol
include/flicker_hip.h declares (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from flickering_adversarial_video_amd import _lib, build
    build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def header_functions():
    src = open(os.path.join(ROOT, "include", "flicker_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(flk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from flickering_adversarial_video_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/flicker_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)


def test_version_and_error_string(lib):
    assert lib.flk_version() >= 100
    assert isinstance(lib.flk_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    """pure host-side validation paths return FLK_EINVAL and set a message (no device work)"""
    import ctypes as C
    from flickering_adversarial_video_amd import _lib
    h = C.c_void_p()
    assert lib.flk_net_create(99, 0, 1, 16, 224, 224, 0, C.byref(h)) == -1
    assert b"arch" in lib.flk_last_error()
    a = _lib.LossArgs()
    a.B, a.C, a.torch_dialect, a.improve_loss, a.targeted, a.margin = 1, 400, 1, 1, 1, 0.05
    assert lib.flk_softmax_adv_loss(C.byref(a), C.c_void_p(8), C.c_void_p(8), None, C.c_void_p(8), C.c_void_p(8), None) == -1
    assert b"non-functional" in lib.flk_last_error()
    assert lib.flk_perturb_grad_scratch_bytes(8, 64, 224, 224) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from flickering_adversarial_video_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FlickerHipError):
        _lib.load()


def test_product_library_has_no_timing_ablation_switches():
    """The wrong-result timing ablations (skip loads / atomics / MFMAs) exist in -DFLK_ABLATE / -DSF_ABLATE / -DFLK_STEM_NI8 builds only
    (tools/*_time.py): the shipped library must not even contain the names of their environment switches."""
    from flickering_adversarial_video_amd import build
    blob = open(build.build(verbose=False), "rb").read()
    for name in (b"FLK_PF_DBG", b"FLK_PG_DBG", b"FLK_SG_DBG", b"FLK_SF_ABLATE", b"FLK_SF_STAGGER", b"FLK_STEM_NI"):
        assert name not in blob, name

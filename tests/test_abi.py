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


def test_group_launch_validation_and_layout_query_without_gpu(lib):
    """flk_conv3d_group / flk_conv_layout_query: host-side checks only (no device work).  The layout query reproduces the heuristics the
    plan builder relies on when it packs the members of a grouped launch: Mixed_4b's Branch_1 forward at the benchmark batch
    (96 -> 208 over 8 x 16 x 14 x 14) takes direct-A weights on 128-row tiles (two waves along the channels of its 128-wide tile),
    Mixed_4f's (160 -> 320) the LDS weight ring on 256-row tiles."""
    import ctypes as C
    from flickering_adversarial_video_amd import _lib
    a = _lib.ConvArgs()
    a.B, a.Ti, a.Hi, a.Wi = 8, 16, 14, 14
    a.To, a.Ho, a.Wo, a.OT, a.OH, a.OW = 16, 14, 14, 16, 14, 14
    a.kt = a.kh = a.kw = 3
    a.st = a.sh = a.sw = a.ost = a.osh = a.osw = 1
    a.pt = a.ph = a.pw = 1
    wn, mode = C.c_int(), C.c_int()
    a.cin, a.cout, a.in_ld, a.out_ld = 96, 208, 96, 208
    assert lib.flk_conv_layout_query(C.byref(a), 8, _lib.FLK_BF16, -1, C.byref(wn), C.byref(mode)) == 0
    assert (wn.value, mode.value) == (2, 1)
    a.cin, a.cout, a.in_ld, a.out_ld = 160, 320, 160, 320
    assert lib.flk_conv_layout_query(C.byref(a), 4, _lib.FLK_BF16, -1, C.byref(wn), C.byref(mode)) == 0
    assert wn.value == 1 and mode.value in (0, 5)          # the ring (5: its form with the weights fetched a row of taps ahead)
    assert lib.flk_conv_layout_query(C.byref(a), 4, _lib.FLK_BF16, 1, C.byref(wn), C.byref(mode)) == 0 and mode.value == 1      # forced direct-A
    assert lib.flk_conv_layout_query(None, 4, _lib.FLK_BF16, -1, C.byref(wn), C.byref(mode)) == -1
    # group launch: argument validation
    ap = (C.POINTER(_lib.ConvArgs) * 1)(C.pointer(a))
    wp = (C.c_void_p * 1)(None)
    assert lib.flk_conv3d_group(ap, wp, 0, 4, 0, _lib.FLK_BF16, None) == -1 and b"members" in lib.flk_last_error()
    assert lib.flk_conv3d_group(ap, wp, 1, 3, 0, _lib.FLK_BF16, None) == -1 and b"fragments" in lib.flk_last_error()
    assert lib.flk_conv3d_group(ap, wp, 1, 4, 0, _lib.FLK_F32, None) == -1
    assert lib.flk_conv3d_group(ap, wp, 1, 4, 0, _lib.FLK_BF16, None) == -1 and b"null member" in lib.flk_last_error()


def test_producer_consumer_routing_model_without_gpu(lib):
    """flk_conv3d_pc_query: what flk_conv3d / flk_conv3d_group send to the persistent producer / consumer kernel (csrc/conv_pc.hip), decided on
    the host from the geometry.  Conv3d_2c_3x3 (i3d.py:183-186) at half the benchmark batch: forward 64 -> 192 on 8x7x8 boxes (448 rows) -- 11 rounds of
    seven eighths the length beat 10 rounds of 8x8x8 -- and its data-gradient 192 -> 64 likewise (3.5 -> 4 rounds instead of 3.06 -> 4); the
    same layer at batch 1 (160 K steps) is routed too since the halo is staged by LDS-DMA, a single round of short items (Mixed_4*) is not;
    (1,3,3) taps are taken, strided and other tap shapes refused with the reason."""
    import ctypes as C
    from flickering_adversarial_video_amd import _lib

    def conv(B, T, H, W, cin, cout, k=3, stride=1):
        a = _lib.ConvArgs()
        a.B, a.Ti, a.Hi, a.Wi = B, T, H, W
        a.To, a.Ho, a.Wo, a.OT, a.OH, a.OW = T // stride, H // stride, W // stride, T // stride, H // stride, W // stride
        a.kt = a.kh = a.kw = k
        a.st = a.sh = a.sw = stride
        a.ost = a.osh = a.osw = 1
        a.pt = a.ph = a.pw = (k - 1) // 2
        a.cin, a.cout, a.in_ld, a.out_ld = cin, cout, cin, cout
        return a

    def query(*members):
        ap = (C.POINTER(_lib.ConvArgs) * len(members))(*[C.pointer(m) for m in members])
        tile, ni, eff, steps = (C.c_int * 3)(), C.c_int(), C.c_double(), C.c_double()
        rc = lib.flk_conv3d_pc_query(ap, len(members), _lib.FLK_BF16, tile, C.byref(ni), C.byref(eff), C.byref(steps))
        return rc, tuple(tile), ni.value, eff.value, steps.value

    rc, tile, ni, eff, steps = query(conv(4, 32, 56, 56, 64, 192))
    assert rc == 1 and tile == (8, 7, 8) and ni == 7 and eff > 0.8 and 500 < steps < 700, (rc, tile, ni, eff, steps)
    rc, tile, ni, eff, steps = query(conv(4, 32, 56, 56, 192, 64))
    assert rc == 1 and tile == (8, 7, 8) and ni == 7, (rc, tile, ni, eff, steps)
    rc, tile, ni, eff, steps = query(conv(1, 32, 56, 56, 64, 192))
    assert rc == 1 and 150 <= steps < 170, (rc, tile, ni, eff, steps)               # batch 1: 160 steps -- routed since the LDS-DMA halo staging (bound 150; 300 before)
    rc, tile, ni, eff, steps = query(conv(8, 32, 28, 28, 128, 192), conv(8, 32, 28, 28, 32, 96))      # Mixed_3c Branch_1 + Branch_2 (i3d.py:229-238)
    assert rc == 1 and tile[0] * tile[1] * tile[2] == 448 and 28 % tile[1] == 0 and 28 % tile[2] == 0 and ni == 7 and eff > 0.8, (rc, tile, ni, eff, steps)
    rc = query(conv(8, 16, 14, 14, 96, 208))[0]
    assert rc == 0                                                                   # Mixed_4*: a single round of items (77 steps)
    a133 = conv(8, 16, 56, 56, 64, 144)                                              # the spatial half of r2plus1d_18's layer1 units (Conv2Plus1D, model.py:421): (1,3,3) taps
    a133.kt, a133.pt = 1, 0
    rc, tile, ni, eff, steps = query(a133)
    assert rc == 1 and tile[0] * tile[1] * tile[2] == 448 and ni == 7 and eff > 0.65 and steps > 150, (rc, tile, ni, eff, steps)
    assert query(conv(4, 32, 56, 56, 64, 192, k=1))[0] == -1 and b"3x3x3" in lib.flk_last_error()
    assert query(conv(4, 32, 56, 56, 64, 192, stride=2))[0] == -1 and b"stride" in lib.flk_last_error()

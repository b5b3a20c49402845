"""ctypes binding of libflicker_hip.so (include/flicker_hip.h).  Fails loudly if the library is missing:
there is no CPU / eager fallback in the product path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLK_LIB_PATH") or os.path.join(HERE, "libflicker_hip.so")     # FLK_LIB_PATH: A/B a second build of the library

FLK_F32, FLK_BF16 = 0, 1
FLK_NET_I3D, FLK_NET_R2PLUS1D_18, FLK_NET_R3D_18, FLK_NET_MC3_18 = 0, 1, 2, 3


class FlickerHipError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_ld", C.c_int), ("in_coff", C.c_int), ("cin", C.c_int),
                ("B", C.c_int), ("Ti", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int),
                ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("st", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("pt", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
                ("To", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("out", C.c_void_p), ("out_ld", C.c_int), ("out_coff", C.c_int), ("cout", C.c_int),
                ("OT", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
                ("ost", C.c_int), ("osh", C.c_int), ("osw", C.c_int),
                ("oot", C.c_int), ("ooh", C.c_int), ("oow", C.c_int),
                ("scale", C.c_void_p), ("bias", C.c_void_p),
                ("add", C.c_void_p), ("add_ld", C.c_int), ("add_coff", C.c_int),
                ("mask", C.c_void_p), ("mask_ld", C.c_int), ("mask_coff", C.c_int),
                ("relu", C.c_int),
                ("in2", C.c_void_p), ("in2_ld", C.c_int), ("in2_coff", C.c_int), ("cin1", C.c_int),
                ("out2", C.c_void_p), ("out2_ld", C.c_int), ("out2_coff", C.c_int), ("cout1", C.c_int),
                ("pos_bias", C.c_void_p), ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64), ("pos_bias_bstride", C.c_int64)]


class PoolArgs(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_ld", C.c_int), ("in_coff", C.c_int), ("C", C.c_int),
                ("B", C.c_int), ("Ti", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int),
                ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("st", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("pt", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
                ("To", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("out", C.c_void_p), ("out_ld", C.c_int), ("out_coff", C.c_int),
                ("idx", C.c_void_p), ("relu_input", C.c_int)]


class ApplyArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_is_u8", C.c_int), ("x_scale", C.c_float), ("x_bias", C.c_float),
                ("delta", C.c_void_p), ("delta_dense", C.c_int),
                ("dclip", C.c_float), ("inv_std", C.c_float * 3),
                ("lo", C.c_float), ("hi", C.c_float), ("adv_flag", C.c_float),
                ("shift_x", C.c_int), ("shift_p", C.c_int),
                ("B", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int), ("fold_t", C.c_int), ("center", C.c_int),
                ("delta_per_clip", C.c_int), ("dclip_dev", C.c_void_p)]


class AdamArgs(C.Structure):
    _fields_ = [("T", C.c_int), ("torch_dialect", C.c_int),
                ("beta0", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("beta3", C.c_float),
                ("dyn_max_norm", C.c_float), ("g_scale", C.c_float),
                ("lr", C.c_float), ("adam_b1", C.c_float), ("adam_b2", C.c_float), ("adam_eps", C.c_float),
                ("step", C.c_int)]


class DenseAdamArgs(C.Structure):
    _fields_ = [("T", C.c_int), ("H", C.c_int), ("W", C.c_int), ("torch_dialect", C.c_int), ("beta", C.c_float), ("g_scale", C.c_float),
                ("lr", C.c_float), ("adam_b1", C.c_float), ("adam_b2", C.c_float), ("adam_eps", C.c_float), ("step", C.c_int),
                ("dyn_max_norm", C.c_float)]


class LossArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("torch_dialect", C.c_int), ("improve_loss", C.c_int),
                ("use_logits", C.c_int), ("targeted", C.c_int), ("margin", C.c_float), ("mean_scale", C.c_float)]


_SIGS = {
    "flk_version": (C.c_int, []),
    "flk_last_error": (C.c_char_p, []),
    "flk_conv_weights_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "flk_conv_weights_create_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "flk_conv_weights_create_s2d_stem": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "flk_conv_weights_destroy": (C.c_int, [C.c_void_p]),
    "flk_conv3d": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p, C.c_int, C.c_void_p]),
    "flk_maxpool3d_fwd": (C.c_int, [C.POINTER(PoolArgs), C.c_int, C.c_void_p]),
    "flk_maxpool3d_bwd": (C.c_int, [C.POINTER(PoolArgs), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "flk_pool_gemm_weights_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "flk_pool_gemm_weights_destroy": (C.c_int, [C.c_void_p]),
    "flk_maxpool3d_bwd_gemm": (C.c_int, [C.POINTER(PoolArgs), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]),
    "flk_perturb_apply_s2d": (C.c_int, [C.POINTER(ApplyArgs), C.c_void_p, C.c_int, C.c_void_p]),
    "flk_perturb_grad_scratch_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "flk_perturb_grad_reduce": (C.c_int, [C.POINTER(ApplyArgs), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_pack_batch_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "flk_perturb_reg_adam": (C.c_int, [C.POINTER(AdamArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_perturb_reg_adam_batched": (C.c_int, [C.POINTER(AdamArgs), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_dense_adam_scratch_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "flk_perturb_dense_l12_adam": (C.c_int, [C.POINTER(DenseAdamArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "flk_softmax_adv_loss": (C.c_int, [C.POINTER(LossArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_net_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "flk_net_destroy": (C.c_int, [C.c_void_p]),
    "flk_net_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "flk_net_finalize": (C.c_int, [C.c_void_p]),
    "flk_net_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "flk_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "flk_net_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_conv_splitk_bytes": (C.c_int64, [C.POINTER(ConvArgs), C.c_void_p]),
    "flk_conv3d_group": (C.c_int, [C.POINTER(C.POINTER(ConvArgs)), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "flk_conv3d_pc": (C.c_int, [C.POINTER(C.POINTER(ConvArgs)), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "flk_conv3d_pc_worthwhile": (C.c_int, [C.POINTER(C.POINTER(ConvArgs)), C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "flk_conv3d_pc_query": (C.c_int, [C.POINTER(C.POINTER(ConvArgs)), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "flk_conv3d_pc_why_not": (C.c_char_p, [C.POINTER(ConvArgs), C.c_void_p, C.c_int]),
    "flk_conv3d_group_check": (C.c_int, [C.POINTER(C.POINTER(ConvArgs)), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "flk_conv_layout_query": (C.c_int, [C.POINTER(ConvArgs), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "flk_comm_unique_id": (C.c_int, [C.c_void_p]),
    "flk_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "flk_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "flk_comm_destroy": (C.c_int, [C.c_void_p]),
    "flk_net_has_forward_flicker": (C.c_int, [C.c_void_p]),
    "flk_net_forward_flicker": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p]),
    "flk_net_forward_apply": (C.c_int, [C.c_void_p, C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_stem_delta_bias_weights_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "flk_stem_delta_bias": (C.c_int, [C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_stem_fwd_u8_weights_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "flk_stem_fwd_u8": (C.c_int, [C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "flk_net_has_backward_delta": (C.c_int, [C.c_void_p]),
    "flk_net_backward_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_net_prepare_backward_delta": (C.c_int, [C.c_void_p, C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p]),
    "flk_stem_delta_grad_scratch_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "flk_stem_delta_grad_weights_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "flk_stem_delta_grad_weights_destroy": (C.c_int, [C.c_void_p]),
    "flk_stem_delta_grad_mask": (C.c_int, [C.POINTER(ApplyArgs), C.c_void_p, C.c_void_p]),
    "flk_stem_delta_grad": (C.c_int, [C.POINTER(ApplyArgs), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "flk_net_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "flk_net_autotune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flk_conv_set_autotune": (C.c_int, [C.c_int]),
    "flk_net_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "flk_net_input_numel": (C.c_int64, [C.c_void_p]),
    "flk_net_input_channels": (C.c_int, [C.c_void_p]),
    "flk_net_input_fold": (C.c_int, [C.c_void_p]),
    "flk_net_num_classes": (C.c_int, [C.c_void_p]),
    "flk_net_get_activation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """dlopen the HIP library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlickerHipError(
            f"{LIB_PATH} is missing: build it with `python -m flickering_adversarial_video_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; load it FIRST so that this library's NEEDED libamdhip64.so.7
    # resolves to the SAME runtime instance (one HIP runtime per process: shared streams / device pointers).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise FlickerHipError(f"libflicker_hip error {rc}: {load().flk_last_error().decode(errors='replace')}")


def ptr(t):
    """raw device/host pointer of a torch tensor / numpy array / None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dtype):
    import torch
    if dtype in (torch.float32, "fp32", "f32", FLK_F32):
        return FLK_F32
    if dtype in (torch.bfloat16, "bf16", FLK_BF16):
        return FLK_BF16
    raise ValueError(f"unsupported dtype {dtype!r}")


def torch_dtype(code):
    import torch
    return torch.float32 if code == FLK_F32 else torch.bfloat16

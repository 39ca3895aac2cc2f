"""ctypes binding of the C-ABI in include/segengine.h.

The product library is the in-tree gfx950 build (pytorchdeeplearing_amd/lib/libsegengine.so).
There is NO CPU implementation: tensors that are not on an AMD GPU raise (`lib_for`), a missing library raises
(`product_library`).  Everything in the package reaches the library through `lib_for(device)` / `host_library()`."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SEGENGINE_LIB: tuning builds only (tools/build_variant.py); the product library is lib/libsegengine.so
LIB_PATH = os.environ.get("SEGENGINE_LIB") or os.path.join(HERE, "lib", "libsegengine.so")

NET_KIND = {"vnet": 0, "unet": 1}
DTYPE = {"f32": 0, "fp32": 0, "float32": 0, "f16": 1, "fp16": 1, "float16": 1, "bf16": 2, "bfloat16": 2}
LABEL_TYPES = {"torch.uint8": 0, "torch.int32": 1, "torch.int64": 2, "torch.float32": 3}
LOSS_KIND = {
    "BinaryDiceLoss": 0, "BinaryCrossEntropyLoss": 1, "BinaryFocalLoss": 2, "BinaryCrossEntropyDiceLoss": 3,
    "MutilCrossEntropyLoss": 4, "MutilFocalLoss": 5, "MutilDiceLoss": 6,
    "BinaryJaccardLoss": 7, "BinaryELDiceLoss": 8, "BinaryTverskyLoss": 9, "MutilCrossEntropyDiceLoss": 10, "MutilELDiceLoss": 11, "BinarySSLoss": 12,
    "MutilTverskyLoss": 13, "MutilSSLoss": 14, "MCC_Loss": 15,
}
LABEL_BINARIZE = 16          # flag: kernels read labels as (value != 0)  (SEG_LABEL_BINARIZE)
MASKS_EVAL, MASKS_GIVEN, MASKS_RANDOM = 0, 1, 2


def label_type(t, binarize=False):
    """C-ABI label type of a target tensor (+ the device-side binarise flag)"""
    return LABEL_TYPES[str(t.dtype)] | (LABEL_BINARIZE if binarize else 0)
KERNEL_CLASSES = ["conv3", "wgrad3", "conv_generic", "wgrad_generic", "stem", "gn_act", "gn_bwd_reduce", "gn_bwd_apply", "head", "conv3_smallbox", "gn_group", "misc"]

_vp, _i, _ll, _f, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double

SIGNATURES = {
    "seg_create": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "seg_destroy": (None, [_vp]),
    "seg_param_count": (_i, [_vp]),
    "seg_param_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_ll)]),
    "seg_param_numel": (_ll, [_vp]),
    "seg_dropout_calls": (_i, [_vp]),
    "seg_dropout_ld": (_i, [_vp]),
    "seg_dropout_channels": (_i, [_vp, _i]),
    "seg_dropout_draws": (_ll, [_vp]),
    "seg_set_dropout_draws": (_i, [_vp, _ll]),
    "seg_plan": (_i, [_vp, _i, _i, _i, _i]),
    "seg_workspace_bytes": (_ll, [_vp]),
    "seg_plan_count": (_i, [_vp, _i]),
    "seg_bind": (_i, [_vp, _vp, _vp, _vp]),
    "seg_pack_weights": (_i, [_vp, _vp]),
    "seg_forward": (_i, [_vp, _vp, _i, _vp, C.c_ulonglong, _vp, _vp, _vp]),
    "seg_backward": (_i, [_vp, _vp, _i, _vp]),
    "seg_backward_ops": (_i, [_vp]),
    "seg_backward_bucket": (_i, [_vp, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "seg_backward_range": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "seg_backward_slice": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "seg_side_wait": (_i, [_vp, _vp]),
    "seg_set_loss_scale": (_i, [_vp, _f]),
    "seg_get_loss_scale": (_f, [_vp]),
    "seg_loss_ws_bytes": (_ll, [_i, _i]),
    "seg_loss_forward": (_i, [_vp, _vp, _i, _i, _i, _ll, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "seg_loss_shared_doubles": (_i, []),
    "seg_loss_reduce": (_i, [_vp, _vp, _i, _i, _i, _ll, _i, _f, _f, _vp, _vp]),
    "seg_loss_finalize": (_i, [_vp, _vp, _i, _i, _i, _ll, _i, _f, _f, _vp, _i, _vp, _vp, _vp]),
    "seg_loss_backward": (_i, [_vp, _vp, _i, _i, _i, _ll, _i, _f, _f, _vp, _f, _vp, _vp]),
    "seg_lovasz_ws_bytes": (_ll, [_i, _ll]),
    "seg_lovasz_forward": (_i, [_vp, _vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp]),
    "seg_ssim_ws_bytes": (_ll, [_i, _i, _ll]),
    "seg_ssim_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "seg_ssim_forward_cols": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "seg_ssim_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "seg_predict_mask": (_i, [_vp, _vp, _i, _i, _ll, C.c_float, _i, _vp]),
    "seg_metric": (_i, [_vp, _vp, _i, _i, _i, _ll, _vp, _vp, _vp]),
    "seg_train_step": (_i, [_vp, _vp, _vp]),
    "seg_train_graph_capture": (_i, [_vp, _vp, _vp]),
    "seg_train_graph_launch": (_i, [_vp, _vp]),
    "seg_train_graph_ready": (_i, [_vp]),
    "seg_adam_step": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _f, _i, _vp, _vp]),
    "seg_op_conv": (_i, [_vp, _i, _vp]),
    "seg_op_conv_kernel": (_i, [_vp]),
    "seg_op_wgrad": (_i, [_vp, _vp, _i, _vp]),
    "seg_op_wgrad_partial_bytes": (_ll, [_vp]),
    "seg_op_pack": (_i, [_vp, _i, _ll, _i, _vp]),
    "seg_op_conv3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "seg_op_conv3x": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "seg_op_conv3x_num_cfgs": (_i, []),
    "seg_op_conv3x_cfg_info": (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "seg_op_conv3x_default_cfg": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "seg_op_conv3x_cfg_frag": (_i, [_i]),
    "seg_op_wgrad3_partial_bytes": (_ll, [_i, _i, _i, _i, _i, _i, _i]),
    "seg_op_wgrad3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "seg_op_wgrad3_cat": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "seg_op_stemx_partial_bytes": (_ll, [_i, _i, _i, _i, _i, _i]),
    "seg_op_stemx": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "seg_abi_sizeof": (_i, [_i]),
    "seg_op_pool3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "seg_op_skel_iter": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "seg_op_skel_iter_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "seg_op_skel_update": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "seg_op_skel_update_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "seg_op_pool3_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "seg_op_plane_dot_scratch_bytes": (_ll, [_i, _ll]),
    "seg_op_plane_dot": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _vp]),
    "seg_op_plane_axpb": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp]),
    "seg_cldice_ws_bytes": (_ll, [_i, _i, _i, _i, _i, _i]),
    "seg_cldice_target": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "seg_cldice_binary": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "seg_op_resample3d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _d, _d, _d, _i, _vp]),
    "seg_op_normalize_ws_bytes": (_ll, []),
    "seg_op_normalize_meanstd": (_i, [_vp, _vp, _ll, _i, _f, _f, _vp, _vp]),
    "seg_op_normalize_percentile": (_i, [_vp, _vp, _ll, _f, _f, _vp, _vp]),
    "seg_op_gather_patches": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "seg_op_stitch_mask": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "seg_set_rccl_comm": (_i, [_vp, _vp, _vp]),
    "seg_profile_enable": (_i, [_vp, C.c_uint]),
    "seg_profile_read": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "seg_last_error": (C.c_char_p, []),
    "seg_build_info": (C.c_char_p, []),
}


class TrainArgs(C.Structure):
    """seg_train_args of include/segengine.h"""
    _fields_ = [("x", _vp), ("target", _vp), ("label_type", _i),
                ("loss_kind", _i), ("focal_alpha", _f), ("focal_gamma", _f), ("class_alpha", _vp),
                ("logits", _vp), ("probs", _vp), ("dlogits", _vp), ("loss_ws", _vp), ("out3", _vp),
                ("mask_mode", _i), ("masks", _vp), ("seed", C.c_ulonglong),
                ("exp_avg", _vp), ("exp_avg_sq", _vp), ("opt_state", _vp),
                ("lr", _f), ("beta1", _f), ("beta2", _f), ("eps", _f), ("weight_decay", _f), ("decoupled", _i), ("grad_div", _f),
                ("check_finite", _i), ("packed", _i),
                ("bucket_cb", _vp), ("loss_cb", _vp), ("cb_user", _vp), ("nfrac", _i), ("fractions", C.c_double * 4), ("aux_stream", _vp)]


BUCKET_CB = C.CFUNCTYPE(_i, _vp, _i, _ll, _ll)
NCCL_ALLREDUCE = C.CFUNCTYPE(_i, _vp, _vp, C.c_size_t, _i, _i, _vp, _vp)      # ncclAllReduce(send, recv, count, dtype, op, comm, stream)
LOSS_CB = C.CFUNCTYPE(_ll, _vp, _vp, _i)


class SegLib:
    def __init__(self, path):
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)     # raises AttributeError when the symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, rc, what=""):
        if rc is not None and rc < 0:
            raise RuntimeError("segengine %s failed: %s" % (what, self.seg_last_error().decode()))

    def build_info(self):
        return self.seg_build_info().decode()


_product = None


def product_library():
    global _product
    if _product is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsegengine.so (gfx950) is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m pytorchdeeplearing_amd.build`. There is no CPU fallback.")
        _product = SegLib(LIB_PATH)
    return _product


def lib_for(device):
    import torch
    device = torch.device(device)
    if device.type == "cuda":
        return product_library()
    raise RuntimeError("the segmentation engine runs on AMD MI355X (gfx950) only; got a %s tensor" % device.type)


def host_library():
    """the library for entry points that do no device work (network tables: seg_create / seg_param_info)"""
    return product_library()


def stream_for(device):
    import torch
    device = torch.device(device)
    if device.type == "cuda":
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return C.c_void_p(0)

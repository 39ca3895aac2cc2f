"""`model.losses` boundary: the seven losses reachable from the reference's model
wrappers (model/losses.py:33-53,129-197,247-325) as nn.Modules with the same names, constructor
arguments and call signature `loss(y_pred_logits, y_true) -> 0-dim tensor with grad`.  Forward and
backward both run in libsegengine (one reduction pass + one elementwise pass)."""
import ctypes as C

import torch
from torch import nn

from . import _capi
from .engine import aligned_empty

_LABEL_OK = (torch.uint8, torch.int32, torch.int64, torch.float32)


# loss kinds whose reference classes apply a sigmoid and sum over EVERY element of a same-shaped target (model/losses.py:19-30, 43-53, 66-74,
# 87-99, 113-126, 141-147, 160-181, 192-197): a multi-label [N, C > 1, ...] prediction is C*N independent planes to them
_BINARY_KINDS = (0, 1, 2, 3, 7, 8, 9, 12)


def _prep(logits, target, kind=None):
    lg = logits.float().contiguous()
    t = target
    if t.dtype not in _LABEL_OK:
        t = t.to(torch.int64)
    t = t.contiguous()
    n, c = lg.shape[0], lg.shape[1]
    v = lg.numel() // (n * c)
    if kind in _BINARY_KINDS and c > 1:
        # multi-label head: the reference views both tensors as (bs, num_classes, -1) and reduces with plain .sum() / .mean(), i.e. over all
        # N*C planes alike - the same numbers as a one-channel batch of N*C planes
        assert t.numel() == n * c * v, "a multi-label binary loss needs a target of the prediction's shape"
        n, c = n * c, 1
    assert t.numel() == n * v, "target must have one label per voxel"
    return lg, t, n, c, v


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, kind, falpha, fgamma, class_alpha):
        lg, t, n, c, v = _prep(logits, target, kind)
        lib = _capi.lib_for(lg.device)
        ws = aligned_empty(lib.seg_loss_ws_bytes(n, c), lg.device)
        out3 = torch.zeros(4, dtype=torch.float32, device=lg.device)
        ca = None if class_alpha is None else class_alpha.to(device=lg.device, dtype=torch.float32).contiguous()
        lib.check(lib.seg_loss_forward(lg.data_ptr(), t.data_ptr(), _capi.LABEL_TYPES[str(t.dtype)], n, c, v, kind, falpha, fgamma,
                                       ca.data_ptr() if ca is not None else None, ws.data_ptr(), out3.data_ptr(),
                                       _capi.stream_for(lg.device)), "seg_loss_forward")
        ctx.stuff = (lg, t, n, c, v, kind, falpha, fgamma, ws, lib)
        ctx.in_dtype, ctx.in_shape = logits.dtype, logits.shape
        return out3[0].clone()

    @staticmethod
    def backward(ctx, g):
        lg, t, n, c, v, kind, falpha, fgamma, ws, lib = ctx.stuff
        dl = torch.empty_like(lg)
        lib.check(lib.seg_loss_backward(lg.data_ptr(), t.data_ptr(), _capi.LABEL_TYPES[str(t.dtype)], n, c, v, kind, falpha, fgamma,
                                        ws.data_ptr(), 1.0, dl.data_ptr(), _capi.stream_for(lg.device)), "seg_loss_backward")
        return (dl * g).to(ctx.in_dtype).reshape(ctx.in_shape), None, None, None, None, None


class _Loss(nn.Module):
    kind = None

    def __init__(self):
        super().__init__()
        self.focal_alpha, self.focal_gamma, self.class_alpha = 0.25, 2.0, None

    def forward(self, y_pred_logits, y_true):
        return _LossFn.apply(y_pred_logits, y_true, _capi.LOSS_KIND[self.kind], float(self.focal_alpha), float(self.focal_gamma),
                             self.class_alpha)


class BinaryDiceLoss(_Loss):
    """model/losses.py:33-53"""
    kind = "BinaryDiceLoss"


class BinaryCrossEntropyLoss(_Loss):
    """model/losses.py:129-147"""
    kind = "BinaryCrossEntropyLoss"


class BinaryFocalLoss(_Loss):
    """model/losses.py:150-181"""
    kind = "BinaryFocalLoss"

    def __init__(self, alpha=0.25, gamma=2):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma
        self.focal_alpha, self.focal_gamma = alpha, gamma


class BinaryCrossEntropyDiceLoss(_Loss):
    """model/losses.py:184-197"""
    kind = "BinaryCrossEntropyDiceLoss"


class MutilCrossEntropyLoss(_Loss):
    """model/losses.py:247-260 (alpha is accepted and, like in the reference, unused)"""
    kind = "MutilCrossEntropyLoss"

    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha


class MutilFocalLoss(_Loss):
    """model/losses.py:263-285"""
    kind = "MutilFocalLoss"

    def __init__(self, alpha, gamma=2, torch=True):
        super().__init__()
        self.alpha, self.gamma, self.torch = alpha, gamma, torch
        self.focal_gamma = gamma


class MutilDiceLoss(_Loss):
    """model/losses.py:288-325"""
    kind = "MutilDiceLoss"

    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha
        self.class_alpha = alpha if torch.is_tensor(alpha) else torch.as_tensor(alpha, dtype=torch.float32)


# ---- losses of model/losses.py that no wrapper's `loss_name` selects: ratios of the same sums ----
class BinaryJaccardLoss(_Loss):
    """model/losses.py:9-30"""
    kind = "BinaryJaccardLoss"


class BinaryELDiceLoss(_Loss):
    """model/losses.py:56-74"""
    kind = "BinaryELDiceLoss"


class BinaryTverskyLoss(_Loss):
    """model/losses.py:102-126 (alpha = 0.3 on false positives, beta = 0.7 on false negatives)"""
    kind = "BinaryTverskyLoss"


class BinarySSLoss(_Loss):
    """model/losses.py:77-99 (sensitivity-specificity, r = 0.1)"""
    kind = "BinarySSLoss"


class MutilCrossEntropyDiceLoss(MutilDiceLoss):
    """model/losses.py:328-342: MutilDiceLoss(alpha) + MutilCrossEntropyLoss(alpha)"""
    kind = "MutilCrossEntropyDiceLoss"


class MutilELDiceLoss(MutilDiceLoss):
    """model/losses.py:345-382"""
    kind = "MutilELDiceLoss"


class MutilTverskyLoss(_Loss):
    """model/losses.py:421-459.  As in the reference, `alpha` (tensor [C]) is both the class weight and the false-positive weight, and
    the class does not define `beta`: set `loss.beta` before the first call (the reference raises AttributeError otherwise, and so does
    this one).  Forward/backward: the multi-class reduction sums + seg_loss kind 13."""
    kind = "MutilTverskyLoss"

    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha
        self.class_alpha = alpha if torch.is_tensor(alpha) else torch.as_tensor(alpha, dtype=torch.float32)

    def forward(self, y_pred_logits, y_true):
        return _LossFn.apply(y_pred_logits, y_true, _capi.LOSS_KIND[self.kind], 0.0, float(self.beta), self.class_alpha)


class MutilSSLoss(_Loss):
    """model/losses.py:385-418; `r` is not defined by the reference class: set `loss.r` before the first call."""
    kind = "MutilSSLoss"

    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha
        self.class_alpha = alpha if torch.is_tensor(alpha) else torch.as_tensor(alpha, dtype=torch.float32)
        self.smooth = 1.e-5

    def forward(self, y_pred_logits, y_true):
        return _LossFn.apply(y_pred_logits, y_true, _capi.LOSS_KIND[self.kind], 0.0, float(self.r), self.class_alpha)


class MCC_Loss(_Loss):
    """model/losses.py:200-232: Matthews-correlation loss of a PROBABILITY map `inputs` against `targets` of the same shape (the one
    loss of the file that does not take logits); the gradient is with respect to `inputs`."""
    kind = "MCC_Loss"

    def forward(self, inputs, targets):
        assert inputs.numel() == targets.numel(), "inputs and targets must have the same number of elements"
        v = _LossFn.apply(inputs.reshape(1, 1, -1), targets.reshape(1, -1), _capi.LOSS_KIND[self.kind], 0.0, 0.0, None)
        return v


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, n, c, v):
        xs = x.float().contiguous()
        t = target if target.dtype in _LABEL_OK else target.to(torch.int64)
        t = t.contiguous()
        lib = _capi.lib_for(xs.device)
        nbytes = lib.seg_lovasz_ws_bytes(n, v)
        lib.check(nbytes, "seg_lovasz_ws_bytes")
        ws = aligned_empty(nbytes, xs.device)
        out1 = torch.zeros(4, dtype=torch.float32, device=xs.device)
        dx = torch.empty_like(xs)
        lib.check(lib.seg_lovasz_forward(xs.data_ptr(), t.data_ptr(), _capi.LABEL_TYPES[str(t.dtype)], n, c, v, ws.data_ptr(), out1.data_ptr(),
                                         dx.data_ptr(), _capi.stream_for(xs.device)), "seg_lovasz_forward")
        ctx.dx, ctx.in_dtype, ctx.in_shape = dx, x.dtype, x.shape
        return out1[0].clone()

    @staticmethod
    def backward(ctx, g):
        return (ctx.dx * g).to(ctx.in_dtype).reshape(ctx.in_shape), None, None, None, None


class BinaryLovaszLoss(nn.Module):
    """model/losses.py:235-242 -> model/lovasz.py:34-71 Lovasz hinge over the whole batch (per_image=False, ignore_index=None; the other
    settings are not built).  logits [B, ...] and target of the same shape.  (The reference class cannot be CALLED as shipped - its
    __init__ skips nn.Module.__init__ - only .forward() runs; this one supports both.)"""

    def __init__(self, per_image=False, ignore_index=None):
        super().__init__()
        if per_image or ignore_index is not None:
            raise NotImplementedError("BinaryLovaszLoss: only per_image=False, ignore_index=None (the reference defaults) are built")
        self.per_image, self.ignore_index = per_image, ignore_index

    def forward(self, logits, target):
        assert logits.numel() == target.numel(), "logits and target must have the same number of elements"
        return _LovaszFn.apply(logits.reshape(1, 1, -1), target.reshape(1, -1), 1, 1, logits.numel())


class LovaszLoss(nn.Module):
    """model/losses.py:462-473 -> model/lovasz.py:90-141 (classes='present', per_image=False, ignore=None).  As in the reference the first
    argument is used as the class scores AS GIVEN (no soft-max is applied on the way)."""

    def __init__(self, per_image=False, ignore=None):
        super().__init__()
        if per_image or ignore is not None:
            raise NotImplementedError("LovaszLoss: only per_image=False, ignore=None (the reference defaults) are built")
        self.per_image, self.ignore = per_image, ignore

    def forward(self, logits, target):
        n, c = logits.shape[0], logits.shape[1]
        v = logits.numel() // (n * c)
        assert target.numel() == n * v, "target must have one label per voxel"
        return _LovaszFn.apply(logits.reshape(n, c, v), target.reshape(n, v), n, c, v)

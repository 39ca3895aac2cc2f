"""`model.metric` boundary: dice_coeff / iou_coeff / multiclass_dice_coeff /
multiclass_iou_coeff of model/metric.py:146-215 (threshold 0.5, per-sample, mean over batch; the
multi-class variants skip the background class), computed by one reduction kernel."""
import torch

from . import _capi
from .engine import aligned_empty
from .losses import _LABEL_OK


def _metric(probs, target, c):
    p = probs.float().contiguous()
    t = target if target.dtype in _LABEL_OK else target.to(torch.int64)
    t = t.contiguous()
    n = p.shape[0]
    v = p.numel() // (n * c)
    assert t.numel() == n * v
    lib = _capi.lib_for(p.device)
    ws = aligned_empty(8 * 3 * n * c + 256, p.device)
    out2 = torch.zeros(2, dtype=torch.float32, device=p.device)
    lib.check(lib.seg_metric(p.data_ptr(), t.data_ptr(), _capi.LABEL_TYPES[str(t.dtype)], n, c, v, ws.data_ptr(), out2.data_ptr(),
                             _capi.stream_for(p.device)), "seg_metric")
    return out2


def dice_coeff(input, target):
    return _metric(input, target, 1)[0]


def iou_coeff(input, target):
    return _metric(input, target, 1)[1]


def multiclass_dice_coeff(input, target):
    return _metric(input, target, input.shape[1])[0]


def multiclass_iou_coeff(input, target):
    return _metric(input, target, input.shape[1])[1]


def predict_mask(probs, threshold=0.5, scale=255):
    """predict() post-processing on the device (modelVNet.py:670-676, modelUnet.py:672-680): probs (N, C, *spatial) fp32 ->
    uint8 mask (N, *spatial); C == 1: (p > threshold) * scale, C > 1: np.argmax over the class axis (first maximum)."""
    p = probs.float().contiguous()
    n, c = p.shape[0], p.shape[1]
    v = p.numel() // (n * c)
    lib = _capi.lib_for(p.device)
    out = torch.empty((n,) + tuple(p.shape[2:]), dtype=torch.uint8, device=p.device)
    lib.check(lib.seg_predict_mask(p.data_ptr(), out.data_ptr(), n, c, v, float(threshold), int(scale), _capi.stream_for(p.device)),
              "seg_predict_mask")
    return out

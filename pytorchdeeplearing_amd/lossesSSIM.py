"""`model.lossesSSIM` boundary (reference: model/lossesSSIM.py): `ssim`, `ssim3D`, `SSIM`, `SSIM3D` with the reference's signatures.
Forward and backward run in libsegengine (csrc/ssim.hip: separable Gaussian passes + one map pass; the backward re-blurs the derivative
maps).  `size_average=False` returns the per-sample means for 4-D inputs; for 5-D inputs the reference's `.mean(1).mean(1).mean(1)` leaves
an (N, W) tensor (means over channel, depth and height) - reproduced as it is.  A backward node can be traversed more than once."""
import torch

from . import _capi
from .engine import aligned_empty


class _SsimFn(torch.autograd.Function):
    """reduce: 0 = mean over everything, 1 = per-sample means, 2 = means over (C, D, H) per (n, w) (the reference's size_average=False on a 5-D map)."""

    @staticmethod
    def forward(ctx, img1, img2, window_size, reduce, nd):
        a, b = img1.float().contiguous(), img2.float().contiguous()
        assert a.shape == b.shape and a.dim() == nd + 2
        n, c = a.shape[0], a.shape[1]
        d, h, w = (1,) + tuple(a.shape[2:]) if nd == 2 else tuple(a.shape[2:])
        lib = _capi.lib_for(a.device)
        ws = aligned_empty(lib.seg_ssim_ws_bytes(n, c, d * h * w), a.device)
        ctx.geom = (n, c, d, h, w, nd, int(window_size), int(reduce))
        ctx.lib, ctx.ws, ctx.consumed = lib, ws, False
        ctx.dtypes = (img1.dtype, img2.dtype)
        ctx.save_for_backward(a, b)               # autograd then notices in-place edits of the inputs between forward and backward
        out, cols = _SsimFn._run_forward(ctx, a, b)
        return out[0].clone() if reduce == 0 else (out[1:].clone() if reduce == 1 else cols)

    @staticmethod
    def _run_forward(ctx, a, b):
        n, c, d, h, w, nd, win, reduce = ctx.geom
        out = torch.zeros(1 + n, dtype=torch.float32, device=a.device)
        st = _capi.stream_for(a.device)
        if reduce == 2:
            cols = torch.empty((n, w), dtype=torch.float32, device=a.device)
            ctx.lib.check(ctx.lib.seg_ssim_forward_cols(a.data_ptr(), b.data_ptr(), n, c, d, h, w, nd, win, ctx.ws.data_ptr(), out.data_ptr(),
                                                        cols.data_ptr(), st), "seg_ssim_forward_cols")
            return out, cols
        ctx.lib.check(ctx.lib.seg_ssim_forward(a.data_ptr(), b.data_ptr(), n, c, d, h, w, nd, win, ctx.ws.data_ptr(), out.data_ptr(), st), "seg_ssim_forward")
        return out, None

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n, c, d, h, w, nd, win, reduce = ctx.geom
        if ctx.consumed:
            # seg_ssim_backward blurs the derivative maps of the forward pass IN PLACE inside the workspace: a second backward through this node
            # (retain_graph=True, two autograd.grad calls) first rebuilds them
            _SsimFn._run_forward(ctx, a, b)
        ctx.consumed = True
        per = c * d * h * w
        if reduce == 0:
            gs, mode = g.reshape(1).float() / (per * n), 0
        elif reduce == 1:
            gs, mode = g.reshape(n).float() / per, 1
        else:
            gs, mode = g.reshape(n, w).float() / (c * d * h), w           # gscale[n][x]; the row length doubles as the mode
            if w < 2:
                gs, mode = gs.reshape(n), 1
        gs = gs.contiguous()
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d1 = torch.empty_like(a) if need1 else None
        d2 = torch.empty_like(b) if need2 else None
        ctx.lib.check(ctx.lib.seg_ssim_backward(a.data_ptr(), b.data_ptr(), n, c, d, h, w, nd, win, ctx.ws.data_ptr(), gs.data_ptr(), mode,
                                                d1.data_ptr() if need1 else None, d2.data_ptr() if need2 else None, _capi.stream_for(a.device)),
                      "seg_ssim_backward")
        return (d1.to(ctx.dtypes[0]) if need1 else None), (d2.to(ctx.dtypes[1]) if need2 else None), None, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """model/lossesSSIM.py:146-155"""
    return _SsimFn.apply(img1, img2, window_size, 0 if size_average else 1, 2)


def ssim3D(img1, img2, window_size=11, size_average=True):
    """model/lossesSSIM.py:158-167"""
    # size_average=False: the reference's `.mean(1).mean(1).mean(1)` of the 5-D map leaves an (N, W) tensor - reproduced as it is
    return _SsimFn.apply(img1, img2, window_size, 0 if size_average else 2, 3)


class SSIM(torch.nn.Module):
    """model/lossesSSIM.py:102-122"""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average, self.channel = window_size, size_average, 1

    def forward(self, img1, img2):
        self.channel = img1.size(1)
        return ssim(img1, img2, self.window_size, self.size_average)


class SSIM3D(torch.nn.Module):
    """model/lossesSSIM.py:125-143"""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average, self.channel = window_size, size_average, 1

    def forward(self, img1, img2):
        self.channel = img1.size(1)
        return ssim3D(img1, img2, self.window_size, self.size_average)

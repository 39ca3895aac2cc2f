"""`model.lossesSSIM` boundary (reference: model/lossesSSIM.py): `ssim`, `ssim3D`, `SSIM`, `SSIM3D` with the reference's signatures.
Forward and backward run in libsegengine (csrc/ssim.hip: separable Gaussian passes + one map pass; the backward re-blurs the derivative
maps).  `size_average=False` returns the per-sample means for 4-D inputs; for 5-D inputs the reference's `.mean(1).mean(1).mean(1)` leaves
an (N, W) tensor - not built (NotImplementedError)."""
import torch

from . import _capi
from .engine import aligned_empty


class _SsimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, window_size, size_average, nd):
        a, b = img1.float().contiguous(), img2.float().contiguous()
        assert a.shape == b.shape and a.dim() == nd + 2
        n, c = a.shape[0], a.shape[1]
        d, h, w = (1,) + tuple(a.shape[2:]) if nd == 2 else tuple(a.shape[2:])
        lib = _capi.lib_for(a.device)
        nbytes = lib.seg_ssim_ws_bytes(n, c, d * h * w)
        ws = aligned_empty(nbytes, a.device)
        out = torch.zeros(1 + n, dtype=torch.float32, device=a.device)
        lib.check(lib.seg_ssim_forward(a.data_ptr(), b.data_ptr(), n, c, d, h, w, nd, int(window_size), ws.data_ptr(), out.data_ptr(),
                                       _capi.stream_for(a.device)), "seg_ssim_forward")
        ctx.stuff = (a, b, n, c, d, h, w, nd, int(window_size), ws, lib, bool(size_average))
        ctx.dtypes = (img1.dtype, img2.dtype)
        return out[0].clone() if size_average else out[1:].clone()

    @staticmethod
    def backward(ctx, g):
        a, b, n, c, d, h, w, nd, win, ws, lib, avg = ctx.stuff
        per = c * d * h * w
        gs = (g.reshape(1).float() / (per * n)) if avg else (g.reshape(n).float() / per)
        gs = gs.contiguous()
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d1 = torch.empty_like(a) if need1 else None
        d2 = torch.empty_like(b) if need2 else None
        lib.check(lib.seg_ssim_backward(a.data_ptr(), b.data_ptr(), n, c, d, h, w, nd, win, ws.data_ptr(), gs.data_ptr(), 0 if avg else 1,
                                        d1.data_ptr() if need1 else None, d2.data_ptr() if need2 else None, _capi.stream_for(a.device)),
                  "seg_ssim_backward")
        return (d1.to(ctx.dtypes[0]) if need1 else None), (d2.to(ctx.dtypes[1]) if need2 else None), None, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """model/lossesSSIM.py:146-155"""
    return _SsimFn.apply(img1, img2, window_size, size_average, 2)


def ssim3D(img1, img2, window_size=11, size_average=True):
    """model/lossesSSIM.py:158-167"""
    if not size_average:
        raise NotImplementedError("ssim3D(size_average=False): the reference reduces dims 1,1,1 of a 5-D map, which leaves (N, W); not built")
    return _SsimFn.apply(img1, img2, window_size, size_average, 3)


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

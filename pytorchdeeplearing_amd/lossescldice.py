"""`model.lossescldice` boundary: soft-clDice as the reference *intends* it
(model/lossescldice.py:5-86).  The reference file cannot execute as shipped (rank test on a list, `__int__`
constructors, max-pool on a Long one-hot); the repairs are listed in oracle/make_golden.py:CLDICE_REPAIRS and this
module implements the repaired behaviour.  All volume-sized work — the 10 min/max-pool skeleton iterations, their
backward (gradient routed to the first extremum of each window, as ATen does), the per-plane reductions and the
per-plane affine gradient maps — runs in libsegengine (csrc/cldice.hip); only the O(batch x channel) ratio arithmetic
is done with torch scalars-on-device here."""
import torch
from torch import nn

from . import _capi

THRESH_WIDTH = 10


def _dims(x):
    """[planes, D, H, W, nd] for a 4-D (N,C,H,W) or 5-D (N,C,D,H,W) tensor."""
    if x.dim() == 4:
        return x.shape[0] * x.shape[1], 1, x.shape[2], x.shape[3], 2
    if x.dim() == 5:
        return x.shape[0] * x.shape[1], x.shape[2], x.shape[3], x.shape[4], 3
    raise ValueError("soft_skeletonize expects a 4-D or 5-D tensor")


def _lib(x):
    return _capi.lib_for(x.device)                     # raises for CPU tensors: there is no CPU implementation


class _SoftSkel(torch.autograd.Function):
    """model/lossescldice.py:5-21."""

    @staticmethod
    def forward(ctx, x, width):
        xin = x.float().contiguous()
        lib, st = _lib(xin), _capi.stream_for(xin.device)
        p, d, h, w, nd = _dims(xin)
        # ONE work buffer for the whole iteration chain ([iteration][e | next x]) instead of 2 x width allocations per call: the
        # caching allocator was the dominant cost whenever the tensor shape changed between calls (13-16 ms vs 2.3 ms of kernels,
        # profiles/r01_cldice_step23_shapes.jsonl)
        work = torch.empty((width, 2) + tuple(xin.shape), dtype=torch.float32, device=xin.device)
        xs, es = [], []
        cur = xin
        for it in range(width):
            e, nxt = work[it, 0], work[it, 1]
            lib.check(lib.seg_op_skel_iter(cur.data_ptr(), e.data_ptr(), nxt.data_ptr(), p, d, h, w, nd, st), "seg_op_skel_iter")
            xs.append(cur)
            es.append(e)
            cur = nxt
        if width == 0:
            cur = xin.clone()
        ctx.xs, ctx.es, ctx.geom, ctx.in_dtype = xs, es, (p, d, h, w, nd), x.dtype
        return cur

    @staticmethod
    def backward(ctx, g):
        g = g.float().contiguous()
        lib, st = _lib(g), _capi.stream_for(g.device)
        p, d, h, w, nd = ctx.geom
        # work buffers: de for the gather passes (fully overwritten every iteration) and two ping-pong gradient buffers
        buf = torch.empty((3,) + tuple(g.shape), dtype=torch.float32, device=g.device)
        de = buf[2]
        for i, (x, e) in enumerate(zip(reversed(ctx.xs), reversed(ctx.es))):
            dx = buf[i & 1]
            lib.check(lib.seg_op_skel_iter_bwd(g.data_ptr(), x.data_ptr(), e.data_ptr(), dx.data_ptr(), de.data_ptr(), p, d, h, w, nd, st),
                      "seg_op_skel_iter_bwd")
            g = dx
        g = g.clone() if len(ctx.xs) else g               # detach the result from the work buffer
        return g.to(ctx.in_dtype), None


def soft_skeletonize(x, thresh_width=THRESH_WIDTH):
    if x.dim() not in (4, 5):
        return x                                       # the reference leaves other ranks untouched (lossescldice.py:11-21)
    return _SoftSkel.apply(x, thresh_width)


class _NormIntersection(torch.autograd.Function):
    """model/lossescldice.py:24-35: (sum(cl*v) + 1) / (sum(cl) + 1) per (batch, dim-1) plane."""

    @staticmethod
    def forward(ctx, cl, v):
        clf, vf = cl.float().contiguous(), v.float().contiguous()
        lib, st = _lib(clf), _capi.stream_for(clf.device)
        n, c = clf.shape[0], clf.shape[1]
        vol = clf.numel() // (n * c)
        sums = torch.empty((n * c, 2), dtype=torch.float64, device=clf.device)
        scratch = torch.empty(lib.seg_op_plane_dot_scratch_bytes(n * c, vol) // 8 + 2, dtype=torch.float64, device=clf.device)
        lib.check(lib.seg_op_plane_dot(clf.data_ptr(), vf.data_ptr(), sums.data_ptr(), scratch.data_ptr(), n * c, vol, st), "seg_op_plane_dot")
        inter, s = sums[:, 0] + 1.0, sums[:, 1] + 1.0
        ctx.save_for_backward(clf, vf, inter, s)
        ctx.meta = (n, c, vol, cl.dtype, v.dtype)
        return (inter / s).float().reshape(n, c)

    @staticmethod
    def backward(ctx, g):
        clf, vf, inter, s = ctx.saved_tensors
        n, c, vol, cdt, vdt = ctx.meta
        lib, st = _lib(clf), _capi.stream_for(clf.device)
        gp = g.reshape(-1).double()
        dcl = dv = None
        if ctx.needs_input_grad[0]:                    # d/dcl = g * (v * s - inter) / s^2
            a = (gp / s).float().contiguous()
            b = (-gp * inter / (s * s)).float().contiguous()
            dcl = torch.empty_like(clf)
            lib.check(lib.seg_op_plane_axpb(vf.data_ptr(), a.data_ptr(), b.data_ptr(), dcl.data_ptr(), n * c, vol, 0, st), "seg_op_plane_axpb")
            dcl = dcl.to(cdt)
        if ctx.needs_input_grad[1]:                    # d/dv = g * cl / s
            a = (gp / s).float().contiguous()
            b = torch.zeros_like(a)
            dv = torch.empty_like(vf)
            lib.check(lib.seg_op_plane_axpb(clf.data_ptr(), a.data_ptr(), b.data_ptr(), dv.data_ptr(), n * c, vol, 0, st), "seg_op_plane_axpb")
            dv = dv.to(vdt)
        return dcl, dv


def norm_intersection(center_line, vessel):
    return _NormIntersection.apply(center_line, vessel)


class Binary_Soft_cldice_loss(nn.Module):
    """model/lossescldice.py:38-61 (`pred` holds probabilities, `target` has pred's shape)."""

    def __init__(self):
        super().__init__()
        self.smooth = 1e-5
        self.eps = 1e-7

    def forward(self, pred, target):
        target = target.to(pred.dtype)
        cl_pred = soft_skeletonize(pred)
        with torch.no_grad():
            target_skeleton = soft_skeletonize(target)
        iflat = norm_intersection(cl_pred, target)
        tflat = norm_intersection(target_skeleton, pred)
        intersection = iflat * tflat
        cldsc = (2. * intersection.sum() + self.smooth) / (iflat.sum() + tflat.sum() + self.smooth).clamp_min(self.eps)
        return (1. - cldsc).mean()


class Mutil_Soft_cldice_loss(nn.Module):
    """model/lossescldice.py:64-86: per-class binary clDice on `input[:, c]` (class axis dropped, as written), alpha-weighted."""

    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha
        self.bscldice = Binary_Soft_cldice_loss()

    def forward(self, input, target):
        n, c = input.shape[0], input.shape[1]
        y_true = torch.nn.functional.one_hot(target.long().reshape(n, -1), c).permute(0, 2, 1).reshape(input.shape).to(input.dtype)
        dice = 0
        for ch in range(c):
            dice = dice + self.bscldice(input[:, ch, ...], y_true[:, ch, ...]) * float(self.alpha[ch])
        return dice / c

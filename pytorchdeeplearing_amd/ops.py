"""Operator-level host bindings (seg_op_conv / seg_op_wgrad / seg_op_pack of include/segengine.h).

Tensors are channels-last [N][D][H][W][C] in the run dtype (2-D: D = 1).  These are the same kernel
launches the network engine issues; they exist so every kernel can be checked in isolation against
torch.nn.functional (the reference's operator layer: networks/VNet3d.py:8,28,29,49,65,70,88)."""
import ctypes as C

import torch

from . import _capi
from .engine import aligned_empty

TORCH_DTYPE = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


class Taps(C.Structure):
    _fields_ = [("n", C.c_int), ("d", C.c_byte * 27), ("h", C.c_byte * 27), ("w", C.c_byte * 27)]


class ConvArgs(C.Structure):
    _fields_ = [("in0", C.c_void_p), ("in1", C.c_void_p), ("C0", C.c_int), ("C1", C.c_int), ("w", C.c_void_p),
                ("bias", C.c_void_p), ("out", C.c_void_p), ("stats", C.c_void_p),
                ("N", C.c_int), ("ID", C.c_int), ("IH", C.c_int), ("IW", C.c_int),
                ("OD", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
                ("FD", C.c_int), ("FH", C.c_int), ("FW", C.c_int),
                ("Cout", C.c_int), ("Ngemm", C.c_int), ("K", C.c_int), ("Kpad", C.c_int),
                ("sd", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("scatter", C.c_int), ("taps", Taps),
                ("act_scale", C.c_void_p), ("act_shift", C.c_void_p), ("out1", C.c_void_p), ("Cout0", C.c_int),
                ("rq_r", C.c_void_p), ("rq_scale", C.c_void_p), ("rq_shift", C.c_void_p), ("rq_Q", C.c_void_p)]


class WgradArgs(C.Structure):
    _fields_ = [("dr", C.c_void_p), ("x0", C.c_void_p), ("x1", C.c_void_p), ("C0", C.c_int), ("C1", C.c_int),
                ("dw", C.c_void_p), ("P", C.c_int), ("Q", C.c_int),
                ("N", C.c_int), ("ID", C.c_int), ("IH", C.c_int), ("IW", C.c_int),
                ("OD", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
                ("sd", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("taps", Taps),
                ("sP", C.c_longlong), ("sQ", C.c_longlong), ("sT", C.c_longlong), ("stem", C.c_int),
                ("act_scale", C.c_void_p), ("act_shift", C.c_void_p)]


class PackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("R1", C.c_int), ("R2", C.c_int), ("T", C.c_int),
                ("Cc", C.c_int), ("Kpad", C.c_int), ("s1", C.c_longlong), ("s2", C.c_longlong),
                ("sT", C.c_longlong), ("sC", C.c_longlong), ("flipT", C.c_int), ("csrc", C.c_int), ("frag", C.c_int)]


class StemxArgs(C.Structure):
    _fields_ = [("img", C.c_void_p), ("w3", C.c_void_p), ("w1", C.c_void_p), ("bias3", C.c_void_p), ("bias1", C.c_void_p),
                ("stats3", C.c_void_p), ("stats1", C.c_void_p),
                ("scale3", C.c_void_p), ("shift3", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p),
                ("out", C.c_void_p), ("dy", C.c_void_p * 3), ("ndy", C.c_int),
                ("Q3", C.c_void_p), ("Q1", C.c_void_p), ("coef3", C.c_void_p), ("coef1", C.c_void_p), ("partial", C.c_void_p),
                ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cimg", C.c_int)]


def check_abi(lib):
    assert lib.dll.seg_abi_sizeof(3) == C.sizeof(StemxArgs), (lib.dll.seg_abi_sizeof(3), C.sizeof(StemxArgs))
    assert lib.dll.seg_abi_sizeof(0) == C.sizeof(ConvArgs), (lib.dll.seg_abi_sizeof(0), C.sizeof(ConvArgs))
    assert lib.dll.seg_abi_sizeof(1) == C.sizeof(WgradArgs)
    assert lib.dll.seg_abi_sizeof(2) == C.sizeof(PackDesc)
    assert lib.dll.seg_abi_sizeof(4) == C.sizeof(_capi.TrainArgs), (lib.dll.seg_abi_sizeof(4), C.sizeof(_capi.TrainArgs))


def make_taps(ndim, k, pad):
    t = Taps()
    n = 0
    for a in range(k if ndim == 3 else 1):
        for b in range(k):
            for c in range(k):
                t.d[n] = (a - pad) if ndim == 3 else 0
                t.h[n] = b - pad
                t.w[n] = c - pad
                n += 1
    t.n = n
    return t


def _kpad(k):
    return (k + 31) // 32 * 32


def _alloc(shape, dtype, device, zero=False):
    n = 1
    for s in shape:
        n *= s
    t = aligned_empty(n * torch.empty((), dtype=dtype).element_size(), device).view(dtype).view(shape)
    if zero:
        t.zero_()
    return t


def aligned_like(t, dtype=None):
    o = _alloc(tuple(t.shape), dtype or t.dtype, t.device)
    o.copy_(t)
    return o


def pack(w, layout, dtype, frag=False):
    """Re-layout an fp32 PyTorch-layout conv / conv-transpose weight for the GEMM kernels.
    layouts: conv_fwd [co][(t,ci)] | conv_dgrad [ci][(flip t,co)] | k2s2_dgrad [(a,ci)][co] |
             convT_fwd [(a,co)][ci] | convT_dgrad [ci][(a,co)]
    frag=True: the same elements in MFMA-fragment-major order (conv3x: layout 1, or 2 for 16 channels); frag=3: the row-reuse layout of the
    16-channel 3-D tilings 28 .. 31 (seg_op_conv3x_cfg_frag); frag="all": a dict {layout: tensor} of every layout the shape has (conv3x picks by tiling)."""
    if frag == "all":
        out = {}
        t = pack(w, layout, dtype, frag=True)
        out[t._seg_frag] = t
        if t._seg_frag == 2 and w.dim() == 5:
            out[3] = pack(w, layout, dtype, frag=3)
        return out
    lib = _capi.lib_for(w.device)
    w = aligned_like(w.float().contiguous())
    A, B = w.shape[0], w.shape[1]
    T = 1
    for s in w.shape[2:]:
        T *= s
    d = PackDesc()
    d.src = w.data_ptr()
    if layout == "conv_fwd":            # w (Cout, Cin, k..)
        rows, d.R1, d.R2, d.T, d.Cc = A, A, 1, T, B
        d.s1, d.s2, d.sT, d.sC, d.flipT = B * T, 0, 1, T, 0
    elif layout == "conv_dgrad":
        rows, d.R1, d.R2, d.T, d.Cc = B, B, 1, T, A
        d.s1, d.s2, d.sT, d.sC, d.flipT = T, 0, 1, B * T, 1
    elif layout == "k2s2_dgrad":
        rows, d.R1, d.R2, d.T, d.Cc = T * B, T, B, 1, A
        d.s1, d.s2, d.sT, d.sC, d.flipT = 1, T, 0, B * T, 0
    elif layout == "convT_fwd":         # w (Cin, Cout, k..)
        rows, d.R1, d.R2, d.T, d.Cc = T * B, T, B, 1, A
        d.s1, d.s2, d.sT, d.sC, d.flipT = 1, T, 0, B * T, 0
    elif layout == "convT_dgrad":
        rows, d.R1, d.R2, d.T, d.Cc = A, A, 1, T, B
        d.s1, d.s2, d.sT, d.sC, d.flipT = B * T, 0, 1, T, 0
    else:
        raise ValueError(layout)
    d.Kpad = _kpad(d.T * d.Cc)
    d.frag = 3 if (frag == 3 and frag is not True) else ((2 if d.Cc == 16 else 1) if frag else 0)
    if frag:
        assert (d.Cc % 32 == 0 or d.Cc == 16) and rows % 16 == 0, "fragment-major packing needs Cc % 32 == 0 (or 16) and rows % 16 == 0"
    if d.frag == 3:
        assert d.Cc == 16 and d.T == 27, "layout 3 is the 16-channel 3-D layout"
        d.Kpad = 480
    out = _alloc((rows, d.Kpad), TORCH_DTYPE[dtype], w.device)
    d.dst = out.data_ptr()
    raw = bytes(d)
    dev = aligned_empty(len(raw), w.device)
    dev.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    lib.check(lib.dll.seg_op_pack(C.c_void_p(dev.data_ptr()), 1, C.c_longlong(rows * d.Kpad), _capi.DTYPE[dtype],
                                  _capi.stream_for(w.device)), "seg_op_pack")
    out._seg_frag = int(d.frag)
    return out


last_conv_kernel = None


def conv(x0, wpacked, dtype, ndim, k, stride=1, pad=0, x1=None, bias=None, cout=None, scatter=False, want_stats=False, act=None, split=0, rq=None):
    """x0 (and optional concat source x1): [N,D,H,W,C] in dtype.  Gather conv (k, stride, pad) or, with
    scatter=True, the k2-s2 transposed conv.  Returns out [N,OD,OH,OW,cout] (+ stats [N,cout,2] fp64).
    act = (scale, shift), fp32 [N, C0]: x0 is read as relu(scale * x0 + shift) rounded to dtype (streaming kernel only).
    split = c (a multiple of 16): returns (out[..., :c], out[..., c:]) written as two tensors by one launch (streaming kernel, gather form, no bias / stats).
    rq = (r, scale, shift) with split: also returns Q [N, c, 2] fp64 = sum of out0 * [scale * r + shift > 0] (* r) over the voxels (GroupNorm-backward sums)."""
    lib = _capi.lib_for(x0.device)
    N, D, H, W, C0 = x0.shape
    a = ConvArgs()
    a.in0, a.C0 = x0.data_ptr(), C0
    a.in1, a.C1 = (x1.data_ptr(), x1.shape[-1]) if x1 is not None else (None, 0)
    cin = a.C0 + a.C1
    a.w = wpacked.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    if act is not None:
        a.act_scale, a.act_shift = act[0].data_ptr(), act[1].data_ptr()
    a.N, a.ID, a.IH, a.IW = N, D, H, W
    a.Cout = cout
    up = lambda v, dim3: v * 2 if (dim3 or ndim == 3) else v
    if scatter:
        a.scatter = 1
        a.OD, a.OH, a.OW = D, H, W
        a.FD, a.FH, a.FW = (D * 2 if ndim == 3 else 1), H * 2, W * 2
        a.sd, a.sh, a.sw = (2 if ndim == 3 else 1), 2, 2
        a.taps = make_taps(ndim, 2, 0)
        a.K, a.Ngemm = cin, a.taps.n * cout
        oshape = (N, a.FD, a.FH, a.FW, cout)
    else:
        a.scatter = 0
        od = (D + 2 * pad - k) // stride + 1 if ndim == 3 else 1
        a.OD, a.OH, a.OW = od, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        a.sd, a.sh, a.sw = (stride if ndim == 3 else 1), stride, stride
        a.taps = make_taps(ndim, k, pad)
        a.K, a.Ngemm = a.taps.n * cin, cout
        oshape = (N, a.OD, a.OH, a.OW, cout)
    a.Kpad = _kpad(a.K)
    assert tuple(wpacked.shape) == (a.Ngemm, a.Kpad), (tuple(wpacked.shape), a.Ngemm, a.Kpad)
    out = _alloc(oshape if not split else oshape[:-1] + (split,), TORCH_DTYPE[dtype], x0.device, zero=True)
    a.out = out.data_ptr()
    out1 = None
    if split:
        out1 = _alloc(oshape[:-1] + (cout - split,), TORCH_DTYPE[dtype], x0.device, zero=True)
        a.out1, a.Cout0 = out1.data_ptr(), split
    q = None
    if rq is not None:
        q = _alloc((32, N, split, 2), torch.float64, x0.device, zero=True)
        a.rq_r, a.rq_scale, a.rq_shift, a.rq_Q = rq[0].data_ptr(), rq[1].data_ptr(), rq[2].data_ptr(), q.data_ptr()
    stats = None
    if want_stats:
        stats = _alloc((32, N, cout, 2), torch.float64, x0.device, zero=True)
        a.stats = stats.data_ptr()
    global last_conv_kernel
    last_conv_kernel = lib.dll.seg_op_conv_kernel(C.byref(a))      # 1: streaming kernel, 0: LDS-staged implicit GEMM
    lib.check(lib.dll.seg_op_conv(C.byref(a), _capi.DTYPE[dtype], _capi.stream_for(x0.device)), "seg_op_conv")
    if split:
        return (out, out1, q.sum(0)) if q is not None else (out, out1)
    return (out, stats.sum(0)) if want_stats else out


def wgrad(dr, x0, dtype, ndim, k, stride=1, pad=0, x1=None, stem=False, act=None):
    """dW[p][q][tap] (PyTorch conv-weight layout (P, Q, k..)) = sum_m dr[m][p] * x[vox(m,tap)][q].
    act = (scale, shift), fp32 [N, C0]: x0 is read as relu(scale * x0 + shift) rounded to dtype (1^d convs on 16-bit tensors only)."""
    lib = _capi.lib_for(dr.device)
    N, OD, OH, OW, P = dr.shape
    _, D, H, W, C0 = x0.shape
    a = WgradArgs()
    a.dr, a.x0, a.C0 = dr.data_ptr(), x0.data_ptr(), C0
    a.x1, a.C1 = (x1.data_ptr(), x1.shape[-1]) if x1 is not None else (None, 0)
    a.taps = make_taps(ndim, k, pad)
    T = a.taps.n
    qc = C0 + a.C1
    a.P, a.Q = P, (T * qc if stem else qc)
    a.N, a.ID, a.IH, a.IW, a.OD, a.OH, a.OW = N, D, H, W, OD, OH, OW
    a.sd, a.sh, a.sw = (stride if ndim == 3 else 1), stride, stride
    a.sP, a.sQ, a.sT = qc * T, T, 1
    a.stem = 1 if stem else 0
    if act is not None:
        a.act_scale, a.act_shift = act[0].data_ptr(), act[1].data_ptr()
    dw = _alloc((P, qc) + (k,) * ndim, torch.float32, dr.device, zero=True)
    a.dw = dw.data_ptr()
    scratch = aligned_empty(lib.seg_op_wgrad_partial_bytes(C.byref(a)), dr.device)
    lib.check(lib.seg_op_wgrad(C.byref(a), scratch.data_ptr(), _capi.DTYPE[dtype], _capi.stream_for(dr.device)), "seg_op_wgrad")
    return dw


def conv3(x, wpacked, dtype, ndim, cout, bias=None, want_stats=False, out=None):
    """LDS halo-tile 3^d / 3^2 stride-1 pad-1 conv (forward, or data-gradient with the conv_dgrad layout)."""
    lib = _capi.lib_for(x.device)
    N, D, H, W, cin = x.shape
    if out is None:
        out = _alloc((N, D, H, W, cout), TORCH_DTYPE[dtype], x.device, zero=True)
    stats = _alloc((32, N, cout, 2), torch.float64, x.device, zero=True) if want_stats else None
    lib.check(lib.seg_op_conv3(x.data_ptr(), wpacked.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                               stats.data_ptr() if stats is not None else None, N, D, H, W, cin, cout, ndim, _capi.DTYPE[dtype],
                               _capi.stream_for(x.device)), "seg_op_conv3")
    return (out, stats.sum(0)) if want_stats else out


def wgrad3(dr, x, dtype, ndim, x1=None):
    """weight gradient of the 3^d / 3^2 stride-1 pad-1 conv in PyTorch layout (P, Q, k..); x1: second concat source of x."""
    lib = _capi.lib_for(dr.device)
    N, D, H, W, P = dr.shape
    Q = x.shape[-1] + (x1.shape[-1] if x1 is not None else 0)
    if x1 is not None:
        nbytes = lib.seg_op_wgrad3_partial_bytes(ndim, N, D, H, W, P, Q)
        partial = aligned_empty(nbytes, dr.device).view(torch.float32)
        dw = _alloc((P, Q) + (3,) * ndim, torch.float32, dr.device, zero=True)
        lib.check(lib.seg_op_wgrad3_cat(dr.data_ptr(), x.data_ptr(), x1.data_ptr(), x.shape[-1], partial.data_ptr(), dw.data_ptr(), N, D, H, W,
                                        P, Q, ndim, _capi.DTYPE[dtype], _capi.stream_for(dr.device)), "seg_op_wgrad3_cat")
        return dw
    nbytes = lib.seg_op_wgrad3_partial_bytes(ndim, N, D, H, W, P, Q)
    partial = aligned_empty(nbytes, dr.device).view(torch.float32)
    dw = _alloc((P, Q) + (3,) * ndim, torch.float32, dr.device, zero=True)
    lib.check(lib.seg_op_wgrad3(dr.data_ptr(), x.data_ptr(), partial.data_ptr(), dw.data_ptr(), N, D, H, W, P, Q, ndim,
                                _capi.DTYPE[dtype], _capi.stream_for(dr.device)), "seg_op_wgrad3")
    return dw


def conv3x_cfgs(device):
    """The tilings of the register-blocked halo conv: list of dicts (id, ndim, box, bn, nres, name)."""
    lib = _capi.lib_for(device)
    out = []
    for i in range(lib.seg_op_conv3x_num_cfgs()):
        cid, nd, bn, nres = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        box = (C.c_int * 3)()
        name = C.create_string_buffer(96)
        lib.check(lib.seg_op_conv3x_cfg_info(i, C.byref(cid), C.byref(nd), box, C.byref(bn), C.byref(nres), name, 96), "cfg_info")
        out.append(dict(id=cid.value, ndim=nd.value, box=tuple(box), bn=bn.value, nres=nres.value, name=name.value.decode()))
    return out


def conv3x(x, wfrag, dtype, ndim, cout, bias=None, want_stats=False, out=None, x1=None, cfg=-1):
    """Register-blocked halo conv (16-bit, Cin % 32 == 0 or 16); wfrag = pack(..., frag=True / 3 / "all") in the layout the tiling reads
    (seg_op_conv3x_cfg_frag; a dict from frag="all" is indexed by it).  x1: second concat source."""
    lib = _capi.lib_for(x.device)
    N, D, H, W, c0 = x.shape
    cin = c0 + (x1.shape[-1] if x1 is not None else 0)
    if cfg < 0:
        dflt = lib.dll.seg_op_conv3x_default_cfg(ndim, N, D, H, W, cin, cout, _capi.DTYPE[dtype])
        cfg = dflt if dflt >= 0 else cfg
    need = lib.dll.seg_op_conv3x_cfg_frag(cfg) if cfg >= 0 else 0
    if isinstance(wfrag, dict):
        if need not in wfrag:
            raise RuntimeError("conv3x: tiling %d reads weight layout %d, packed: %s" % (cfg, need, sorted(wfrag)))
        wfrag = wfrag[need]
    elif need and getattr(wfrag, "_seg_frag", need) != need:
        raise RuntimeError("conv3x: tiling %d reads weight layout %d, the weights are packed in layout %d" % (cfg, need, wfrag._seg_frag))
    if out is None:
        out = _alloc((N, D, H, W, cout), TORCH_DTYPE[dtype], x.device, zero=True)
    stats = _alloc((32, N, cout, 2), torch.float64, x.device, zero=True) if want_stats else None
    lib.check(lib.seg_op_conv3x(cfg, x.data_ptr(), x1.data_ptr() if x1 is not None else None, c0, wfrag.data_ptr(),
                                bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                stats.data_ptr() if stats is not None else None, N, D, H, W, cin, cout, ndim, _capi.DTYPE[dtype],
                                _capi.stream_for(x.device)), "seg_op_conv3x")
    return (out, stats.sum(0)) if want_stats else out


def stemx(mode, img, w3p, dtype, ndim, w1p=None, bias3=None, bias1=None, scale=None, shift=None, dys=(), coef=None):
    """Fused input block (csrc/stemx.hip).  img [N,D,H,W,Cimg] run dtype; w3p / w1p packed "conv_fwd" rows; scale / shift / coef:
    pairs (branch 3, branch 1) of fp32 [N,16] / [N,16,3] tensors.  Returns per mode:
    0 -> (stats3, stats1) fp64 [N,16,2];  1 -> out [N,D,H,W,16];  2 -> (Q3, Q1) fp64 [N,16,2];  3 -> (dw3, dw1) fp32 PyTorch layout."""
    lib = _capi.lib_for(img.device)
    N, D, H, W, cimg = img.shape
    dev = img.device
    a = StemxArgs()
    a.img, a.w3, a.w1 = img.data_ptr(), w3p.data_ptr(), (w1p.data_ptr() if w1p is not None else None)
    a.bias3 = bias3.data_ptr() if bias3 is not None else None
    a.bias1 = bias1.data_ptr() if bias1 is not None else None
    a.N, a.D, a.H, a.W, a.Cimg = N, D, H, W, cimg
    keep = []
    two = w1p is not None
    if scale is not None:
        a.scale3, a.shift3 = scale[0].data_ptr(), shift[0].data_ptr()
        if two:
            a.scale1, a.shift1 = scale[1].data_ptr(), shift[1].data_ptr()
    a.ndy = len(dys)
    for i, d in enumerate(dys):
        a.dy[i] = d.data_ptr()
    dw3 = dw1 = None
    if mode == 0:
        st = [_alloc((32, N, 16, 2), torch.float64, dev, zero=True) for _ in range(2)]
        a.stats3, a.stats1 = st[0].data_ptr(), st[1].data_ptr()
        res = lambda: (st[0].sum(0), st[1].sum(0))
    elif mode == 1:
        out = _alloc((N, D, H, W, 16), TORCH_DTYPE[dtype], dev, zero=True)
        a.out = out.data_ptr()
        res = lambda: out
    elif mode == 2:
        qs = [_alloc((32, N, 16, 2), torch.float64, dev, zero=True) for _ in range(2)]
        a.Q3, a.Q1 = qs[0].data_ptr(), qs[1].data_ptr()
        res = lambda: (qs[0].sum(0), qs[1].sum(0))
    else:
        a.coef3 = coef[0].data_ptr()
        if two:
            a.coef1 = coef[1].data_ptr()
        part = aligned_empty(lib.seg_op_stemx_partial_bytes(ndim, N, D, H, W, cimg), dev)
        a.partial = part.data_ptr()
        keep.append(part)
        dw3 = _alloc((16, cimg) + (3,) * ndim, torch.float32, dev, zero=True)
        dw1 = _alloc((16, cimg) + (1,) * ndim, torch.float32, dev, zero=True)
        res = lambda: (dw3, dw1)
    lib.check(lib.seg_op_stemx(C.byref(a), mode, ndim, _capi.DTYPE[dtype], dw3.data_ptr() if dw3 is not None else None,
                               dw1.data_ptr() if (dw1 is not None and two) else None, _capi.stream_for(dev)), "seg_op_stemx")
    return res()

"""`networks` boundary: VNet3d / VNet2d / UNet3d / UNet2d as nn.Modules with the
reference's constructor signatures, `forward(x) -> (logits, probs)` and bit-compatible `state_dict`
keys/shapes (networks/VNet3d.py:102-158, VNet2d.py, Unet3d.py:6-86, Unet2d.py of the reference), so
old `.pth` files load and `model.apply(initialize_weights)` keeps working.  The sub-modules are real
nn.Conv / nn.ConvTranspose / nn.GroupNorm objects that only HOLD parameters — their tensors are views
into the engine's flat fp32 buffer; all arithmetic runs in libsegengine (HIP, gfx950)."""
import os

import torch
from torch import nn

from . import _capi
from .engine import SegEngine


def _default_dtype():
    return os.environ.get("SEGENGINE_DTYPE", "f16")


class _Container(nn.Module):
    pass


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, eng, mask_mode, x, *params):
        logits, probs = eng.forward(x, mask_mode)
        ctx.net, ctx.eng = net, eng
        ctx.save_for_backward(probs)
        ctx.mark_non_differentiable()
        return logits, probs

    @staticmethod
    def backward(ctx, dlogits, dprobs):
        eng = ctx.eng
        (probs,) = ctx.saved_tensors
        dl = torch.zeros_like(probs) if dlogits is None else dlogits.float()
        if dprobs is not None:
            dp = dprobs.float()
            if probs.shape[1] == 1:
                dl = dl + dp * probs * (1 - probs)
            else:
                dl = dl + probs * (dp - (dp * probs).sum(1, keepdim=True))
        scale = eng.loss_scale
        eng.backward((dl * scale).contiguous(), zero_grads=True)
        inv = 1.0 / scale
        grads = [eng.param_view(name, grad=True) * inv for name in ctx.net._param_names]
        return (None, None, None, None) + tuple(grads)


class _SegNet(nn.Module):
    _kind = "vnet"
    _ndim = 3

    def __init__(self, in_channels, numclass, init_features=16, dtype=None):
        super().__init__()
        self.features = init_features
        self._dtype = dtype or _default_dtype()
        self._in_channels, self._numclass = in_channels, numclass
        self._engines = {}
        lib = _capi.host_library()
        table = _read_table(lib, self._kind, self._ndim, in_channels, numclass, init_features)
        self._param_names = list(table.keys())
        nd = self._ndim
        Conv = nn.Conv3d if nd == 3 else nn.Conv2d
        ConvT = nn.ConvTranspose3d if nd == 3 else nn.ConvTranspose2d
        done = set()
        for name, shape in table.items():
            leaf, attr = name.rsplit(".", 1)
            if leaf in done:
                continue
            done.add(leaf)
            wshape = table[leaf + ".weight"]
            has_bias = (leaf + ".bias") in table
            if len(wshape) == 1:
                mod = nn.GroupNorm(8, wshape[0])
            else:
                k = wshape[2]
                last = leaf.rsplit(".", 1)[-1]
                if last == "up_conv" or last.startswith("upconv"):
                    mod = ConvT(wshape[0], wshape[1], kernel_size=k, stride=2, bias=has_bias)
                else:
                    mod = Conv(wshape[1], wshape[0], kernel_size=k, stride=2 if k == 2 else 1, padding=1 if k == 3 else 0,
                               bias=has_bias)
            parent = self
            parts = leaf.split(".")
            for p in parts[:-1]:
                if not hasattr(parent, p):
                    parent.add_module(p, _Container())
                parent = getattr(parent, p)
            parent.add_module(parts[-1], mod)

    # ---- engine plumbing ------------------------------------------------------------------------
    def _engine_for(self, device):
        key = str(device)
        eng = self._engines.get(key)
        if eng is None:
            eng = SegEngine(self._kind, self._ndim, self._in_channels, self._numclass, self.features, self._dtype, device)
            self._engines = {key: eng}            # one live engine (the module lives on one device)
        return eng

    def _sync(self, eng):
        """make every Parameter a view of the engine's flat buffer (after .to(), load_state_dict, ...)."""
        sd = dict(self.named_parameters())
        for name in self._param_names:
            p = sd[name]
            view = eng.param_view(name)
            if p.data_ptr() != view.data_ptr():
                view.copy_(p.data.to(view.device))
                p.data = view
        eng.packed = False                           # weights may have been updated in place by an optimiser

    @property
    def engine(self):
        p = next(self.parameters())
        eng = self._engine_for(p.device)
        self._sync(eng)
        return eng

    def forward(self, x):
        eng = self._engine_for(x.device)
        self._sync(eng)
        x = x.float().contiguous()
        mask_mode = _capi.MASKS_RANDOM if self.training else _capi.MASKS_EVAL
        params = [p for _, p in self.named_parameters()]
        return _NetFn.apply(self, eng, mask_mode, x, *params)


def _read_table(lib, kind, ndim, in_ch, numclass, feat):
    import ctypes as C
    from collections import OrderedDict
    h = C.c_void_p()
    lib.check(lib.seg_create(_capi.NET_KIND[kind], ndim, in_ch, numclass, feat, 0, C.byref(h)), "seg_create")
    name = C.create_string_buffer(256)
    shape = (C.c_int * 8)()
    nd = C.c_int()
    off = C.c_longlong()
    t = OrderedDict()
    for i in range(lib.seg_param_count(h)):
        lib.check(lib.seg_param_info(h, i, name, 256, shape, C.byref(nd), C.byref(off)), "seg_param_info")
        t[name.value.decode()] = tuple(shape[:nd.value])
    lib.seg_destroy(h)
    return t


class VNet3d(_SegNet):
    """networks/VNet3d.py:102-158"""
    _kind, _ndim = "vnet", 3

    def __init__(self, image_channel, numclass, init_features=16, dtype=None):
        super().__init__(image_channel, numclass, init_features, dtype)
        self.image_channel, self.numclass = image_channel, numclass


class VNet2d(_SegNet):
    """networks/VNet2d.py:102-160"""
    _kind, _ndim = "vnet", 2

    def __init__(self, image_channel, numclass, init_features=16, dtype=None):
        super().__init__(image_channel, numclass, init_features, dtype)
        self.image_channel, self.numclass = image_channel, numclass


class UNet3d(_SegNet):
    """networks/Unet3d.py:6-86"""
    _kind, _ndim = "unet", 3

    def __init__(self, in_channels, out_channels, init_features=16, dtype=None):
        super().__init__(in_channels, out_channels, init_features, dtype)
        self.in_channels, self.out_channels = in_channels, out_channels


class UNet2d(_SegNet):
    """networks/Unet2d.py:6-85"""
    _kind, _ndim = "unet", 2

    def __init__(self, in_channels, out_channels, init_features=16, dtype=None):
        super().__init__(in_channels, out_channels, init_features, dtype)
        self.in_channels, self.out_channels = in_channels, out_channels


def initialize_weights(net):
    """networks/__init__.py:11-26 — kaiming_normal_(relu) for conv / conv-transpose weights, zero biases,
    GroupNorm weight 1 / bias 0; use as `model.apply(initialize_weights)`."""
    if isinstance(net, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose3d, nn.ConvTranspose2d)):
        nn.init.kaiming_normal_(net.weight.data, nonlinearity="relu")
        if net.bias is not None:
            nn.init.constant_(net.bias.data, 0)
    elif isinstance(net, (nn.BatchNorm2d, nn.BatchNorm3d, nn.BatchNorm1d, nn.GroupNorm)):
        nn.init.constant_(net.weight.data, 1)
        if net.bias is not None:
            nn.init.constant_(net.bias.data, 0)
    elif isinstance(net, nn.Linear):
        nn.init.kaiming_uniform_(net.weight.data)
        nn.init.constant_(net.bias.data, 0)

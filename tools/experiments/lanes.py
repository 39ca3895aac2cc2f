"""EXPERIMENT, not part of the package (moved out of pytorchdeeplearing_amd/ in round 5): measured 15-50 % slower than one engine on MI355X
(7.6 / 10 ms vs 6.6 ms per step with 2 / 4 lanes, round 1) - the small kernels of the lanes contend for the same CUs.

Intra-GPU batch lanes.

GroupNorm statistics and channel-dropout masks are per sample, so the network forward/backward of
disjoint batch slices are independent.  At 4 x 96^3 the 48^3 ... 6^3 levels launch kernels that cannot
fill 256 CUs (tens of workgroups, each a latency-bound chain), so `LaneEngine` runs the batch as L
slices ("lanes"), each on its own HIP stream with its own engine handle / workspace, all lanes sharing
ONE flat parameter buffer.  The loss (batch-global Dice / CE, model/losses.py:50-51) and the metric are
evaluated on the FULL batch of logits exactly as the reference does, gradients of the lanes are summed
into lane 0's flat gradient buffer and one fused optimiser step follows — the arithmetic is the same
optimisation step as a single engine at batch N (tests/test_lanes.py).
"""
import ctypes as C

import torch

from pytorchdeeplearing_amd import _capi
from pytorchdeeplearing_amd.engine import SegEngine, aligned_empty, _ptr


class LaneEngine:
    def __init__(self, kind, ndim, in_channels, numclass, init_features=16, dtype="f16", device="cuda", lanes=2):
        self.device = torch.device(device)
        self.lanes = lanes
        self.engines = [SegEngine(kind, ndim, in_channels, numclass, init_features, dtype, device) for _ in range(lanes)]
        e0 = self.engines[0]
        for e in self.engines[1:]:
            e.params = e0.params                     # one parameter buffer for all lanes (bound at plan time)
        self.lib, self.numclass, self.dtype = e0.lib, numclass, dtype
        self.cuda = self.device.type == "cuda"
        self.streams = [torch.cuda.Stream(self.device) for _ in range(lanes)] if self.cuda else [None] * lanes
        self._shape = None

    # ---- parameter plumbing (lane 0 owns the buffers) ----------------------------------------------
    @property
    def params(self):
        return self.engines[0].params

    @property
    def grads(self):
        return self.engines[0].grads

    @property
    def table(self):
        return self.engines[0].table

    def load_state_dict(self, sd):
        self.engines[0].load_state_dict(sd)
        for e in self.engines:
            e.packed = False

    def state_dict(self):
        return self.engines[0].state_dict()

    def grad_dict(self, unscale=True):
        return self.engines[0].grad_dict(unscale)

    @property
    def loss_scale(self):
        return self.engines[0].loss_scale

    @property
    def opt_state(self):
        return self.engines[0].opt_state

    def profile_enable(self, classes):
        for e in self.engines:
            e.profile_enable(classes)

    def profile_read(self):
        out = {}
        for e in self.engines:
            for k, v in e.profile_read().items():
                o = out.setdefault(k, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
                for f in o:
                    o[f] += v[f]
        return out

    # ---- helpers -------------------------------------------------------------------------------------
    def _slices(self, n):
        base, rem = divmod(n, self.lanes)
        out, s = [], 0
        for i in range(self.lanes):
            k = base + (1 if i < rem else 0)
            out.append((s, s + k))
            s += k
        return [sl for sl in out if sl[1] > sl[0]]

    def _prepare(self, x):
        key = tuple(x.shape)
        if self._shape == key:
            return
        n = x.shape[0]
        v = 1
        for s in x.shape[2:]:
            v *= s
        self.V = v
        self._loss_ws = aligned_empty(self.lib.seg_loss_ws_bytes(n, self.numclass), self.device)
        self._out3 = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._dlogits = torch.empty((n, self.numclass) + tuple(x.shape[2:]), dtype=torch.float32, device=self.device)
        self._shape = key

    def _on_lane(self, i):
        return torch.cuda.stream(self.streams[i]) if self.cuda else _Null()

    def _fork(self):
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            for s in self.streams:
                s.wait_stream(main)

    def _join(self):
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            for s in self.streams:
                main.wait_stream(s)

    # ---- one optimisation step --------------------------------------------------------------------
    def train_step(self, x, target, loss_name="BinaryDiceLoss", lr=1e-3, weight_decay=0.01, decoupled=True, focal_alpha=0.25,
                   focal_gamma=2.0, class_alpha=None, mask_mode=_capi.MASKS_RANDOM, masks=None, allreduce=None, logits=None,
                   probs=None):
        self._prepare(x)
        n = x.shape[0]
        sl = self._slices(n)
        if logits is None:
            logits = torch.empty((n, self.numclass) + tuple(x.shape[2:]), dtype=torch.float32, device=self.device)
        if probs is None:
            probs = torch.empty_like(logits)
        target = target.contiguous()
        # forward of every lane on its own stream
        self._fork()
        for i, (a, b) in enumerate(sl):
            e = self.engines[i]
            e.seed = 0x5EEDC0DE + 7919 * i
            m = None
            if mask_mode == _capi.MASKS_GIVEN:
                m = [mk[a:b] for mk in masks]
            with self._on_lane(i):
                e.forward(x[a:b], mask_mode, m, logits[a:b], probs[a:b])
        self._join()
        # loss + metric on the FULL batch (batch-global sums, like the reference)
        lt = _capi.LABEL_TYPES[str(target.dtype)]
        st = _capi.stream_for(self.device)
        kind = _capi.LOSS_KIND[loss_name]
        self.lib.check(self.lib.seg_loss_forward(_ptr(logits), _ptr(target), lt, n, self.numclass, self.V, kind, float(focal_alpha),
                                                 float(focal_gamma), _ptr(class_alpha), _ptr(self._loss_ws), _ptr(self._out3), st),
                       "seg_loss_forward")
        self.lib.check(self.lib.seg_loss_backward(_ptr(logits), _ptr(target), lt, n, self.numclass, self.V, kind, float(focal_alpha),
                                                  float(focal_gamma), _ptr(self._loss_ws), float(self.loss_scale), _ptr(self._dlogits), st),
                       "seg_loss_backward")
        # backward of every lane
        self._fork()
        for i, (a, b) in enumerate(sl):
            with self._on_lane(i):
                self.engines[i].backward(self._dlogits[a:b], zero_grads=True)
        self._join()
        g = self.engines[0].grads
        for i in range(1, len(sl)):
            g.add_(self.engines[i].grads)
        if allreduce is not None:
            allreduce(g)
        self.engines[0].adam_step(lr=lr, weight_decay=weight_decay, decoupled=decoupled,
                                  grad_div=getattr(allreduce, "world", 1) if allreduce is not None else 1)
        # re-pack the run-dtype weights of every lane from the shared master buffer
        self._fork()
        for i in range(len(sl)):
            with self._on_lane(i):
                self.engines[i].pack_weights()
        self._join()
        self._last_probs = probs
        return self._out3


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

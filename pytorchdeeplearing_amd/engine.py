"""Host-side owner of one libsegengine handle: flat fp32 parameters / gradients / Adam state, the
workspace, and the train-step sequence of the reference loop (model/modelVNet.py:570-596):

    fwd -> loss (+ Dice metric) -> zero_grad -> backward -> [RCCL all-reduce] -> Adam(W) step

PyTorch is used for device memory, streams and torch.distributed only; all arithmetic runs in the
HIP library through the C-ABI (include/segengine.h)."""
import ctypes as C
from collections import OrderedDict

import torch

from . import _capi


import os
_ONE_CALL = os.environ.get("SEG_ONE_CALL", "1") != "0"      # 0: the step as separate library calls (round-2 host path, A/B only)


def aligned_empty(nbytes, device, align=256):
    """uint8 tensor whose data pointer is `align`-byte aligned (the CPU allocator only gives 64)."""
    raw = torch.empty(int(nbytes) + align, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % align
    return raw[off:off + int(nbytes)]


def aligned_zeros_f32(numel, device):
    t = aligned_empty(int(numel) * 4, device).view(torch.float32)
    t.zero_()
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


class SegEngine:
    def __init__(self, kind, ndim, in_channels, numclass, init_features=16, dtype="f16", device="cuda", lib=None):
        self.device = torch.device(device)
        self.lib = lib if lib is not None else _capi.lib_for(self.device)
        self.kind, self.ndim, self.in_channels, self.numclass = kind, ndim, in_channels, numclass
        self.dtype = dtype
        h = C.c_void_p()
        self.lib.check(self.lib.seg_create(_capi.NET_KIND[kind], ndim, in_channels, numclass, init_features,
                                           _capi.DTYPE[dtype], C.byref(h)), "seg_create")
        self.h = h
        self.numel = self.lib.seg_param_numel(h)
        self.table = OrderedDict()
        name = C.create_string_buffer(256)
        shape = (C.c_int * 8)()
        nd = C.c_int()
        off = C.c_longlong()
        for i in range(self.lib.seg_param_count(h)):
            self.lib.check(self.lib.seg_param_info(h, i, name, 256, shape, C.byref(nd), C.byref(off)), "seg_param_info")
            self.table[name.value.decode()] = (tuple(shape[:nd.value]), off.value)
        self.n_drop = self.lib.seg_dropout_calls(h)
        self.drop_ld = self.lib.seg_dropout_ld(h)
        self.drop_channels = [self.lib.seg_dropout_channels(h, i) for i in range(self.n_drop)]
        self.params = aligned_zeros_f32(self.numel, self.device)
        self.grads = aligned_zeros_f32(self.numel, self.device)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.opt_state = None
        self.shape = None
        self.ws = None
        self._loss_ws = None
        self._out3 = None
        self._dlogits = None
        # dropout stream: per-rank under data parallelism (sample n of rank 0 and of rank 1 must not share channel masks).  The rank
        # is resolved at the first engine-drawn forward (not here), so an engine built before init_process_group still gets its own
        # stream; `seed` (constructor-independent base) and the draw counter travel with optimizer_state_dict().
        self.base_seed = 0x5EEDC0DE
        self._seed = None
        self.packed = False
        self._targs = None
        # True: every kernel that reads the target (losses, metrics, clDice) takes (label != 0) - the `y[y != 0] = 1` of the reference's
        # binary training loops (model/modelVNet.py:576) without a host pass; the script-facing binary wrappers switch it on
        self.binarize_labels = False
        # dynamic loss scaling for the f16 run dtype: the fused optimiser skips a step whose gradients overflow and tallies it on
        # the device; every `scale_check_every` steps the host reads the tally (one tiny sync) and halves the scale if anything
        # was skipped, doubles it again after `scale_growth_steps` clean steps (torch.cuda.amp.GradScaler's policy, coarse-grained)
        self.scale_check_every = 64
        self.scale_growth_steps = 2048
        self._steps_since_check = 0
        self._clean_steps = 0
        self.skipped_steps = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.seg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def seed(self):
        if self._seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self._seed = self.base_seed ^ ((rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        return self._seed

    @seed.setter
    def seed(self, v):
        self._seed = int(v) & 0xFFFFFFFFFFFFFFFF

    # ---- parameters -----------------------------------------------------------------------------
    def param_view(self, name, grad=False):
        shape, off = self.table[name]
        n = 1
        for s in shape:
            n *= s
        return (self.grads if grad else self.params)[off:off + n].view(shape)

    def _wait_side(self):
        """the backward-only weight layouts of the last step are packed on the engine's side stream, which nothing orders against the
        caller's stream until the next backward pass: whoever overwrites the parameters before that waits for it here"""
        if self.ws is not None and self.device.type == "cuda" and getattr(self, "_pack_pending", False):
            self.lib.check(self.lib.seg_side_wait(self.h, self.stream()), "seg_side_wait")
            self._pack_pending = False

    def load_state_dict(self, sd):
        self._wait_side()
        with torch.no_grad():
            for k in self.table:
                self.param_view(k).copy_(sd[k].to(device=self.device, dtype=torch.float32))
        self.packed = False

    def state_dict(self):
        return OrderedDict((k, self.param_view(k).clone()) for k in self.table)

    def grad_dict(self, unscale=True):
        inv = 1.0 / self.loss_scale if unscale else 1.0
        return OrderedDict((k, self.param_view(k, grad=True) * inv) for k in self.table)

    @property
    def loss_scale(self):
        return float(self.lib.seg_get_loss_scale(self.h))

    @loss_scale.setter
    def loss_scale(self, v):
        self.lib.check(self.lib.seg_set_loss_scale(self.h, float(v)), "seg_set_loss_scale")

    # ---- planning -------------------------------------------------------------------------------
    def plan(self, n, spatial):
        spatial = tuple(int(s) for s in spatial)
        key = (int(n),) + spatial
        if self.shape == key and self.ws is not None:
            return
        d, hgt, wid = (spatial if self.ndim == 3 else (1,) + spatial)
        self.lib.check(self.lib.seg_plan(self.h, int(n), d, hgt, wid), "seg_plan")
        nbytes = self.lib.seg_workspace_bytes(self.h)
        self.ws = None
        self.ws = aligned_empty(nbytes, self.device)
        self.lib.check(self.lib.seg_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.ws)), "seg_bind")
        self.shape = key
        self.packed = False
        v = 1
        for s in spatial:
            v *= s
        self.V = v
        self._loss_ws = aligned_empty(self.lib.seg_loss_ws_bytes(int(n), self.numclass), self.device)
        self._out3 = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._dlogits = torch.empty((int(n), self.numclass) + spatial, dtype=torch.float32, device=self.device)

    def rebind(self):
        """Re-bind after the caller replaced self.params / self.grads (they must be 256-B aligned)."""
        self.lib.check(self.lib.seg_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.ws)), "seg_bind")
        self.packed = False

    def stream(self):
        return _capi.stream_for(self.device)

    def pack_weights(self):
        self.lib.check(self.lib.seg_pack_weights(self.h, self.stream()), "seg_pack_weights")
        self.packed = True
        self._pack_pending = True          # part of the pack may run on the side stream until the next backward pass joins it

    # ---- forward / backward ---------------------------------------------------------------------
    def forward(self, x, mask_mode=_capi.MASKS_EVAL, masks=None, logits=None, probs=None):
        """x: fp32 NC[D]HW.  Returns (logits, probs) fp32 NC[D]HW."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device.type == self.device.type
        self.plan(x.shape[0], x.shape[2:])
        if not self.packed:
            self.pack_weights()
        oshape = (x.shape[0], self.numclass) + tuple(x.shape[2:])
        if logits is None:
            logits = torch.empty(oshape, dtype=torch.float32, device=self.device)
        if probs is None:
            probs = torch.empty(oshape, dtype=torch.float32, device=self.device)
        mt = None
        if mask_mode == _capi.MASKS_GIVEN:
            mt = self.mask_table(masks)
        self.lib.check(self.lib.seg_forward(self.h, _ptr(x), mask_mode, _ptr(mt), C.c_ulonglong(self.seed),
                                            _ptr(logits), _ptr(probs), self.stream()), "seg_forward")
        self._keep = (x, mt)
        return logits, probs

    def mask_table(self, masks):
        """list of (N, C_i) multipliers in dropout-call order -> dense [calls][N][ld] fp32 table."""
        if torch.is_tensor(masks):
            return masks.to(device=self.device, dtype=torch.float32).contiguous()
        n = masks[0].shape[0]
        t = torch.zeros((self.n_drop, n, self.drop_ld), dtype=torch.float32, device=self.device)
        assert len(masks) == self.n_drop
        for i, m in enumerate(masks):
            assert m.shape[1] == self.drop_channels[i]
            t[i, :, :m.shape[1]] = m.to(self.device)
        return t

    def backward(self, dlogits, zero_grads=True, op_range=None, join=True):
        """dlogits: fp32 NC[D]HW, already multiplied by self.loss_scale.  Accumulates into self.grads.
        op_range = (begin, end): only that slice of the backward op list (bucketed gradient exchange); join=False leaves the slice's
        weight gradients running on the engine's side stream without making the current stream wait (see side_wait)."""
        assert dlogits.dtype == torch.float32 and dlogits.is_contiguous()
        self._pack_pending = False         # the backward pass waits for the side-stream pack itself
        if op_range is None:
            self.lib.check(self.lib.seg_backward(self.h, _ptr(dlogits), 1 if zero_grads else 0, self.stream()), "seg_backward")
        else:
            self.lib.check(self.lib.seg_backward_slice(self.h, _ptr(dlogits), 1 if zero_grads else 0, int(op_range[0]), int(op_range[1]),
                                                       1 if join else 0, self.stream()), "seg_backward_slice")

    def side_wait(self, stream):
        """make `stream` (a torch.cuda.Stream) wait for every weight gradient issued to the engine's side stream so far"""
        self.lib.check(self.lib.seg_side_wait(self.h, stream.cuda_stream), "seg_side_wait")

    def backward_bucket(self, tail_fraction=0.5):
        """(op_split, param_offset, n_ops): after ops [0, op_split) the gradients [param_offset, numel) are final."""
        k, off = C.c_int(0), C.c_longlong(0)
        self.lib.check(self.lib.seg_backward_bucket(self.h, float(tail_fraction), C.byref(k), C.byref(off)), "seg_backward_bucket")
        return k.value, off.value, self.lib.seg_backward_ops(self.h)

    # ---- losses ---------------------------------------------------------------------------------
    def loss_forward(self, logits, target, loss_name, focal_alpha=0.25, focal_gamma=2.0, class_alpha=None, out3=None, exchange=None):
        """exchange: None = the reference loss on this rank's batch; a `parallel.GlobalBatchLoss` = the reference loss of the
        batch over ALL ranks (the batch-global sums of model/losses.py:50-51,259,315-325 are SUM-all-reduced between the
        reduction and the finalize kernel; 32 doubles)."""
        n, c = logits.shape[0], logits.shape[1]
        out3 = self._out3 if out3 is None else out3
        target = target.contiguous()
        if exchange is not None and exchange.world > 1:
            args = (_ptr(logits), _ptr(target), _capi.label_type(target, self.binarize_labels), n, c, self.V, _capi.LOSS_KIND[loss_name],
                    float(focal_alpha), float(focal_gamma))
            self.lib.check(self.lib.seg_loss_reduce(*args, _ptr(self._loss_ws), self.stream()), "seg_loss_reduce")
            shared = self._loss_ws[:8 * self.lib.seg_loss_shared_doubles()].view(torch.float64)
            n_global = exchange(shared, n)
            self.lib.check(self.lib.seg_loss_finalize(*args, _ptr(class_alpha), int(n_global), _ptr(self._loss_ws), _ptr(out3),
                                                      self.stream()), "seg_loss_finalize")
            self._keep_loss = (target, class_alpha)
            return out3
        self.lib.check(self.lib.seg_loss_forward(
            _ptr(logits), _ptr(target), _capi.label_type(target, self.binarize_labels), n, c, self.V, _capi.LOSS_KIND[loss_name],
            float(focal_alpha), float(focal_gamma), _ptr(class_alpha), _ptr(self._loss_ws), _ptr(out3), self.stream()),
            "seg_loss_forward")
        self._keep_loss = (target, class_alpha)
        return out3

    def loss_backward(self, logits, target, loss_name, focal_alpha=0.25, focal_gamma=2.0, dlogits=None, grad_scale=None):
        n, c = logits.shape[0], logits.shape[1]
        dlogits = self._dlogits if dlogits is None else dlogits
        target = target.contiguous()
        gs = self.loss_scale if grad_scale is None else grad_scale
        self.lib.check(self.lib.seg_loss_backward(
            _ptr(logits), _ptr(target), _capi.label_type(target, self.binarize_labels), n, c, self.V, _capi.LOSS_KIND[loss_name],
            float(focal_alpha), float(focal_gamma), _ptr(self._loss_ws), float(gs), _ptr(dlogits), self.stream()),
            "seg_loss_backward")
        return dlogits

    def _cldice_plan(self, n, sp, width):
        d, h, w = (sp if self.ndim == 3 else (1,) + tuple(sp))
        key = (n, d, h, w, width)
        if getattr(self, "_cld_key", None) != key:
            nbytes = self.lib.seg_cldice_ws_bytes(n, d, h, w, self.ndim, width)
            if nbytes < 0:
                raise ValueError("cldice_term: bad extents")
            self._cld_ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
            self._cld_out = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._cld_key = key
            self._cld_target_ready = False
        return d, h, w

    def cldice_prepare_target(self, target, n, sp, width=10):
        """The label-only half of the clDice term (float labels + target skeleton) on a second stream, to be called BEFORE the forward
        pass so the two overlap; cldice_term() then waits for it.  GPU only (the host checker has one stream)."""
        d, h, w = self._cldice_plan(n, tuple(sp), width)
        if self.device.type != "cuda":
            return
        if getattr(self, "_cld_stream", None) is None:
            self._cld_stream = torch.cuda.Stream(device=self.device)
        target = target.contiguous()
        cur = torch.cuda.current_stream(self.device)
        self._cld_stream.wait_stream(cur)                  # labels (and the previous step's readers of the workspace) are done
        self.lib.check(self.lib.seg_cldice_target(_ptr(target), _capi.label_type(target, self.binarize_labels), n, d, h, w, self.ndim, int(width),
                                                  _ptr(self._cld_ws), self._cld_stream.cuda_stream), "seg_cldice_target")
        target.record_stream(self._cld_stream)
        self._cld_target_ready = True

    def cldice_term(self, probs, target, weight=1.0, width=10, dlogits=None, grad_scale=None):
        """Binary soft-clDice (model/lossescldice.py:37-59) on the head's probabilities as one library call (seg_cldice_binary): returns a
        1-element device tensor with the loss; when `dlogits` is given, weight * loss scale * d loss / d logit is ADDED to it (call after
        loss_backward of the companion loss).  The workspace is planned once per shape and kept.
        The target is read as the binary mask (label != 0) whatever `binarize_labels` says - what the reference's train loop hands to every loss
        (model/modelVNet.py:576); soft targets go through `lossescldice.soft_skeletonize` and the plane operators instead (include/segengine.h)."""
        if probs.shape[1] != 1:
            raise ValueError("cldice_term: the binary clDice term needs a one-channel head (numclass == 1)")
        n = probs.shape[0]
        d, h, w = self._cldice_plan(n, tuple(probs.shape[2:]), width)
        target = target.contiguous()
        ready = 1 if getattr(self, "_cld_target_ready", False) else 0
        if ready:
            torch.cuda.current_stream(self.device).wait_stream(self._cld_stream)
            self._cld_target_ready = False
        gs = (self.loss_scale if grad_scale is None else grad_scale) * float(weight)
        self.lib.check(self.lib.seg_cldice_binary(
            _ptr(probs), _ptr(target), _capi.label_type(target, self.binarize_labels), n, d, h, w, self.ndim, int(width), float(gs),
            _ptr(self._cld_ws), _ptr(self._cld_out), _ptr(dlogits) if dlogits is not None else None, ready, self.stream()), "seg_cldice_binary")
        return self._cld_out

    # ---- optimiser ------------------------------------------------------------------------------
    def init_optimizer(self):
        self.exp_avg = aligned_zeros_f32(self.numel, self.device)
        self.exp_avg_sq = aligned_zeros_f32(self.numel, self.device)
        self.opt_state = torch.zeros(64, dtype=torch.int32, device=self.device)

    def optimizer_state_dict(self):
        """Everything a resumed run needs besides state_dict(): Adam moments and step, loss scale, dropout stream position."""
        if self.exp_avg is None:
            self.init_optimizer()
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": int(self.opt_state[0]),
                "loss_scale": self.loss_scale, "dropout_seed": self.seed, "dropout_draws": int(self.lib.seg_dropout_draws(self.h))}

    def load_optimizer_state_dict(self, sd):
        if self.exp_avg is None:
            self.init_optimizer()
        with torch.no_grad():
            self.exp_avg.copy_(sd["exp_avg"].to(self.device))
            self.exp_avg_sq.copy_(sd["exp_avg_sq"].to(self.device))
            self.opt_state.zero_()
            self.opt_state[0] = int(sd["step"])
        self.loss_scale = float(sd["loss_scale"])
        self.seed = int(sd["dropout_seed"])
        self.lib.check(self.lib.seg_set_dropout_draws(self.h, int(sd["dropout_draws"])), "seg_set_dropout_draws")

    def adam_step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decoupled=True, check_finite=None, grad_div=1.0):
        """grad_div: extra divisor of the gradient (world size after a SUM all-reduce), folded into the loss-scale factor."""
        if self.exp_avg is None:
            self.init_optimizer()
        if check_finite is None:
            check_finite = self.dtype in ("f16", "fp16", "float16")
        self._wait_side()
        self.lib.check(self.lib.seg_adam_step(
            _ptr(self.params), _ptr(self.grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), self.numel,
            float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), 1 if decoupled else 0,
            1.0 / (self.loss_scale * float(grad_div)), 1 if check_finite else 0, _ptr(self.opt_state), self.stream()), "seg_adam_step")
        self.packed = False

    # ---- measurement ------------------------------------------------------------------------------
    def profile_enable(self, classes):
        """bracket every launch of the named kernel classes with HIP events (empty list: off)."""
        mask = 0
        for c in classes:
            mask |= 1 << _capi.KERNEL_CLASSES.index(c)
        self.lib.check(self.lib.seg_profile_enable(self.h, mask), "seg_profile_enable")

    def profile_read(self):
        n = len(_capi.KERNEL_CLASSES)
        calls, ms = (C.c_int * n)(), (C.c_float * n)()
        nbytes, flops = (C.c_double * n)(), (C.c_double * n)()
        self.lib.check(self.lib.seg_profile_read(self.h, calls, ms, nbytes, flops), "seg_profile_read")
        return {k: dict(calls=calls[i], ms=ms[i], bytes=nbytes[i], flops=flops[i])
                for i, k in enumerate(_capi.KERNEL_CLASSES) if calls[i]}

    # ---- one optimisation step of the reference loop --------------------------------------------
    def train_step(self, x, target, loss_name="BinaryDiceLoss", lr=1e-3, weight_decay=0.01, decoupled=True,
                   focal_alpha=0.25, focal_gamma=2.0, class_alpha=None, mask_mode=_capi.MASKS_RANDOM, masks=None,
                   allreduce=None, logits=None, probs=None, loss_exchange=None, cldice_weight=0.0, cldice_width=10, launch="stream"):
        """Returns out3 = device tensor [loss, dice metric, iou metric] (no host sync).
        cldice_weight > 0 (binary heads): loss = loss_name + cldice_weight * soft-clDice(probabilities, target) — BASELINE configs[4];
        out3[0] then holds the sum (the clDice part alone is `self.last_cldice`).
        loss_exchange (parallel.GlobalBatchLoss): exact loss of the global batch over all ranks; the parameter gradients
        of the ranks are then summed, not averaged (the 1/world factor is dropped).
        launch (single-rank path): "stream" = the step's launches are enqueued one by one (one library call); "graph" = the step is captured
        once per (buffers, hyper-parameters) as a HIP graph and replayed - for hosts that cannot enqueue ~250 launches per step as fast as
        the GPU runs them (the tensors must then stay at the same addresses from step to step); falls back to "stream" where a capture is
        not possible."""
        world = 1
        if allreduce is not None:
            world = getattr(allreduce, "world", None)
            if world is None:
                # an exchange object without a world size would be dropped silently (world = 1) and the ranks would train unsynchronised; the size of the
                # default process group is taken for it (kept in a local: the callable may be a bound method, a partial, a builtin - nothing is set on it)
                import torch.distributed as dist
                if not (dist.is_available() and dist.is_initialized()):
                    raise TypeError("train_step: `allreduce` has no `.world` attribute and no torch.distributed process group is initialised; "
                                    "pass parallel.GradAllReduce / BucketedGradAllReduce or give the callable a `.world`")
                world = dist.get_world_size()
        xworld = loss_exchange.world if loss_exchange is not None else 1
        if cldice_weight and xworld > 1:
            # the clDice term is a rank-local ratio: under GlobalBatchLoss the summed (not averaged) gradients would carry it `world` times
            raise NotImplementedError("train_step: cldice_weight together with a global-batch loss exchange is not supported; "
                                      "use the per-rank (DDP) loss semantics for the clDice term")
        if not cldice_weight and _ONE_CALL:
            # ONE library call per step for every world size: with ranks to exchange with, seg_train_step calls back at the points of its
            # schedule where a collective belongs (gradient buckets, the loss sums) and this side enqueues it (torch.distributed)
            return self._train_step_one_call(x, target, loss_name, lr, weight_decay, decoupled, focal_alpha, focal_gamma, class_alpha,
                                             mask_mode, masks, logits, probs, launch,
                                             allreduce if (world > 1 or getattr(allreduce, "native", False)) else None,
                                             loss_exchange if xworld > 1 else None)
        if cldice_weight:
            self.cldice_prepare_target(target, x.shape[0], tuple(x.shape[2:]), cldice_width)     # overlaps the forward pass
        logits, probs = self.forward(x, mask_mode, masks, logits, probs)
        self._last_probs = probs
        out3 = self.loss_forward(logits, target, loss_name, focal_alpha, focal_gamma, class_alpha, exchange=loss_exchange)
        dl = self.loss_backward(logits, target, loss_name, focal_alpha, focal_gamma)
        if cldice_weight:
            self.last_cldice = self.cldice_term(probs, target, cldice_weight, cldice_width, dlogits=dl)
            out3[0:1].add_(self.last_cldice, alpha=float(cldice_weight))
        grad_div = 1 if xworld > 1 else world
        if allreduce is not None and getattr(allreduce, "bucketed", False) and world > 1:
            # buckets: every finished suffix of the flat gradient buffer is exchanged while the finer levels still run.  On the GPU the
            # collective is ordered after an auxiliary stream that waits for the main stream AND the weight-gradient stream, so the
            # backward pass itself never stalls at a bucket boundary (a joined boundary cost 0.14 ms of a 4.5 ms step, tools/bench_bucket_stall.py)
            fractions = getattr(allreduce, "fractions", None) or (allreduce.tail_fraction,)
            nops = self.lib.seg_backward_ops(self.h)
            on_gpu = self.device.type == "cuda"
            if on_gpu and getattr(self, "_ar_stream", None) is None:
                self._ar_stream = torch.cuda.Stream(device=self.device)
            prev_k, prev_off, works = 0, self.numel, []
            for f in fractions:
                k, off, _ = self.backward_bucket(f)
                if k <= prev_k or k >= nops or off >= prev_off:
                    continue
                self.backward(dl, zero_grads=(prev_k == 0), op_range=(prev_k, k), join=not on_gpu)
                if on_gpu:
                    aux = self._ar_stream
                    aux.wait_stream(torch.cuda.current_stream(self.device))
                    self.side_wait(aux)
                    with torch.cuda.stream(aux):
                        works.append(allreduce.start(self.grads[off:prev_off]))
                else:
                    works.append(allreduce.start(self.grads[off:prev_off]))
                prev_k, prev_off = k, off
            self.backward(dl, zero_grads=(prev_k == 0), op_range=(prev_k, nops))
            if on_gpu:
                torch.cuda.current_stream(self.device).wait_stream(self._ar_stream)      # whatever `start` queued on the auxiliary stream itself
            works.append(allreduce.start(self.grads[:prev_off]))
            allreduce.finish(works)
        else:
            self.backward(dl, zero_grads=True)
            if allreduce is not None:
                allreduce(self.grads)
        self.adam_step(lr=lr, weight_decay=weight_decay, decoupled=decoupled, grad_div=grad_div)
        self.pack_weights()
        self._after_step()
        return out3

    def _after_step(self):
        self._steps_since_check += 1
        if self.dtype in ("f16", "fp16", "float16") and self._steps_since_check >= self.scale_check_every:
            self.update_loss_scale(blocking=False)

    def _exchange_hooks(self):
        """ctypes callbacks handed to seg_train_step (created once; they act on the exchange objects of the current step, self._xchg)."""
        if getattr(self, "_hooks", None) is None:
            def bucket(_user, idx, off, cnt):
                try:
                    st = self._xchg
                    ar = st["allreduce"]
                    if idx < 0:                                   # every exchange has been enqueued: the caller's stream waits for them
                        if getattr(ar, "bucketed", False):
                            ar.finish(st["works"])
                        return 0
                    sl = self.grads[off:off + cnt]
                    if not getattr(ar, "bucketed", False):
                        ar(sl)                                    # one blocking / stream-ordered collective after the backward pass
                    elif off > 0 and st["aux"] is not None:       # an early bucket: behind the auxiliary stream (ordered by the library)
                        with torch.cuda.stream(st["aux"]):
                            st["works"].append(ar.start(sl))
                    else:
                        st["works"].append(ar.start(sl))
                    return 0
                except BaseException as ex:                       # never let an exception unwind through the C frames: the library returns at
                    self._xchg["error"] = ex                      # once on a non-zero code and the step re-raises it (KeyboardInterrupt included)
                    return 1

            def loss(_user, sums_ptr, n_doubles):
                try:
                    st = self._xchg
                    if int(sums_ptr or 0) != self._loss_ws.data_ptr():
                        raise RuntimeError("seg_train_step handed the loss hook sums at 0x%x, not the loss workspace (0x%x)" % (int(sums_ptr or 0), self._loss_ws.data_ptr()))
                    shared = self._loss_ws[:8 * n_doubles].view(torch.float64)
                    return int(st["loss_exchange"](shared, st["n_local"]))
                except BaseException as ex:
                    self._xchg["error"] = ex
                    return -1
            self._hooks = (_capi.BUCKET_CB(bucket), _capi.LOSS_CB(loss))
        return self._hooks

    def _train_step_one_call(self, x, target, loss_name, lr, weight_decay, decoupled, focal_alpha, focal_gamma, class_alpha,
                             mask_mode, masks, logits, probs, launch="stream", allreduce=None, loss_exchange=None):
        """The rank-local step as ONE library call (seg_train_step): the argument block is filled once per (shape, buffers) and only
        the pointers that change are rewritten, so the host side of a step is one FFI crossing."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device.type == self.device.type
        self.plan(x.shape[0], x.shape[2:])
        if self.exp_avg is None:
            self.init_optimizer()
        oshape = (x.shape[0], self.numclass) + tuple(x.shape[2:])
        if logits is None:
            logits = torch.empty(oshape, dtype=torch.float32, device=self.device)
        if probs is None:
            probs = torch.empty(oshape, dtype=torch.float32, device=self.device)
        target = target.contiguous()
        mt = self.mask_table(masks) if mask_mode == _capi.MASKS_GIVEN else None
        a = self._targs
        if a is None:
            a = self._targs = _capi.TrainArgs()
            a.beta1, a.beta2, a.eps, a.grad_div = 0.9, 0.999, 1e-8, 1.0
        a.x, a.target, a.label_type = x.data_ptr(), target.data_ptr(), _capi.label_type(target, self.binarize_labels)
        a.loss_kind, a.focal_alpha, a.focal_gamma = _capi.LOSS_KIND[loss_name], float(focal_alpha), float(focal_gamma)
        a.class_alpha = class_alpha.data_ptr() if class_alpha is not None else None
        a.logits, a.probs, a.dlogits = logits.data_ptr(), probs.data_ptr(), self._dlogits.data_ptr()
        a.loss_ws, a.out3 = self._loss_ws.data_ptr(), self._out3.data_ptr()
        a.mask_mode, a.masks, a.seed = int(mask_mode), (mt.data_ptr() if mt is not None else None), self.seed
        a.exp_avg, a.exp_avg_sq, a.opt_state = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.opt_state.data_ptr()
        a.lr, a.weight_decay, a.decoupled = float(lr), float(weight_decay), 1 if decoupled else 0
        a.check_finite = 1 if self.dtype in ("f16", "fp16", "float16") else 0
        a.packed = 1 if self.packed else 0
        native = allreduce is not None and getattr(allreduce, "native", False) and loss_exchange is None
        want = allreduce if native else None
        if getattr(self, "_rccl_on", None) is not want:
            # the in-library exchange is a property of the handle: set when a step runs with a NativeRcclAllReduce, removed when one runs without it
            comm, fn = want.handles(self.device) if want is not None else (None, None)
            self.lib.check(self.lib.seg_set_rccl_comm(self.h, comm, fn), "seg_set_rccl_comm")
            self._rccl_on = want
        hooked = (allreduce is not None and not native) or loss_exchange is not None
        if native:
            fr = getattr(allreduce, "fractions", None) or (allreduce.tail_fraction,)
            a.bucket_cb, a.loss_cb, a.aux_stream = None, None, None
            a.nfrac = min(len(fr), 4)
            for i in range(4):
                a.fractions[i] = float(fr[i]) if i < a.nfrac else 0.0
            a.grad_div = float(getattr(allreduce, "world", 1))
            self.lib.check(self.lib.seg_train_step(self.h, C.byref(a), self.stream()), "seg_train_step")
            self.packed = True
            self._pack_pending = True
            self._keep = (x, mt)
            self._keep_loss = (target, class_alpha)
            self._last_probs = probs
            self._after_step()
            return self._out3
        if hooked:
            bucket_cb, loss_cb = self._exchange_hooks()
            on_gpu = self.device.type == "cuda"
            bucketed = allreduce is not None and getattr(allreduce, "bucketed", False)
            if on_gpu and bucketed and getattr(self, "_ar_stream", None) is None:
                self._ar_stream = torch.cuda.Stream(device=self.device)
            aux = self._ar_stream if (on_gpu and bucketed) else None
            self._xchg = {"allreduce": allreduce, "loss_exchange": loss_exchange, "works": [], "aux": aux, "n_local": int(x.shape[0]), "error": None}
            a.bucket_cb = C.cast(bucket_cb, C.c_void_p) if allreduce is not None else None
            a.loss_cb = C.cast(loss_cb, C.c_void_p) if loss_exchange is not None else None
            fr = (getattr(allreduce, "fractions", None) or (allreduce.tail_fraction,)) if bucketed else ()
            a.nfrac = min(len(fr), 4)
            for i in range(4):
                a.fractions[i] = float(fr[i]) if i < a.nfrac else 0.0
            a.aux_stream = aux.cuda_stream if aux is not None else None
            # a global-batch loss sums the ranks' gradients (each already carries the global 1/count); plain data parallelism averages them
            a.grad_div = 1.0 if loss_exchange is not None else float(getattr(allreduce, "world", 1))
        else:
            a.bucket_cb, a.loss_cb, a.nfrac, a.aux_stream, a.grad_div = None, None, 0, None, 1.0
        if hooked:
            rc = self.lib.seg_train_step(self.h, C.byref(a), self.stream())
            err, self._xchg["error"] = self._xchg["error"], None
            if err is not None:
                raise err
            self.lib.check(rc, "seg_train_step")
            self.packed = True
            self._pack_pending = True
            self._keep = (x, mt)
            self._keep_loss = (target, class_alpha)
            self._last_probs = probs
            self._after_step()
            return self._out3
        if (launch == "graph" and self.packed and self.device.type == "cuda" and mask_mode != _capi.MASKS_GIVEN and
                not getattr(self, "_graph_refused", False)):
            # the argument block as bytes is the identity of the captured step (pointers, scalars; the loss scale is tracked by the library)
            key = bytes(a)
            # the legacy default stream cannot be captured: graph steps run on a stream of their own, ordered after / before the caller's
            cur = torch.cuda.current_stream(self.device)
            if getattr(self, "_graph_stream", None) is None:
                self._graph_stream = torch.cuda.Stream(device=self.device)
            gs = self._graph_stream if cur.cuda_stream == 0 else cur
            gsp = C.c_void_p(gs.cuda_stream)
            if gs is not cur:
                gs.wait_stream(cur)
            if getattr(self, "_graph_key", None) != key or not self.lib.seg_train_graph_ready(self.h):
                # every scalar of the step (lr included) is baked into the captured launches: a schedule that changes one of them each step would
                # re-capture each step (stream syncs + ~250 captured launches + instantiate - slower than the stream launches it replaces)
                self._graph_recaptures = getattr(self, "_graph_recaptures", 0) + (1 if getattr(self, "_graph_key", None) is not None else 0)
                if self._graph_recaptures >= 3:                   # three steps in a row (a replay in between resets the count)
                    import warnings
                    warnings.warn("segengine: launch='graph' re-captured the step on 3 consecutive steps (lr / buffers change from step to step); falling back to stream launches")
                    self._graph_key, self._graph_refused = None, True
                elif self.lib.seg_train_graph_capture(self.h, C.byref(a), gsp) == 0:
                    self._graph_key = key
                else:                         # not retried step after step: a failed capture costs milliseconds
                    self._graph_key, self._graph_refused = None, True
                    self.graph_error = self.lib.seg_last_error().decode()
            else:
                self._graph_recaptures = 0
            if self._graph_key is not None:
                self.lib.check(self.lib.seg_train_graph_launch(self.h, gsp), "seg_train_graph_launch")
                if gs is not cur:
                    cur.wait_stream(gs)
                self._pack_pending = False          # the captured step packs on one stream
                self._keep = (x, mt)
                self._keep_loss = (target, class_alpha)
                self._last_probs = probs
                self._after_step()
                return self._out3
        self.lib.check(self.lib.seg_train_step(self.h, C.byref(a), self.stream()), "seg_train_step")
        self.packed = True
        self._pack_pending = True
        self._keep = (x, mt)
        self._keep_loss = (target, class_alpha)
        self._last_probs = probs
        self._after_step()
        return self._out3

    def update_loss_scale(self, blocking=True):
        """Read the device-side tally of skipped (overflowed) optimiser steps and adapt the loss scale; returns the number of
        skipped steps applied by this call.  A run whose gradients overflow persistently is reported instead of silently
        training nothing.  blocking=False (what train_step uses): the tally is snapshotted to pinned host memory asynchronously
        (copy, then clear, in stream order) and applied at the NEXT check, so the step never waits for the device; the policy then
        lags by one interval."""
        n = self._steps_since_check
        self._steps_since_check = 0
        if self.opt_state is None:
            return 0
        pend = getattr(self, "_tally_pending", None)
        if blocking or self.device.type != "cuda":
            skipped = int(self.opt_state[2].item())
            if skipped:
                self.opt_state[2:3].zero_()
            if pend is not None:
                pend[1].synchronize()
                skipped += int(pend[0][0])
                n += pend[2]
                self._tally_pending = None
            return self._apply_tally(skipped, n)
        applied = 0
        if pend is not None:
            if not pend[1].query():              # the previous snapshot has not landed yet: look again at the next step
                self._steps_since_check = n
                return 0
            applied = self._apply_tally(int(pend[0][0]), pend[2])
        if getattr(self, "_tally_host", None) is None:
            self._tally_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._tally_host.copy_(self.opt_state[2:3], non_blocking=True)
        self.opt_state[2:3].zero_()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._tally_pending = (self._tally_host, ev, n)
        return applied

    def _apply_tally(self, skipped, n):
        if skipped:
            self.skipped_steps += skipped
            self._clean_steps = 0
            new = max(self.loss_scale * 0.5, 1.0)
            import warnings
            warnings.warn("segengine: %d of the last %d optimiser steps overflowed in f16 and were skipped; loss scale %g -> %g"
                          % (skipped, n, self.loss_scale, new))
            self.loss_scale = new
        else:
            self._clean_steps += n
            if self._clean_steps >= self.scale_growth_steps and self.loss_scale < 65536.0:
                self.loss_scale = self.loss_scale * 2.0
                self._clean_steps = 0
        return skipped

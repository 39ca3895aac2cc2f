"""Data parallelism over the batch axis: one process per GPU, full weight replica per rank, ONE
all-reduce of the flat fp32 gradient buffer per step (38 MB for VNet3d) over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  GroupNorm statistics and
channel-dropout masks are per sample, so forward/backward need no collective.

Semantics = what wrapping the reference in DistributedDataParallel would do: every rank computes
the reference loss on ITS shard (the Dice/CE losses are batch-global ratios, model/losses.py:50-51),
gradients are averaged over ranks."""
import torch
import torch.distributed as dist


class GradAllReduce:
    """One blocking SUM all-reduce of the whole flat gradient buffer after the backward pass; the 1/world average is
    folded into the fused optimiser step (`SegEngine.adam_step(grad_div=world)`), not a separate pass over the buffer."""
    bucketed = False

    def __init__(self, world_size=None, group=None):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)

    def __call__(self, flat_grads: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grads


class BucketedGradAllReduce(GradAllReduce):
    """Buckets overlapped with the backward pass (two in round 1, up to four now).  Gradients finish in reverse registration order, so after the
    deepest level's backward ops a SUFFIX of the flat buffer (>= `tail_fraction` of the elements: the 256-/128-channel
    blocks and the whole decoder, ~80 % of VNet3d's 38 MB) is final: its all-reduce is issued asynchronously on the
    process group's own stream while the fine-level backward ops (the longest kernels of the step) still run; the head
    of the buffer follows at the end.  xGMI is point-to-point, so two large ring collectives (not many small buckets)
    keep every link busy.  `SegEngine.train_step` drives it through seg_backward_bucket / seg_backward_range."""
    bucketed = True

    def __init__(self, world_size=None, group=None, tail_fraction=0.5, fractions=None):
        super().__init__(world_size, group)
        self.tail_fraction = tail_fraction
        # bucket boundaries as finished fractions of the flat buffer.  Default: the deepest level + decoder (>= 50 %, in fact ~80 % of
        # VNet3d), then the next encoder levels as they finish (97 %, 99.5 %), so that only a few hundred KB are left for the collective
        # that nothing can hide any more (the one after the last backward op).  Boundaries that coincide are skipped.
        self.fractions = tuple(fractions) if fractions is not None else (tail_fraction, 0.97, 0.995)

    def start(self, flat_slice: torch.Tensor):
        """asynchronous SUM all-reduce of one bucket (ordered after the work already queued on the current stream)."""
        if self.world == 1 or flat_slice.numel() == 0:
            return None
        return dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def finish(works):
        for w in works:
            if w is not None:
                w.wait()                      # current stream waits for the collective; no host block on GPU backends


class NativeRcclAllReduce(BucketedGradAllReduce):
    """The same buckets with NO Python between the slices of the backward pass: the process group's RCCL communicator and the
    `ncclAllReduce` of the RCCL library torch loaded are handed to the engine once (seg_set_rccl_comm); seg_train_step then issues the in-place
    SUM all-reduces itself, on its own exchange stream, where `BucketedGradAllReduce` is called back into `torch.distributed`.  GPU ranks with the
    "nccl" backend only; `comm` / `allreduce_fn` may also be given explicitly (a caller that owns its communicator, or a test double)."""
    native = True

    def __init__(self, world_size=None, group=None, tail_fraction=0.5, fractions=None, comm=None, allreduce_fn=None):
        super().__init__(world_size, group, tail_fraction, fractions)
        self.comm, self.allreduce_fn = comm, allreduce_fn
        self._keep = None

    def handles(self, device):
        """(ncclComm_t, address of ncclAllReduce) as integers"""
        if self.comm is None:
            pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
            backend = pg._get_backend(torch.device(device))
            if not hasattr(backend, "_comm_ptr"):
                raise RuntimeError("NativeRcclAllReduce needs the 'nccl' (RCCL) backend; this process group is %r" % (dist.get_backend(pg),))
            # a communicator is created lazily by the first collective on the device
            t = torch.zeros(1, device=device)
            dist.all_reduce(t, group=self.group)
            torch.cuda.synchronize(device)
            self.comm = int(backend._comm_ptr())
        if self.allreduce_fn is None:
            import ctypes
            import os
            lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))      # the library torch itself loaded (same handle)
            self._keep = lib
            self.allreduce_fn = ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p).value
        fn = self.allreduce_fn
        if not isinstance(fn, int):                    # a ctypes callback object (tests): keep it alive, pass its address
            import ctypes
            self._keep = fn
            fn = ctypes.cast(fn, ctypes.c_void_p).value
        return int(self.comm), int(fn)


class GlobalBatchLoss:
    """Exact global-batch loss across ranks (SURVEY.md section 8e mode ii): the reference losses are ratios of sums over
    the WHOLE batch (model/losses.py:50-51, 259, 315-325), so the loss of N volumes split over R ranks needs the sums of
    all ranks before the ratio is taken.  Called by `SegEngine.loss_forward` between the reduction kernel and the
    finalize kernel with the rank's batch-global sums (32 fp64 values, on the device); returns the global sample count.
    One tiny SUM all-reduce; the parameter gradients are then summed over ranks instead of averaged."""

    def __init__(self, world_size=None, group=None, equal_shards=False):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        # False (default): the global sample count is taken from the exchanged sums ON THE DEVICE (seg_loss_reduce leaves the local sample
        # count in the shared doubles, the all-reduce sums it, seg_loss_finalize(n_global = 0) reads it) - right for a partial last batch
        # spread unevenly over the ranks, and free (no allocation, no read-back).  True: opt-in shortcut n_local * world for equal shards
        self.equal_shards = equal_shards

    def __call__(self, shared_sums: torch.Tensor, n_local: int):
        if self.world > 1:
            dist.all_reduce(shared_sums, op=dist.ReduceOp.SUM, group=self.group)
            return n_local * self.world if self.equal_shards else 0
        return n_local


def broadcast_parameters(engine, src=0, group=None):
    """make every replica start from rank `src`'s weights (call once after init / load)."""
    dist.broadcast(engine.params, src=src, group=group)
    engine.packed = False

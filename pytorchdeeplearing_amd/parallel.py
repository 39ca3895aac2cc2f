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
    def __init__(self, world_size=None, group=None):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)

    def __call__(self, flat_grads: torch.Tensor):
        if self.world == 1:
            return flat_grads
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        flat_grads.mul_(1.0 / self.world)
        return flat_grads


def broadcast_parameters(engine, src=0, group=None):
    """make every replica start from rank `src`'s weights (call once after init / load)."""
    dist.broadcast(engine.params, src=src, group=group)
    engine.packed = False

"""N > 1 path on CPU: world_size-2 gloo, each rank drives its own engine (host-checker build) on its
batch shard; the averaged-gradient update must equal the oracle emulation of R independent shards."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, bucketed):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import conftest
    conftest.emu_library()
    from oracle import seg_oracle as seg
    from pytorchdeeplearing_amd import SegEngine, _capi
    from pytorchdeeplearing_amd.parallel import BucketedGradAllReduce, GradAllReduce, broadcast_parameters
    kind, shape, ncls, loss = "unet", (2, 1, 16, 16), 1, "BinaryDiceLoss"
    e = SegEngine(kind, 2, 1, ncls, dtype="f32", device="cpu")
    # rank 0 owns the initial weights; the others start from garbage and must receive them
    params = seg.perturb_params(seg.init_params(kind, 2, 1, ncls, seed=0), seed=7)
    if rank == 0:
        e.load_state_dict(params)
    else:
        e.params.fill_(123.0)
    broadcast_parameters(e)
    x, y = seg.synthetic_batch(shape[0], shape[2:], 1, ncls, seed=100 + rank)      # this rank's shard
    ar = BucketedGradAllReduce() if bucketed else GradAllReduce()
    for it in range(2):
        g = torch.Generator().manual_seed(10 * it + rank)
        masks = seg.draw_masks(kind, shape[0], generator=g)
        e.train_step(x, y, loss, lr=1e-3, mask_mode=_capi.MASKS_GIVEN, masks=masks, allreduce=ar)
    if bucketed:
        k, off, nops = e.backward_bucket(ar.tail_fraction)
        assert 0 < k < nops and 0 < off < e.numel and (e.numel - off) >= 0.5 * e.numel
    if rank == 0:
        q.put({k: v.numpy().copy() for k, v in e.state_dict().items()})      # by value: this process exits before the parent reads
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucketed", [False, True])
def test_ddp_two_ranks_matches_oracle_emulation(bucketed):
    from oracle import seg_oracle as seg
    world, port = 2, 29500 + (os.getpid() * 2 + int(bucketed)) % 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, bucketed)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # oracle: R independent shards, gradients averaged on the host, one AdamW state
    kind, shape, ncls, loss = "unet", (2, 1, 16, 16), 1, "BinaryDiceLoss"
    cur = seg.perturb_params(seg.init_params(kind, 2, 1, ncls, seed=0), seed=7)
    st = {}
    for it in range(2):
        grads = None
        for rank in range(world):
            x, y = seg.synthetic_batch(shape[0], shape[2:], 1, ncls, seed=100 + rank)
            g = torch.Generator().manual_seed(10 * it + rank)
            masks = seg.draw_masks(kind, shape[0], generator=g)
            r = seg.forward_backward(kind, cur, x, y, loss, masks=masks)
            grads = r["grads"] if grads is None else {k: grads[k] + r["grads"][k] for k in grads}
        grads = {k: v / world for k, v in grads.items()}
        cur = seg.adamw_step(cur, grads, st)
    tot = bad = 0
    for k in cur:
        d = (torch.from_numpy(got[k]) - cur[k]).abs()
        assert float(d.max()) < 2 * 2e-3, k          # Adam moves every weight by <= lr per step
        tot += d.numel()
        bad += int((d > 1e-4).sum())
    assert bad <= 0.01 * tot, (bad, tot)

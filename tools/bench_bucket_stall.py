"""What the bucketed exchange costs on ONE GPU apart from the collective itself: the VNet3d 4x96^3 f16 train step driven through
SegEngine.train_step's N > 1 sequencing (backward slice -> exchange the finished suffix -> rest -> exchange the head) with a loop-back
exchange object that moves no data, against the plain step.  The difference is the price of the mid-backward join(s) of the main stream with the
weight-gradient stream."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import SegEngine, synthetic as seg
from pytorchdeeplearing_amd.parallel import BucketedGradAllReduce


class Loopback(BucketedGradAllReduce):
    def __init__(self, **kw):
        super().__init__(world_size=2, group=None, **kw)     # world 2: train_step takes the bucketed path; nothing is exchanged

    def start(self, flat_slice):
        return None


dev = torch.device("cuda")
x, y = seg.synthetic_batch(4, (96, 96, 96), 1, 1, seed=1)
x, y = x.to(dev), y.to(dev)
variants = {"plain": None, "two_buckets": Loopback()}
if hasattr(BucketedGradAllReduce, "fractions") or "fractions" in BucketedGradAllReduce.__init__.__code__.co_varnames:
    variants["four_buckets"] = Loopback(fractions=(0.5, 0.97, 0.995))
for name, ar in variants.items():
    e = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
    seg.init_engine(e, seed=0)
    for _ in range(5):
        e.train_step(x, y, "BinaryDiceLoss", allreduce=ar)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        e.train_step(x, y, "BinaryDiceLoss", allreduce=ar)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
    print(json.dumps({"variant": name, "ms_per_step": round(ms, 3)}), flush=True)
    del e

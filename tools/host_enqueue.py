"""How long does the HOST need to enqueue one train step (no GPU wait)?  If this is close to the step time the step is
launch-bound wherever kernels are shorter than a launch (the 12^3 / 6^3 levels)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import synthetic as seg
from pytorchdeeplearing_amd import SegEngine, _capi
dev = torch.device("cuda:0")
e = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
seg.init_engine(e, seed=0)
x, y = seg.synthetic_batch(4, (96, 96, 96), 1, 1, seed=1234)
x, y = x.to(dev), y.to(dev)
for _ in range(5):
    e.train_step(x, y, "BinaryDiceLoss")
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.train_step(x, y, "BinaryDiceLoss")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
# phases
def phase(fn, n=10):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    return round(min(ts), 3)
logits, probs = e.forward(x, _capi.MASKS_RANDOM)
res = {"host_enqueue_ms_per_step": round(min(host), 3), "median": round(sorted(host)[5], 3), "step_ms_from_idle": round(min(total), 3),
       "forward_enqueue_ms": phase(lambda: e.forward(x, _capi.MASKS_RANDOM)),
       "loss_enqueue_ms": phase(lambda: (e.loss_forward(logits, y, "BinaryDiceLoss"), e.loss_backward(logits, y, "BinaryDiceLoss"))),
       "backward_enqueue_ms": phase(lambda: e.backward(e._dlogits)),
       "adam_pack_enqueue_ms": phase(lambda: (e.adam_step(), e.pack_weights()))}
print(json.dumps(res))

"""Experiment: whole train step captured in a HIP graph (torch.cuda.CUDAGraph around the C-ABI launches)."""
import os, sys, time
# a captured step must end with every forked stream joined: the backward-only weight layouts of the NEXT step are packed on the side stream at the
# end of train_step and only joined by that step's backward pass, so the split is switched off for the capture
os.environ.setdefault("SEG_PACK_SPLIT", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import synthetic as seg
from pytorchdeeplearing_amd import SegEngine

dev = torch.device("cuda")
e = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
seg.init_engine(e, seed=0)
x, y = seg.synthetic_batch(4, (96, 96, 96), 1, 1, seed=1234)
x, y = x.to(dev), y.to(dev)
logits = torch.empty((4, 1, 96, 96, 96), dtype=torch.float32, device=dev)
probs = torch.empty_like(logits)
step = lambda: e.train_step(x, y, "BinaryDiceLoss", lr=1e-3, logits=logits, probs=probs)
for _ in range(5):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = step()
torch.cuda.synchronize()
print("stream launches: %.3f ms/step, loss %.5f" % ((time.perf_counter() - t0) / 20 * 1e3, float(out[0])))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph replay:    %.3f ms/step, loss %.5f" % ((time.perf_counter() - t0) / 20 * 1e3, float(out[0])))

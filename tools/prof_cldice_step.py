"""C5 + soft-clDice train steps for a profiler run (rocprofv3 --kernel-trace --stats -- python tools/prof_cldice_step.py): VNet3d 1 x 160^3 bf16,
BinaryDiceLoss + train_step(cldice_weight=1) - BASELINE configs[4] as worded.  Prints ms per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import SegEngine, synthetic as seg
dev = torch.device("cuda")
S = int(os.environ.get("SEG_CLD_SIZE", "160"))
e = SegEngine("vnet", 3, 1, 1, dtype="bf16", device=dev)
seg.init_engine(e, seed=0)
x, y = seg.synthetic_batch(1, (S, S, S), 1, 1, seed=1)
x, y = x.to(dev), y.to(dev)
for w in (0.0, 1.0):
    for _ in range(3):
        e.train_step(x, y, "BinaryDiceLoss", cldice_weight=w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get("SEG_CLD_STEPS", "6"))
    for _ in range(n):
        e.train_step(x, y, "BinaryDiceLoss", cldice_weight=w)
    torch.cuda.synchronize()
    print("cldice_weight %.0f: %.3f ms per step" % (w, (time.perf_counter() - t0) / n * 1e3), flush=True)

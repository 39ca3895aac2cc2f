"""Pre/post-processing kernels around predict on one GPU: time per call and the fraction of the
HBM roofline from algorithmic bytes (read source once, write result once; the percentile normalise reads the volume
five times: three radix-select passes, statistics, apply), plus the end-to-end `inference` latency of a wrapper."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pytorchdeeplearing_amd import prepost as PP

PEAK = 8000.0
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for S, T in ((256, 96), (512, 160)):
    g = torch.Generator().manual_seed(0)
    vol = (torch.randn((S, S, S), generator=g) * 300.0).to(dev)
    small = torch.empty((T, T, T), device=dev)
    mask = (torch.rand((T, T, T), generator=g) > 0.5).to(torch.uint8).to(dev)
    rows = {
        "resample_linear %d^3->%d^3" % (S, T): (lambda: PP.resample3d(vol, (T, T, T)), 8 * 4 * T ** 3 + 4 * T ** 3),
        "resample_nearest_u8 %d^3->%d^3" % (T, S): (lambda: PP.resample3d(mask, (S, S, S), mode=PP.NEAREST), T ** 3 + S ** 3),
        "normalize_meanstd %d^3" % S: (lambda: PP.normalize_meanstd(vol, -100.0, 100.0), 3 * 4 * S ** 3),
        "normalize_percentile %d^3" % S: (lambda: PP.normalize_percentile(vol), 6 * 4 * S ** 3),
    }
    for name, (fn, nbytes) in rows.items():
        ms = timed(fn)
        print(json.dumps({"op": name, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                          "GBps": round(nbytes / ms / 1e6, 1), "frac_hbm": round(nbytes / ms / 1e6 / PEAK, 4)}))

import model
os.environ.setdefault("SEGENGINE_DTYPE", "f16")
m = model.BinaryVNet3dModel(96, 96, 96, 1, 1, 4, use_cuda=True)
arr = (np.random.RandomState(0).randn(200, 256, 256) * 200.0).astype(np.float32)
m.inference(arr)
t0 = time.perf_counter()
for _ in range(5):
    out = m.inference(arr)
ms = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"op": "BinaryVNet3dModel.inference 200x256x256 -> 96^3 -> back (host array in, host mask out)", "ms": round(ms, 2),
                  "pcie_MB": round((arr.nbytes + out.nbytes) / 1e6, 1)}))
ct = (-1024.0 + 224.0 * np.random.RandomState(1).rand(120, 160, 160)).astype(np.float32)
u = model.BinaryUNet3dModel(96, 96, 96, 1, 1, 4, use_cuda=True)
u.inference_patch(ct, newSpacing=(0.5, 0.5, 0.5), spacing=(1.0, 1.0, 1.0))
t0 = time.perf_counter()
for _ in range(3):
    out = u.inference_patch(ct, newSpacing=(0.5, 0.5, 0.5), spacing=(1.0, 1.0, 1.0))
ms = (time.perf_counter() - t0) / 3 * 1e3
print(json.dumps({"op": "BinaryUNet3dModel.inference_patch 120x160x160 @1mm -> 0.5mm (240x320x320), 8 windows of 96^3, batch 4", "ms": round(ms, 2)}))

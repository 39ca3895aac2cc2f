"""SURVEY.md section 8f N3, end to end: does the input pipeline feed the engine?  64 synthetic 96^3 volumes as .npy files on local disk (float32
image + int64 label, what the reference's offline tooling writes) -> `BinaryVNet3dModel.trainprocess`'s own batch loop (DataLoader of
datasetModelSegwithnpy, batch 4, shuffle) through model/pipeline.DevicePrefetcher -> SegEngine.train_step, against the same engine stepping on
ONE resident batch.  Prints one JSON line: epoch volumes/s of the pipeline-fed loop, the resident rate, their ratio, and the loader-only rate
(how fast the reader threads deliver batches when nothing trains).  usage: python tools/bench_pipeline.py [volumes] [epochs] [threads ...]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SEGENGINE_DTYPE", "f16")
import numpy as np
import torch
from torch.utils.data import DataLoader

from pytorchdeeplearing_amd.model.dataset import datasetModelSegwithnpy
from pytorchdeeplearing_amd.model.pipeline import DevicePrefetcher
from pytorchdeeplearing_amd import SegEngine, synthetic


def measure(nvol=64, epochs=3, threads=4, S=96, batch=4, direct=True, label_dtype=np.int64, dev=None, tmp=None):
    dev = dev or torch.device("cuda:0")
    os.environ["SEGENGINE_DIRECT_NPY"] = "1" if direct else "0"
    rng = np.random.default_rng(0)
    imgs, labs = [], []
    for i in range(nvol):
        ip, lp = os.path.join(tmp, "img%03d.npy" % i), os.path.join(tmp, "lab%03d.npy" % i)
        if not os.path.exists(ip):
            np.save(ip, rng.standard_normal((S, S, S), dtype=np.float32))
            np.save(lp, (rng.random((S, S, S)) > 0.8).astype(label_dtype) * 255)
        imgs.append(ip); labs.append(lp)
    ds = datasetModelSegwithnpy(imgs, labs, targetsize=(1, S, S, S))
    loader = DataLoader(ds, shuffle=True, batch_size=batch, num_workers=0, pin_memory=False)
    e = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
    synthetic.init_engine(e, seed=0)
    e.binarize_labels = True
    # resident-data rate (the number bench.py reports)
    x, y = next(iter(DevicePrefetcher(loader, dev, True, workers=threads)))
    for _ in range(30):
        e.train_step(x, y, "BinaryDiceLoss")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        e.train_step(x, y, "BinaryDiceLoss")
    torch.cuda.synchronize()
    resident = 60 * batch / (time.perf_counter() - t0)
    # loader only
    t0 = time.perf_counter()
    n = 0
    for x, y in DevicePrefetcher(loader, dev, True, workers=threads):
        n += x.shape[0]
    torch.cuda.synchronize()
    loader_only = n / (time.perf_counter() - t0)
    # the training loop of trainprocess (one warm epoch, then timed epochs)
    for x, y in DevicePrefetcher(loader, dev, True, workers=threads):
        e.train_step(x, y, "BinaryDiceLoss")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(epochs):
        for x, y in DevicePrefetcher(loader, dev, True, workers=threads):
            out3 = e.train_step(x, y, "BinaryDiceLoss")
            n += x.shape[0]
    torch.cuda.synchronize()
    fed = n / (time.perf_counter() - t0)
    return {"volumes": nvol, "size": S, "batch": batch, "epochs_timed": epochs, "reader_threads": threads, "direct_npy_path": direct,
            "label_dtype_on_disk": np.dtype(label_dtype).name, "resident_volumes_per_s": round(resident, 1), "loader_only_volumes_per_s": round(loader_only, 1),
            "pipeline_fed_volumes_per_s": round(fed, 1), "fed_over_resident": round(fed / resident, 3), "final_loss": round(float(out3[0]), 5),
            "host_cores": os.cpu_count()}


if __name__ == "__main__":
    nvol = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    threads = [int(t) for t in sys.argv[3:]] or [4]
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        for direct in (True, False):
            for t in threads:
                print(json.dumps(measure(nvol, epochs, t, direct=direct, tmp=tmp)), flush=True)

"""Input pipeline for `trainprocess`.  The reference loop reads one batch synchronously
(`num_workers=0`, modelVNet.py:508-510), binarises the int64 label on the host (`y[y != 0] = 1`, :576) and copies both
tensors with a blocking `.to(device)` (:577-578) before every step.  At ~900 volumes/s that serial chain starves the GPU,
so here:

  * `workers` reader threads (default 2, SEGENGINE_READER_THREADS) fetch and collate whole batches ahead of the consumer, in
    DataLoader order (np.load / image decode release the GIL; a DataLoader that brings its own worker PROCESSES is simply
    iterated by one thread);
  * the binarisation happens on the DEVICE: the consumer kernels read labels as (value != 0) (SEG_LABEL_BINARIZE,
    `SegEngine.binarize_labels`): uint8 / bool label batches (0/255 masks as stored) go to the GPU untouched.  int64 labels
    (what the reference datasets hand out) are narrowed to one byte per voxel before PCIe in ONE host pass (`ne(0)` viewed as
    uint8 for the binary nets - 8x less traffic; multi-class ids keep int64 if one does not fit a byte);
  * batches are staged in pinned memory and copied on a dedicated HIP stream; the consumer's stream waits on the copy's
    event, never the host (no `hipDeviceSynchronize` in the loop).

  * `.npy` volume data sets (model/dataset.py:82-115, what `trainprocess` of the 3-D wrappers reads) take a direct path (round 4): every reader
    thread memory-maps the two files of a sample and writes them straight into its slot of a recycled PINNED batch buffer - image cast to
    float32, label narrowed to one byte (`!= 0` for the binary nets) in the same pass.  One copy per voxel on the host instead of five
    (np.load, `.long()`, collate, narrowing, pin_memory - the last one a fresh pinned allocation per batch), which is what it takes to feed a
    GPU that trains ~1000 volumes/s from the page cache (tools/bench_pipeline.py, profiles/r04_pipeline_end_to_end.json).

Yields (x float32 (N,C,...) contiguous, y uint8/int64 (N,...) contiguous) on `device`, in DataLoader order."""
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

_END = object()


class _PinnedPool:
    """recycled pinned staging buffers: a buffer goes back to the pool with the event of the copy that read it and is handed out again once
    that event has completed (a fresh one is allocated while none is free: the pool grows to the pipeline depth and stays there)"""

    def __init__(self, cuda):
        self.cuda, self.free, self.lock = cuda, {}, threading.Lock()

    def take(self, shape, dtype):
        key = (tuple(shape), dtype)
        with self.lock:
            lst = self.free.setdefault(key, [])
            for i, (buf, ev) in enumerate(lst):
                if ev is None or ev.query():
                    lst.pop(i)
                    return buf
        buf = torch.empty(shape, dtype=dtype)
        return buf.pin_memory() if self.cuda else buf

    def give(self, buf, ev):
        with self.lock:
            self.free.setdefault((tuple(buf.shape), buf.dtype), []).append((buf, ev))


class DevicePrefetcher:
    def __init__(self, loader, device, binary, depth=2, workers=None):
        self.loader, self.device, self.binary, self.depth = loader, torch.device(device), binary, max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        if workers is None:
            workers = int(os.environ.get("SEGENGINE_READER_THREADS", "2"))     # 2 / 4 / 8 threads: 0.94 / 0.91 / 0.72 of the resident rate (profiles/r04_pipeline_end_to_end.json)
        self.workers = max(1, int(workers))
        self.pool = _PinnedPool(self.cuda)
        self._u8_ok = None                               # multi-class labels of this data set fit one byte (decided on the first batch)

    def __len__(self):
        return len(self.loader)

    def _prepare(self, batch):
        x, y = batch["image"], batch["label"]
        if y.dtype == torch.bool:
            y = y.view(torch.uint8)
        elif y.dtype != torch.uint8:
            if self.binary:
                y = y.ne(0).view(torch.uint8)              # the ONE narrowing pass (8 -> 1 byte per voxel) keeps "non-zero" exactly
            elif y.numel() and int(y.min()) >= 0 and int(y.max()) < 256:
                y = y.to(torch.uint8)
        x, y = x.float().contiguous(), y.contiguous()
        if self.cuda:
            x, y = x.pin_memory(), y.pin_memory()
        return x, y

    def _batches(self):
        """iterator over prepared (pinned) batches in loader order; several reader threads when the loader allows it"""
        ld = self.loader
        threaded = (self.workers > 1 and getattr(ld, "num_workers", 1) == 0 and getattr(ld, "batch_sampler", None) is not None and
                    hasattr(ld, "dataset") and hasattr(ld, "collate_fn"))
        if not threaded:
            for batch in ld:
                yield self._prepare(batch)
            return
        ds, collate = ld.dataset, ld.collate_fn
        from .dataset import datasetModelSegwithnpy
        direct = type(ds) is datasetModelSegwithnpy and os.environ.get("SEGENGINE_DIRECT_NPY", "1") != "0"

        def fetch_direct(indices):
            """.npy samples written straight into a pinned batch buffer (same values as ds[i] + collate + _prepare)"""
            c, d, h, w = ds.targetsize
            n = len(indices)
            x = self.pool.take((n, c, d, h, w), torch.float32)
            xn = x.numpy()
            labs = [np.load(ds.labels[i], mmap_mode="r") for i in indices]

            def fits_u8(l):                                # class ids after the reference's `.long()` (truncation): 0..255?
                if l.dtype.kind in "bu" and l.dtype.itemsize == 1:
                    return True
                lo, hi = l.min(), l.max()
                if l.dtype.kind == "f" and not (np.isfinite(lo) and np.isfinite(hi)):
                    raise ValueError("label file with NaN / inf values: %r" % (getattr(l, "filename", None),))
                return int(lo) >= 0 and int(hi) < 256
            if self.binary:
                y = self.pool.take((n, d, h, w), torch.uint8)
            else:
                # class ids: one byte while they fit.  The label dtype is decided per DATA SET, not per batch: the first batch whose ids do not fit
                # switches every following batch to int64 (the captured-graph launch mode needs tensors that keep their dtype).  The reader
                # threads run ahead of each other, so a thread only looks at its own batch here; the switch itself is applied in batch order (`in_order`)
                y = self.pool.take((n, d, h, w), torch.uint8 if (self._u8_ok is not False and all(fits_u8(l) for l in labs)) else torch.int64)
            yn = y.numpy()
            for k, i in enumerate(indices):
                img = np.load(ds.images[i], mmap_mode="r")
                # the data set reshapes (D, H, W) -> (1, D, H, W) and asserts the shape per dimension (model/dataset.py:90-93): a transposed volume must not pass
                assert img.ndim >= 3 and (1,) + tuple(img.shape[:3]) == (c, d, h, w) and img.size == c * d * h * w, (img.shape, ds.targetsize)
                np.copyto(xn[k], img.reshape(c, d, h, w), casting="unsafe")
                lab = labs[k].reshape(d, h, w)
                if lab.dtype.kind == "f":                  # `.long()` truncates toward zero BEFORE anything compares with 0: 0.5 / -0.5 are background
                    lab = np.trunc(lab)
                if self.binary:
                    np.not_equal(lab, 0, out=yn[k].view(np.bool_))
                else:
                    np.copyto(yn[k], lab, casting="unsafe")
            return x, y, True                             # pooled buffers: recycled after their copy

        def fetch(indices):
            if direct:
                return fetch_direct(indices)
            return self._prepare(collate([ds[i] for i in indices]))

        def in_order(item):
            """the per-data-set label dtype, applied as the batches leave in order: after the first int64 batch every later one is int64"""
            if not (direct and not self.binary):
                return item
            x, y, pooled = item
            if y.dtype == torch.int64:
                self._u8_ok = False
            elif self._u8_ok is False:                       # a byte batch that a reader thread finished before the switch was known
                wide = self.pool.take(tuple(y.shape), torch.int64)
                wide.copy_(y)
                self.pool.give(y, None)
                y = wide
            return x, y, pooled

        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            window = []
            for indices in ld.batch_sampler:                 # the sampler is walked once, in order: shuffling stays the loader's
                window.append(pool.submit(fetch, list(indices)))
                if len(window) > self.workers:
                    yield in_order(window.pop(0).result())
            while window:
                yield in_order(window.pop(0).result())

    def _reader(self, q, stop):
        try:
            for item in self._batches():
                if stop.is_set():
                    return
                q.put(item)
            q.put(_END)
        except BaseException as ex:                        # surface loader errors in the consumer
            q.put(ex)

    def __iter__(self):
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        t = threading.Thread(target=self._reader, args=(q, stop), daemon=True)
        t.start()
        try:
            pending = self._upload(q.get())
            while pending is not None:
                nxt = self._upload(q.get())                # enqueue the next copy before handing out the current batch
                yield self._ready(pending)
                pending = nxt
        finally:
            stop.set()
            while t.is_alive():                            # unblock a reader stuck on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)

    def _upload(self, item):
        if item is _END:
            return None
        if isinstance(item, BaseException):
            raise item
        x, y = item[0], item[1]
        if not self.cuda:
            return x, y, None
        with torch.cuda.stream(self.copy_stream):
            xd, yd = x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        if len(item) > 2:
            self.pool.give(x, ev); self.pool.give(y, ev)   # reusable once the copy that reads them has completed
        return xd, yd, ev, (x, y)                          # keep the pinned sources alive until the copy has been waited on

    def _ready(self, p):
        if not self.cuda:
            return p[0], p[1]
        xd, yd, ev = p[0], p[1], p[2]
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)                                 # stream-side wait: the host does not block
        xd.record_stream(cur); yd.record_stream(cur)
        return xd, yd

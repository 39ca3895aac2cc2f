"""Input pipeline for `trainprocess`.  The reference loop reads one batch synchronously
(`num_workers=0`, modelVNet.py:508-510), binarises the int64 label on the host (`y[y != 0] = 1`, :576) and copies both
tensors with a blocking `.to(device)` (:577-578) before every step.  At >600 volumes/s that serial chain starves the GPU,
so here:

  * a reader thread walks the DataLoader ahead of the consumer (np.load / image decode overlap the train step);
  * labels are binarised and narrowed to uint8 on the host (class ids < 256): one byte per voxel crosses PCIe instead of
    eight (the loss kernels read u8 / i32 / i64 / f32 labels alike, common.h:load_label);
  * batches are staged in pinned memory and copied on a dedicated HIP stream; the consumer's stream waits on the copy's
    event, never the host (no `hipDeviceSynchronize` in the loop).

Yields (x float32 (N,C,...) contiguous, y uint8/int64 (N,...) contiguous) on `device`, in DataLoader order."""
import queue
import threading

import torch

_END = object()


class DevicePrefetcher:
    def __init__(self, loader, device, binary, depth=2):
        self.loader, self.device, self.binary, self.depth = loader, torch.device(device), binary, max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _prepare(self, batch):
        x, y = batch["image"], batch["label"]
        if self.binary:
            y = (y != 0)                                   # == `y[y != 0] = 1` for the non-negative class ids of the datasets
        if y.dtype == torch.bool or (y.numel() and int(y.min()) >= 0 and int(y.max()) < 256):
            y = y.to(torch.uint8)
        x, y = x.float().contiguous(), y.contiguous()
        if self.cuda:
            x, y = x.pin_memory(), y.pin_memory()
        return x, y

    def _reader(self, q, stop):
        try:
            for batch in self.loader:
                if stop.is_set():
                    return
                q.put(self._prepare(batch))
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
        x, y = item
        if not self.cuda:
            return x, y, None
        with torch.cuda.stream(self.copy_stream):
            xd, yd = x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return xd, yd, ev, (x, y)                          # keep the pinned sources alive until the copy has been waited on

    def _ready(self, p):
        if not self.cuda:
            return p[0], p[1]
        xd, yd, ev = p[0], p[1], p[2]
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)                                 # stream-side wait: the host does not block
        xd.record_stream(cur); yd.record_stream(cur)
        return xd, yd

"""Bytes-bound or residency-bound?    Three measurements on ONE lease:

  (a) the copy rate of the lease: a wide elementwise copy of 1 GiB (read + write bytes / time)
  (b) that copy co-running on a second stream with the 16-channel 96^3 weight gradient (wgrad3_kernel<16,16> + reduce, one workgroup per CU,
      195 VGPRs / 44 KB LDS) and with the 16-channel 96^3 halo conv (conv3x16_kernel): aggregate algorithmic TB/s and what each partner loses
  (c) two copies on two streams (what two purely bandwidth-bound queues reach together)

If copy + weight gradient together stay near the copy rate, the overlapped region of the step is bytes-bound; if the aggregate falls well below it,
the weight-gradient workgroups cost the streaming kernels more than their bytes (wave slots / LDS / issue).  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchdeeplearing_amd import _capi, ops  # noqa: E402

dev = torch.device("cuda")
lib = _capi.lib_for(dev)
GIB = 1 << 30
src = torch.empty(GIB // 4, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
src2, dst2 = torch.empty_like(src).normal_(), torch.empty_like(src)
N, S, C = 4, 96, 16
x = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
dr = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
nb = lib.seg_op_wgrad3_partial_bytes(3, N, S, S, S, C, C)
partial = ops.aligned_empty(nb, dev).view(torch.float32)
dw = torch.zeros(C, C, 3, 3, 3, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.1
wfrag = ops.pack(w, "conv_fwd", "f16", frag="all")
out = ops.aligned_like(torch.empty(N, S, S, S, C, device=dev).half())
T_BYTES = 2.0 * N * S ** 3 * C * 2          # one operand in + one out / two operands in

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def copy_op(s=src, d=dst):
    d.copy_(s)


def wgrad_op():
    lib.check(lib.seg_op_wgrad3(dr.data_ptr(), x.data_ptr(), partial.data_ptr(), dw.data_ptr(), N, S, S, S, C, C, 3, _capi.DTYPE["f16"],
                                _capi.stream_for(dev)), "seg_op_wgrad3")


def conv_op():
    ops.conv3x(x, wfrag, "f16", 3, C, out=out)


def run(a_fn, a_n, b_fn=None, b_n=0):
    """a_n launches of a_fn on stream A next to b_n launches of b_fn on stream B; returns (wall ms, A's span ms, B's span ms)"""
    torch.cuda.synchronize()
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    with torch.cuda.stream(sa):
        ea0.record()
        for _ in range(a_n):
            a_fn()
        ea1.record()
    if b_fn is not None:
        with torch.cuda.stream(sb):
            eb0.record()
            for _ in range(b_n):
                b_fn()
            eb1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) * 1e3
    return wall, ea0.elapsed_time(ea1), (eb0.elapsed_time(eb1) if b_fn is not None else 0.0)


for fn in (copy_op, wgrad_op, conv_op):
    with torch.cuda.stream(sa):
        for _ in range(3):
            fn()
torch.cuda.synchronize()

res = {"build": lib.build_info()}
_, t_copy, _ = run(copy_op, 20)
copy_tbps = 20 * 2.0 * GIB / (t_copy * 1e-3) / 1e12
res["a_copy_alone"] = {"ms_per_GiB_copy": round(t_copy / 20, 4), "TBps_read_plus_write": round(copy_tbps, 3)}
_, t_w, _ = run(wgrad_op, 40)
res["wgrad3_16ch_96_alone"] = {"us": round(t_w / 40 * 1e3, 1), "TBps_algorithmic": round(T_BYTES / (t_w / 40 * 1e-3) / 1e12, 3)}
_, t_c, _ = run(conv_op, 40)
res["conv3x16_96_alone"] = {"us": round(t_c / 40 * 1e3, 1), "TBps_algorithmic": round(T_BYTES / (t_c / 40 * 1e-3) / 1e12, 3)}

# co-runs: sized so both streams are busy for about the same time
for name, fn, t_alone in (("wgrad3_16ch_96", wgrad_op, t_w / 40), ("conv3x16_96", conv_op, t_c / 40)):
    ncopy = 24
    nk = max(4, int(1.3 * ncopy * (t_copy / 20) / t_alone))
    wall, ta, tb = run(copy_op, ncopy, fn, nk)
    both = min(ta, tb)                          # the window in which both queues had work (they start together)
    # bytes moved inside the common window, assuming each queue's rate is uniform over its own span
    bytes_common = ncopy * 2.0 * GIB * both / ta + nk * T_BYTES * both / tb
    res["b_copy_plus_" + name] = {"copies": ncopy, "kernels": nk, "copy_span_ms": round(ta, 3), "kernel_span_ms": round(tb, 3),
                                  "copy_TBps_beside": round(ncopy * 2.0 * GIB / (ta * 1e-3) / 1e12, 3),
                                  "kernel_us_beside": round(tb / nk * 1e3, 1), "kernel_slowdown": round(tb / nk / t_alone, 2),
                                  "aggregate_TBps_common_window": round(bytes_common / (both * 1e-3) / 1e12, 3),
                                  "aggregate_over_copy_rate": round(bytes_common / (both * 1e-3) / 1e12 / copy_tbps, 3)}
wall, ta, tb = run(copy_op, 20, lambda: copy_op(src2, dst2), 20)
res["c_two_copies"] = {"aggregate_TBps": round(40 * 2.0 * GIB / (max(ta, tb) * 1e-3) / 1e12, 3)}
print(json.dumps(res))

"""Experiment (round 5): the batch as TWO sample groups on two queues, staggered, so that one group's latency-bound deep levels (24^3 ... 6^3: a third of the
step for 15 % of its bytes) run beside the other group's bytes-bound fine levels.  Two engines of N/2 samples share ONE parameter buffer; GroupNorm statistics and
dropout masks are per sample, the loss is the exact batch-global one (sums exchanged between reduction and finalize, as between ranks), the gradients of the two
groups are summed before the fused optimiser.  Python-level prototype on the per-call C-ABI entry points; prints ms per step against the one-engine step.
usage: python tools/experiments/two_groups.py [stagger_fwd_us,stagger_bwd_us ...]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import SegEngine, synthetic, _capi
from pytorchdeeplearing_amd.engine import _ptr

dev = torch.device("cuda")
N, S, LOSS = 4, 96, "BinaryDiceLoss"
x, y = synthetic.synthetic_batch(N, (S, S, S), 1, 1, seed=1)
x, y = x.to(dev), y.to(dev)
alpha = torch.ones(1, device=dev)


def timeit(fn, steps=20, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


# ---- reference: one engine, the product's one-call step and its per-call twin
e = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
synthetic.init_engine(e, seed=0)
ms_one = timeit(lambda: e.train_step(x, y, LOSS, class_alpha=alpha))
print(json.dumps({"arm": "one engine, N=4, seg_train_step", "ms_per_step": round(ms_one, 3)}), flush=True)
del e
torch.cuda.empty_cache()

# ---- two groups
H = N // 2
eA = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
eB = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
synthetic.init_engine(eA, seed=0)
eA.plan(H, (S, S, S)); eB.plan(H, (S, S, S))
eB.params = eA.params; eB.rebind()
eA.init_optimizer()
eB.seed = eA.seed ^ 0x1234567
xs, ys = (x[:H].contiguous(), x[H:].contiguous()), (y[:H].contiguous(), y[H:].contiguous())
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
sA, sB = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=-1)
lib = eA.lib
CLK = 2.4e3          # _sleep cycles per microsecond (approximately; the stagger is a knob, not a measurement)


def loss_args(e_, logits, tgt):
    return (_ptr(logits), _ptr(tgt), _capi.label_type(tgt, False), H, 1, e_.V, _capi.LOSS_KIND[LOSS], 0.25, 2.0)


def step(d_f, d_b):
    outs = []
    with torch.cuda.stream(sA):
        la, pa = eA.forward(xs[0], _capi.MASKS_RANDOM)
        lib.check(lib.seg_loss_reduce(*loss_args(eA, la, ys[0]), _ptr(eA._loss_ws), eA.stream()), "reduce")
    with torch.cuda.stream(sB):
        if d_f > 0:
            torch.cuda._sleep(int(d_f * CLK))
        lb, pb = eB.forward(xs[1], _capi.MASKS_RANDOM)
        lib.check(lib.seg_loss_reduce(*loss_args(eB, lb, ys[1]), _ptr(eB._loss_ws), eB.stream()), "reduce")
    nd = lib.seg_loss_shared_doubles()
    shA, shB = eA._loss_ws[:8 * nd].view(torch.float64), eB._loss_ws[:8 * nd].view(torch.float64)
    sA.wait_stream(sB)
    with torch.cuda.stream(sA):
        shA.add_(shB); shB.copy_(shA)
    sB.wait_stream(sA)
    with torch.cuda.stream(sA):
        lib.check(lib.seg_loss_finalize(*loss_args(eA, la, ys[0]), _ptr(alpha), N, _ptr(eA._loss_ws), _ptr(eA._out3), eA.stream()), "finalize")
        dla = eA.loss_backward(la, ys[0], LOSS)
        eA.backward(dla)
    with torch.cuda.stream(sB):
        lib.check(lib.seg_loss_finalize(*loss_args(eB, lb, ys[1]), _ptr(alpha), N, _ptr(eB._loss_ws), _ptr(eB._out3), eB.stream()), "finalize")
        dlb = eB.loss_backward(lb, ys[1], LOSS)
        if d_b > 0:
            torch.cuda._sleep(int(d_b * CLK))
        eB.backward(dlb)
    sA.wait_stream(sB)
    with torch.cuda.stream(sA):
        eA.grads.add_(eB.grads)
        eA.adam_step(grad_div=1.0)
        eA.pack_weights()
    sB.wait_stream(sA)
    with torch.cuda.stream(sB):
        eB.packed = False
        eB.pack_weights()
    return eA._out3


arms = [tuple(float(v) for v in a.split(",")) for a in sys.argv[1:]] or [(0, 0), (150, 150), (300, 300), (450, 450), (300, 600), (600, 600)]
for d_f, d_b in arms:
    ms = timeit(lambda: step(d_f, d_b))
    print(json.dumps({"arm": "two groups of %d, stagger fwd %g us / bwd %g us" % (H, d_f, d_b), "ms_per_step": round(ms, 3), "vs_one_engine": round(ms_one / ms, 4),
                      "loss": round(float(eA._out3[0]), 5)}), flush=True)

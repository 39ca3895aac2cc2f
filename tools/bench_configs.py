"""Train-step time of the BASELINE.json configs on one GPU (diagnostics; bench.py is the contract benchmark)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import synthetic as seg
from pytorchdeeplearing_amd import SegEngine, _capi

CONFIGS = {
    "C2 VNet2d 16x512^2 f16 2-class": ("vnet", 2, (16, 1, 512, 512), 2, "MutilDiceLoss", "f16"),
    "C3 VNet3d 4x96^3 f16 binary": ("vnet", 3, (4, 1, 96, 96, 96), 1, "BinaryDiceLoss", "f16"),
    "C4 UNet3d 2x128^3 f16 4-class": ("unet", 3, (2, 1, 128, 128, 128), 4, "MutilDiceLoss", "f16"),
    "C5 VNet3d 1x160^3 bf16 binary": ("vnet", 3, (1, 1, 160, 160, 160), 1, "BinaryCrossEntropyDiceLoss", "bf16"),
}
dev = torch.device("cuda")
ONLY = os.environ.get("SEG_BENCH_ONLY", "")          # e.g. "C2,C4": a subset of the configs (A/B sessions); the clDice legs run only without a filter
for name, (kind, ndim, shape, ncls, loss, dt) in CONFIGS.items():
    if ONLY and name.split()[0] not in ONLY.split(","):
        continue
    e = SegEngine(kind, ndim, shape[1], ncls, dtype=dt, device=dev)
    seg.init_engine(e, seed=0)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=1)
    x, y = x.to(dev), y.to(dev)
    alpha = torch.ones(ncls, device=dev)
    for _ in range(3): e.train_step(x, y, loss, class_alpha=alpha)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): e.train_step(x, y, loss, class_alpha=alpha)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    prof = {}
    if not os.environ.get("SEG_BENCH_NOPROF"):          # (a kernel trace of the un-instrumented step: no per-class event brackets)
        e.profile_enable(_capi.KERNEL_CLASSES)
        for _ in range(2): e.train_step(x, y, loss, class_alpha=alpha)
        torch.cuda.synchronize()
        prof = {k: round(v["ms"] / 2, 2) for k, v in e.profile_read().items()}
    print(json.dumps({"config": name, "ms_per_step": round(ms, 2), "samples_per_s": round(shape[0] / ms * 1e3, 1), "class_ms": prof}))
    del e
    torch.cuda.empty_cache()

if ONLY:
    sys.exit(0)
# BASELINE configs[4] as worded: VNet3d 1x160^3 bf16 **+ clDice loss** (Dice on the logits + soft-clDice on the probabilities).
# clDice is an autograd Function over the HIP skeleton kernels, so this step goes through the module (autograd) path:
# net(x) -> losses -> loss.backward() (engine backward inside) -> fused AdamW on the engine's flat buffers.
os.environ["SEGENGINE_DTYPE"] = "bf16"
from pytorchdeeplearing_amd import networks as NW, losses as LS
from pytorchdeeplearing_amd.lossescldice import Binary_Soft_cldice_loss
net = NW.VNet3d(1, 1).to(dev)
net.apply(NW.initialize_weights)
net.train()
eng = net.engine
eng.init_optimizer()
x, y = seg.synthetic_batch(1, (160, 160, 160), 1, 1, seed=1)
x, y = x.to(dev), y.to(dev)
yf = y.float().reshape(1, 1, 160, 160, 160)
dice, cld = LS.BinaryDiceLoss(), Binary_Soft_cldice_loss()


def cl_step():
    for p in net.parameters():
        p.grad = None
    logits, probs = net(x)
    loss = dice(logits, y) + cld(probs, yf)
    loss.backward()
    eng.adam_step()
    eng.pack_weights()
    return loss


first = float(cl_step())
for _ in range(2):
    cl_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    last = cl_step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps({"config": "C5 VNet3d 1x160^3 bf16 + Dice + soft-clDice (autograd path)", "ms_per_step": round(ms, 2),
                  "samples_per_s": round(1e3 / ms, 1), "loss_first": round(first, 5), "loss_after_13_steps": round(float(last), 5)}))

# the same loss as ONE engine call behind SegEngine.train_step (seg_cldice_binary: planned workspace, no autograd, no torch elementwise ops)
del net, eng
torch.cuda.empty_cache()
from pytorchdeeplearing_amd import SegEngine
e = SegEngine("vnet", 3, 1, 1, dtype="bf16", device=dev)
seg.init_engine(e, seed=0)
first = float(e.train_step(x, y, "BinaryDiceLoss", cldice_weight=1.0)[0])
for _ in range(2):
    e.train_step(x, y, "BinaryDiceLoss", cldice_weight=1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    out3 = e.train_step(x, y, "BinaryDiceLoss", cldice_weight=1.0)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps({"config": "C5 VNet3d 1x160^3 bf16 + Dice + soft-clDice (engine call: train_step(cldice_weight=1))", "ms_per_step": round(ms, 2),
                  "samples_per_s": round(1e3 / ms, 1), "loss_first": round(first, 5), "loss_after_13_steps": round(float(out3[0]), 5)}))

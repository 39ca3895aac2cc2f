"""EXPERIMENT (test infrastructure, runs on the CPU; not collected by pytest): where does the 16-bit gradient error of the engine come from?
The oracle's VNet3d (fp32 arithmetic) with the engine's STORAGE roundings emulated: every conv output and every activation is rounded to the run
dtype in the forward pass ("fwd"), every gradient that the engine stores - d(raw) of a unit and the data-gradient handed to the producer - is
rounded in the backward pass ("bwd").  Per-parameter relative L2 error against the fp64 oracle, for fwd only / bwd only / both, and with the
backward rounding switched off at the deep levels only (VERDICT r03 task 7's proposal: fp32 dY at 12^3 / 6^3).
usage: python tests/exp_lowp_fidelity.py [size=48] [dtype=bf16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from oracle import seg_oracle as seg

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
torch.set_num_threads(min(32, os.cpu_count() or 1))
CFG = {"fwd": False, "bwd": False, "bwd_min_vox": 0}
LOSS_SCALE = 16384.0 if DT == torch.float16 else 1.0


def rnd(t):
    return t.to(DT).to(t.dtype)


class Store(torch.autograd.Function):
    """a tensor the engine keeps in the run dtype: rounded when written (forward), its gradient rounded when written (backward)"""
    @staticmethod
    def forward(ctx, x):
        ctx.vox = x[0, 0].numel()
        return rnd(x) if (CFG["fwd"] and x.dtype == torch.float32) else x

    @staticmethod
    def backward(ctx, g):
        if CFG["bwd"] and g.dtype == torch.float32 and ctx.vox >= CFG["bwd_min_vox"]:
            return rnd(g * LOSS_SCALE) / LOSS_SCALE        # the engine's f16 gradients are stored times the loss scale (16384); bf16 needs none
        return g


_conv3d, _convT3d, _gn = F.conv3d, F.conv_transpose3d, seg._gn_drop_relu
seg._conv = lambda ndim: (lambda *a, **k: Store.apply(_conv3d(*a, **k)))
seg._convT = lambda ndim: (lambda *a, **k: Store.apply(_convT3d(*a, **k)))
seg._gn_drop_relu = lambda x, w, b, drop: Store.apply(_gn(x, w, b, drop))

params = seg.perturb_params(seg.init_params("vnet", 3, 1, 1, seed=0), seed=7)
x, y = seg.synthetic_batch(1, (S, S, S), 1, 1, seed=1)
g = torch.Generator().manual_seed(5)
masks = seg.draw_masks("vnet", 1, generator=g)


def grads(dtype):
    P = {k: v.to(dtype) for k, v in params.items()}
    r = seg.forward_backward("vnet", P, x.to(dtype), y, "BinaryDiceLoss", masks=[m.to(dtype) for m in masks])
    return {k: v.double() for k, v in r["grads"].items()}


ref = grads(torch.float64)


def report(tag):
    got = grads(torch.float32)
    errs = {k: float((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-300)) for k in ref}
    order = sorted(errs, key=lambda k: -errs[k])
    med = sorted(errs.values())[len(errs) // 2]
    cos = min(float((got[k] * ref[k]).sum() / (got[k].norm() * ref[k].norm() + 1e-300)) for k in ref)
    print("%-34s worst %.3f  median %.4f  min cosine %.4f   worst tensors: %s" % (tag, errs[order[0]], med, cos, ", ".join("%s %.3f" % (k, errs[k]) for k in order[:4])), flush=True)
    return errs


print("VNet3d 1x1x%d^3, BinaryDiceLoss, %s storage emulation, relative L2 per parameter tensor vs the fp64 oracle" % (S, DT))
CFG.update(fwd=False, bwd=False); report("fp32 (no rounding)")
CFG.update(fwd=True, bwd=False); report("forward storage rounded")
CFG.update(fwd=False, bwd=True); report("backward storage rounded")
CFG.update(fwd=True, bwd=True); e_both = report("both")
for lvl, vox in (("<= 12^3 kept fp32 (S/8)", (S // 8 + 1) ** 3), ("<= 24^3 kept fp32 (S/4)", (S // 4 + 1) ** 3)):
    CFG.update(fwd=True, bwd=True, bwd_min_vox=vox); report("both, backward " + lvl)

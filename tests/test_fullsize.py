"""Parity at BASELINE.json's FULL sizes.
Direct comparisons with the CPU oracle (oracle/seg_oracle.py, pinned to the real reference by tests/test_oracle.py):
  * eval forward of every config C2 - C5 on the host (a few seconds each on 32 threads) vs the engine: fp32 run dtype ->
    logits within 1e-3 (BASELINE.json north_star), integer masks identical except at voxels within 5e-5 of the threshold,
    Dice equal to 1e-6; f16 / bf16 run dtypes -> mask flip fraction <= 7e-4 / 5e-3 against the ORACLE's mask;
  * one 1x96^3 (VNet3d) and one 1x128^3 (UNet3d) forward + backward: per-tensor relative-L2 gradient gate against the
    oracle (the size-dependent policies - 1024-row GroupNorm slabs, weight-gradient partial policy, multi-box stems,
    large index arithmetic - only exist at these sizes), plus a printed per-tensor report for the 16-bit run dtypes.
And size-independent properties:
  * loss / Dice metric of the engine's own logits against the oracle formulas (cheap on CPU even at 3.5 M voxels);
  * batch-permutation equivariance and batch-split invariance of the forward (GroupNorm is per sample);
  * probabilities are a distribution (sigmoid range, softmax sums to one);
  * directional derivative: (L(p + eps d) - L(p - eps d)) / 2 eps == <grad, d> for the engine's own loss and gradients
    (fp32 run dtype) — a whole-network check of forward vs backward at full size;
  * f16 / bf16 run dtypes stay within the Dice tolerance of the fp32 run on the same weights."""
import os

import numpy as np

import pytest
import torch

from oracle import seg_oracle as seg
from pytorchdeeplearing_amd import SegEngine, _capi

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

CONFIGS = {
    # BASELINE.json configs[0] (the reference's own CPU-runnable case, at its own size through the engine) and configs[1..4]
    "C1_unet2d": ("unet", 2, (2, 1, 256, 256), 1, "BinaryDiceLoss"),
    "C2_vnet2d": ("vnet", 2, (16, 1, 512, 512), 2, "MutilDiceLoss"),
    "C3_vnet3d": ("vnet", 3, (4, 1, 96, 96, 96), 1, "BinaryDiceLoss"),
    "C4_unet3d": ("unet", 3, (2, 1, 128, 128, 128), 4, "MutilDiceLoss"),
    "C5_vnet3d": ("vnet", 3, (1, 1, 160, 160, 160), 1, "BinaryCrossEntropyDiceLoss"),
}


# 16-bit gates per config = the MI355X measurement + 25-40 % (profiles/r04_fullsize_report.txt, r05_fullsize_report.txt): (mask-flip fraction, Dice difference).
# C1 is an fp32 config (BASELINE configs[0]); its 16-bit rows are there for completeness - a 2-D net at random initialisation puts more voxels next to
# the 0.5 threshold (measured: f16 6.4e-4 / 6.5e-4 flips, bf16 5.36e-3 / 5.36e-3; Dice differences 2.6e-5 / 4.5e-4)
LOWP_GATES = {"C1_unet2d": {"f16": (9e-4, 2e-4), "bf16": (7e-3, 7e-4)}}


def lowp_gate(tag, dtype, flip_tol, dice_tol):
    return LOWP_GATES.get(tag, {}).get(dtype, (flip_tol, dice_tol))


def make(tag, dtype):
    kind, ndim, shape, ncls, loss = CONFIGS[tag]
    e = SegEngine(kind, ndim, shape[1], ncls, dtype=dtype, device=DEV)
    params = seg.perturb_params(seg.init_params(kind, ndim, shape[1], ncls, seed=0), seed=7)
    e.load_state_dict(params)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=3)
    return e, x.to(DEV), y.to(DEV), ncls, loss


@pytest.mark.parametrize("tag", list(CONFIGS))
def test_forward_properties_full_size(tag):
    e, x, y, ncls, loss = make(tag, "f32")
    logits, probs = e.forward(x)
    logits, probs = logits.clone(), probs.clone()
    assert torch.isfinite(logits).all()
    if ncls == 1:
        assert float(probs.min()) >= 0 and float(probs.max()) <= 1
        assert torch.allclose(probs, torch.sigmoid(logits), atol=1e-6)
    else:
        assert float((probs.sum(1) - 1).abs().max()) < 1e-5
    # loss + metric kernels vs the oracle formulas on the engine's own logits
    alpha = torch.ones(ncls, device=DEV)
    out3 = e.loss_forward(logits, y, loss, class_alpha=alpha).cpu()
    ref_loss = seg.loss_fn(loss, alpha.cpu())(logits.cpu(), y.cpu())
    ref_dice = seg.dice_coeff(probs.cpu(), y.cpu()) if ncls == 1 else seg.multiclass_dice_coeff(probs.cpu(), y.cpu())
    assert abs(float(out3[0]) - float(ref_loss)) < 2e-5
    assert abs(float(out3[1]) - float(ref_dice)) < 1e-6          # integer-mask Dice
    n = x.shape[0]
    if n > 1:
        # batch permutation equivariance
        perm = torch.arange(n - 1, -1, -1, device=DEV)
        lp, _ = e.forward(x[perm].contiguous())
        assert float((lp[perm] - logits).abs().max()) < 2e-5
        # batch-split invariance: the first sample alone gives the same logits
        e1 = SegEngine(CONFIGS[tag][0], CONFIGS[tag][1], x.shape[1], ncls, dtype="f32", device=DEV)
        e1.load_state_dict(e.state_dict())
        l1, _ = e1.forward(x[:1].contiguous())
        assert float((l1 - logits[:1]).abs().max()) < 2e-5


@pytest.mark.parametrize("tag", ["C3_vnet3d", "C4_unet3d"])
def test_directional_derivative_full_size(tag):
    e, x, y, ncls, loss = make(tag, "f32")
    alpha = torch.ones(ncls, device=DEV)
    logits, _ = e.forward(x)
    e.loss_forward(logits, y, loss, class_alpha=alpha)
    e.backward(e.loss_backward(logits, y, loss))
    g = e.grads.clone() / e.loss_scale
    torch.manual_seed(0)
    d = torch.randn_like(g)
    d = d / d.norm() * g.norm().clamp_min(1e-12) / g.norm().clamp_min(1e-12)
    p0 = e.params.clone()
    eps = 2e-3
    vals = []
    for sgn in (+1, -1):
        e.params.copy_(p0 + sgn * eps * d)
        e.packed = False
        lg, _ = e.forward(x)
        vals.append(float(e.loss_forward(lg, y, loss, class_alpha=alpha)[0].double()))
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g.double() * d.double()).sum())
    assert abs(fd - an) <= 0.05 * max(abs(an), abs(fd)) + 2e-4, (fd, an)


@pytest.mark.parametrize("tag", ["C1_unet2d", "C3_vnet3d", "C5_vnet3d"])
# gates = the MI355X measurement + 25-40 % (profiles/r04_fullsize_report.txt: f16 flips 4.4e-5 ... 4.6e-4, bf16 3.2e-4 ... 3.7e-3; Dice differences <= 2.1e-5 / 8.8e-5)
@pytest.mark.parametrize("dtype,flip_tol,dice_tol", [("f16", 7e-4, 2e-4), ("bf16", 5e-3, 5e-4)])
def test_low_precision_vs_fp32_full_size(tag, dtype, flip_tol, dice_tol):
    flip_tol, dice_tol = lowp_gate(tag, dtype, flip_tol, dice_tol)
    e32, x, y, ncls, loss = make(tag, "f32")
    l32, p32 = e32.forward(x)
    d32 = float(e32.loss_forward(l32, y, loss)[1])
    m32 = (p32 > 0.5).clone()
    del e32
    e, _, _, _, _ = make(tag, dtype)
    l, p = e.forward(x)
    d = float(e.loss_forward(l, y, loss)[1])
    flips = float(((p > 0.5) != m32).float().mean())
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write("forward %s %s vs the engine's own f32 run: mask flips %.3e\n" % (tag, dtype, flips))
    assert flips <= flip_tol, flips
    assert abs(d - d32) <= dice_tol
    # and a full optimisation step runs at this size
    e.train_step(x, y, loss)
    torch.cuda.synchronize()
    assert torch.isfinite(e.params).all()


# ---------------------------------------------------------------------------------------------------------------------
# direct oracle comparisons at the BASELINE sizes
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_eval(tag):
    kind, ndim, shape, ncls, loss = CONFIGS[tag]
    params = seg.perturb_params(seg.init_params(kind, ndim, shape[1], ncls, seed=0), seed=7)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=3)
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, max(1, (torch.get_num_threads() or 1))))
    try:
        with torch.no_grad():
            logits, probs = seg.net_forward(kind, params, x)
    finally:
        torch.set_num_threads(nt)
    return params, x, y, logits, probs


_ORACLE_CACHE = {}


def oracle_eval(tag):
    if tag not in _ORACLE_CACHE:
        _ORACLE_CACHE[tag] = _oracle_eval(tag)
    return _ORACLE_CACHE[tag]


def _mask(probs, ncls):
    return (probs > 0.5) if ncls == 1 else probs.argmax(1)


@pytest.mark.parametrize("tag", list(CONFIGS))
def test_oracle_forward_full_size_f32(tag):
    kind, ndim, shape, ncls, loss = CONFIGS[tag]
    params, x, y, ref_logits, ref_probs = oracle_eval(tag)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype="f32", device=DEV)
    e.load_state_dict(params)
    logits, probs = e.forward(x.to(DEV))
    out3 = e.loss_forward(logits, y.to(DEV), loss, class_alpha=torch.ones(ncls, device=DEV)).cpu()
    logits, probs = logits.cpu(), probs.cpu()
    err = float((logits - ref_logits).abs().max())
    print("%s f32: logits max|d| = %.3e over %d values" % (tag, err, logits.numel()))
    assert err < 1e-3                                                             # north_star tolerance
    # integer masks: the Dice metric thresholds every class probability at 0.5 (metric.py:146-181)
    flip = (probs > 0.5) != (ref_probs > 0.5)
    nflip = int(flip.sum())
    tie = float((ref_probs[flip] - 0.5).abs().max()) if nflip else 0.0
    ref_dice = seg.dice_coeff(ref_probs, y) if ncls == 1 else seg.multiclass_dice_coeff(ref_probs, y)
    # north_star: "Dice metric bit-identical at integer mask level" - the flip count per config is printed and kept (profiles/rNN_fullsize_report.txt);
    # a flipped voxel is only tolerated where the ORACLE's own probability is within 5e-5 of the threshold (a numerical tie between two fp32 evaluations)
    line = "forward %s f32 vs the fp32 oracle: logits max|d| %.3e, integer-mask flips %d of %d (largest |p_oracle - 0.5| among them %.1e), Dice %.7f vs %.7f" % (
        tag, err, nflip, flip.numel(), tie, float(out3[1]), float(ref_dice))
    print(line)
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write(line + "\n")
    assert nflip <= 2e-5 * flip.numel(), nflip
    if nflip:
        assert tie < 5e-5                                                         # only numerically tied voxels may differ
    assert abs(float(out3[1]) - float(ref_dice)) < (1e-6 if nflip == 0 else 1e-4)
    ref_loss = seg.loss_fn(loss, torch.ones(ncls))(ref_logits, y)
    assert abs(float(out3[0]) - float(ref_loss)) < 2e-5


@pytest.mark.parametrize("tag", list(CONFIGS))
# gates = the MI355X measurement + 25-40 % (profiles/r04_fullsize_report.txt: f16 flips 4.4e-5 ... 4.6e-4, bf16 3.2e-4 ... 3.7e-3; Dice differences <= 2.1e-5 / 8.8e-5)
@pytest.mark.parametrize("dtype,flip_tol,dice_tol", [("f16", 7e-4, 2e-4), ("bf16", 5e-3, 5e-4)])
def test_oracle_forward_full_size_low_precision(tag, dtype, flip_tol, dice_tol):
    flip_tol, dice_tol = lowp_gate(tag, dtype, flip_tol, dice_tol)
    kind, ndim, shape, ncls, loss = CONFIGS[tag]
    params, x, y, ref_logits, ref_probs = oracle_eval(tag)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype=dtype, device=DEV)
    e.load_state_dict(params)
    logits, probs = e.forward(x.to(DEV))
    out3 = e.loss_forward(logits, y.to(DEV), loss, class_alpha=torch.ones(ncls, device=DEV)).cpu()
    probs = probs.cpu()
    flips = float(((probs > 0.5) != (ref_probs > 0.5)).float().mean())
    ref_dice = seg.dice_coeff(ref_probs, y) if ncls == 1 else seg.multiclass_dice_coeff(ref_probs, y)
    print("%s %s: mask flips %.2e, Dice %.6f vs oracle %.6f" % (tag, dtype, flips, float(out3[1]), float(ref_dice)))
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write("forward %s %s vs the fp32 oracle: mask flips %.3e, Dice %.6f vs %.6f\n" % (tag, dtype, flips, float(out3[1]), float(ref_dice)))
    assert flips <= flip_tol, flips
    assert abs(float(out3[1]) - float(ref_dice)) <= dice_tol


GRAD_CASES = {
    # one sample of C3 / C4: every level of the net at its BASELINE spatial size
    "C3_vnet3d_1x96": ("vnet", 3, (1, 1, 96, 96, 96), 1, "BinaryDiceLoss"),
    "C4_unet3d_1x128": ("unet", 3, (1, 1, 128, 128, 128), 4, "MutilDiceLoss"),
}
_GRAD_CACHE = {}
LOWP_CAL = 1.25          # engine-vs-autocast bound of test_low_precision_gradients_calibrated_against_autocast (SEG_LOWP_CAL overrides: measurement runs)


def oracle_grads(tag):
    if tag not in _GRAD_CACHE:
        kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
        params = seg.perturb_params(seg.init_params(kind, ndim, shape[1], ncls, seed=0), seed=7)
        x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=5)
        nt = torch.get_num_threads()
        torch.set_num_threads(min(32, max(1, nt)))
        try:
            r = seg.forward_backward(kind, params, x, y, loss, alpha=torch.ones(ncls))
        finally:
            torch.set_num_threads(nt)
        _GRAD_CACHE[tag] = (params, x, y, r)
    return _GRAD_CACHE[tag]


def _grad_report(e, r):
    rows = []
    for k, g in e.grad_dict().items():
        ref = r["grads"][k].double()
        nrm = float(ref.norm()) + 1e-30
        rows.append((float((g.cpu().double() - ref).norm()) / nrm, k, nrm))
    rows.sort(reverse=True)
    return rows


_GRAD64_CACHE = {}


def oracle_grads_fp64(tag):
    """the SAME case through the oracle in float64 (the fp32 oracle is itself a few 1e-2 from its fp64 twin at these sizes
    because ReLU gates at ~0 flip; the fp64 run is the yardstick that tells the two apart)."""
    if tag not in _GRAD64_CACHE:
        kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
        params, x, y, _ = oracle_grads(tag)
        nt = torch.get_num_threads()
        torch.set_num_threads(min(32, max(1, nt)))
        try:
            _GRAD64_CACHE[tag] = seg.forward_backward(kind, {k: v.double() for k, v in params.items()}, x.double(), y, loss, alpha=torch.ones(ncls).double())
        finally:
            torch.set_num_threads(nt)
    return _GRAD64_CACHE[tag]


def test_f32_gradients_full_size_against_the_fp64_oracle():
    """C3 at its BASELINE spatial size (1 x 96^3): per-tensor relative L2 of the engine's f32 gradients against the float64 oracle, next to the
    fp32 torch-CPU oracle's own distance from float64.  The engine has to be as close to fp64 as the reference's own fp32 arithmetic is
    (factor 2 + 1e-3 absolute): a gate that BOUNDS the error instead of the 3e-2 the fp32-vs-fp32 comparison needs."""
    tag = "C3_vnet3d_1x96"
    kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
    params, x, y, r32 = oracle_grads(tag)
    r64 = oracle_grads_fp64(tag)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype="f32", device=DEV)
    e.load_state_dict(params)
    logits, probs = e.forward(x.to(DEV))
    e.loss_forward(logits, y.to(DEV), loss, class_alpha=torch.ones(ncls, device=DEV))
    e.backward(e.loss_backward(logits, y.to(DEV), loss))
    assert float((logits.cpu().double() - r64["logits"]).abs().max()) < 1e-3
    worst_e = worst_o = 0.0
    rows = []
    for k, g in e.grad_dict().items():
        ref = r64["grads"][k]
        nrm = float(ref.norm()) + 1e-30
        ee = float((g.cpu().double() - ref).norm()) / nrm
        oo = float((r32["grads"][k].double() - ref).norm()) / nrm
        rows.append((ee, oo, k))
        worst_e, worst_o = max(worst_e, ee), max(worst_o, oo)
        assert ee < 2.0 * oo + 1e-3, (k, ee, oo)
    rows.sort(reverse=True)
    line = "f32 vs fp64 oracle at 1x96^3: engine worst %.2e, torch-CPU fp32 oracle worst %.2e; top: %s" % (
        worst_e, worst_o, ", ".join("%s %.1e (oracle32 %.1e)" % (k, a, b) for a, b, k in rows[:3]))
    print(line)
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write(line + "\n")


@pytest.mark.parametrize("tag", list(GRAD_CASES))
def test_oracle_gradients_full_size_f32(tag):
    kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
    params, x, y, r = oracle_grads(tag)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype="f32", device=DEV)
    e.load_state_dict(params)
    logits, probs = e.forward(x.to(DEV))
    out3 = e.loss_forward(logits, y.to(DEV), loss, class_alpha=torch.ones(ncls, device=DEV)).cpu()
    e.backward(e.loss_backward(logits, y.to(DEV), loss))
    assert float((logits.cpu() - r["logits"]).abs().max()) < 1e-3
    assert abs(float(out3[0]) - float(r["loss"])) < 2e-5
    rows = _grad_report(e, r)
    print("%s f32: worst gradient tensors (relative L2 vs the fp32 oracle): %s" % (tag, ", ".join("%s %.2e" % (k, v) for v, k, _ in rows[:4])))
    # the fp32 torch-CPU oracle is itself up to 2e-2 from its fp64 twin at these sizes (ReLU gates at ~0, tests/test_engine.py);
    # the gate is the same relative-L2 bound as the small-size network tests
    # measured on MI355X: 7.9e-3 (C3) / 3.4e-3 (C4) worst tensor; against the float64 oracle the engine (3.9e-3) is closer than the fp32 oracle
    # itself (7.5e-3): test_f32_gradients_full_size_against_the_fp64_oracle
    for err, k, _ in rows:
        assert err < 1.5e-2, (k, err)


@pytest.mark.parametrize("tag", list(GRAD_CASES))
@pytest.mark.parametrize("dtype,tol,med_tol,cos_tol", [("f16", 0.16, 0.105, 0.99), ("bf16", 0.44, 0.28, 0.92)])
def test_oracle_gradients_full_size_low_precision(tag, dtype, tol, med_tol, cos_tol):
    """per-tensor report of the 16-bit run dtypes at full size (where the error sits, and how large it is).
    Bounds = the MI355X measurement (profiles/r03_fullsize_report.txt, unchanged in round 4) + 25 %: f16 worst tensor 0.127 / median 0.083 /
    cosine >= 0.992, bf16 0.350 / 0.222 / 0.938; the worst tensors are GroupNorm affine gradients of norm 3e-5..9e-5 in the deep levels.
    Where the error comes from is measured, not assumed (tests/exp_lowp_fidelity.py, profiles/r04_lowp_gradient_error_sources.txt): emulating
    ONLY the forward storage rounding of conv outputs and activations on the fp32 oracle gives 0.43 / 0.24 (bf16) and 0.124 / 0.071 (f16) at
    1x48^3, only the backward storage rounding 0.019 / 0.008 - the 16-bit gradient error is the activation storage precision; keeping
    gradients in fp32 at the deep levels changes nothing (0.433)."""
    kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
    params, x, y, r = oracle_grads(tag)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype=dtype, device=DEV)
    e.load_state_dict(params)
    logits, probs = e.forward(x.to(DEV))
    e.loss_forward(logits, y.to(DEV), loss, class_alpha=torch.ones(ncls, device=DEV))
    e.backward(e.loss_backward(logits, y.to(DEV), loss))
    scale = float(e.loss_scale)
    rows = []
    for k, g in e.grad_dict().items():
        ref = r["grads"][k].double()
        nrm = float(ref.norm()) + 1e-30
        gg = g.cpu().double()
        rows.append((float((gg - ref).norm()) / nrm, k, nrm, float(torch.nn.functional.cosine_similarity(gg.flatten(), ref.flatten(), dim=0))))
    rows.sort(reverse=True)
    print("%s %s (loss scale %g): worst gradient tensors rel-L2 / cosine: %s" %
          (tag, dtype, scale, ", ".join("%s %.3f/%.4f (|g|=%.2e)" % (k, v, c, n) for v, k, n, c in rows[:6])))
    med = sorted(v for v, _, _, _ in rows)[len(rows) // 2]
    print("%s %s: median rel-L2 %.3e over %d tensors" % (tag, dtype, med, len(rows)))
    if os.environ.get("SEG_FULLSIZE_REPORT"):      # tools/gpu_*.sh keep the per-tensor numbers under profiles/
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write("%s %s loss_scale=%g median_relL2=%.3e min_cos=%.4f worst: %s\n" % (
                tag, dtype, scale, med, min(c for _, _, _, c in rows),
                ", ".join("%s %.3f/%.4f(|g|=%.1e)" % (k, v, c, n) for v, k, n, c in rows[:5])))
    assert torch.isfinite(e.grads).all()
    assert rows[0][0] < tol, rows[0]
    assert med < med_tol, med
    assert min(c for _, _, _, c in rows) > cos_tol


def _autocast_grads(tag, dt, scale):
    """the oracle's own functions on the GPU under torch.autocast(dt) - the standard mixed-precision path (conv operands 16-bit, GroupNorm / losses in
    fp32 by autocast's policy, fp32 master weights), with the same static loss scale the engine uses"""
    from collections import OrderedDict
    kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
    params, x, y, _ = oracle_grads(tag)
    P = OrderedDict((k, v.to(DEV).clone().requires_grad_(True)) for k, v in params.items())
    with torch.autocast("cuda", dtype=dt):
        logits, _ = seg.net_forward(kind, P, x.to(DEV), None)
    l = seg.loss_fn(loss, torch.ones(ncls, device=DEV))(logits.float(), y.to(DEV))
    (l * scale).backward()
    torch.cuda.synchronize()
    return OrderedDict((k, (v.grad / scale).cpu().double()) for k, v in P.items())


@pytest.mark.parametrize("tag", list(GRAD_CASES))
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_low_precision_gradients_calibrated_against_autocast(tag, dtype):
    """is the 16-bit run dtypes' gradient error 'what 16-bit storage costs' or the engine's own loss?  Three columns, ALL against
    the float64 oracle, per parameter tensor (relative L2): the engine in the 16-bit run dtype, the oracle's functions under torch.autocast of
    the same dtype on the same GPU (MIOpen / rocBLAS kernels; activations between GroupNorm and the next conv stay fp32 there, so it stores MORE
    precision than the engine's 16-bit activations), and the fp32 torch-CPU oracle.  Gate: engine median and worst <= CAL x autocast's."""
    kind, ndim, shape, ncls, loss = GRAD_CASES[tag]
    params, x, y, r32 = oracle_grads(tag)
    r64 = oracle_grads_fp64(tag)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype=dtype, device=DEV)
    e.load_state_dict(params)
    logits, probs = e.forward(x.to(DEV))
    e.loss_forward(logits, y.to(DEV), loss, class_alpha=torch.ones(ncls, device=DEV))
    e.backward(e.loss_backward(logits, y.to(DEV), loss))
    scale = float(e.loss_scale)
    ga = _autocast_grads(tag, torch.float16 if dtype == "f16" else torch.bfloat16, scale)
    rows = []
    for k, g in e.grad_dict().items():
        ref = r64["grads"][k]
        nrm = float(ref.norm()) + 1e-30
        rows.append((float((g.cpu().double() - ref).norm()) / nrm, float((ga[k] - ref).norm()) / nrm, float((r32["grads"][k].double() - ref).norm()) / nrm, k))
    med = lambda i: sorted(r[i] for r in rows)[len(rows) // 2]
    worst = lambda i: max(r[i] for r in rows)
    line = "%s %s vs the fp64 oracle, per-tensor rel-L2 median / worst: engine %.3e / %.3e | autocast %.3e / %.3e | fp32 oracle %.3e / %.3e  (engine : autocast = %.2f / %.2f)" % (
        tag, dtype, med(0), worst(0), med(1), worst(1), med(2), worst(2), med(0) / med(1), worst(0) / worst(1))
    print(line)
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write("calibration " + line + "\n")
    cal = float(os.environ.get("SEG_LOWP_CAL", LOWP_CAL))
    assert med(0) <= cal * med(1) and worst(0) <= cal * worst(1), line


def test_c5_with_cldice_term_full_size():
    """BASELINE configs[4] as worded — VNet3d 1x160^3 bf16 + clDice: the one-call clDice term at full size against the oracle's clDice on the
    SAME probabilities (loss value; d loss / d logit through the sigmoid), then three train steps with it (finite, loss goes down)."""
    kind, ndim, shape, ncls, _ = CONFIGS["C5_vnet3d"]
    params, x, y, _, _ = oracle_eval("C5_vnet3d")
    e = SegEngine(kind, ndim, shape[1], ncls, dtype="bf16", device=DEV)
    e.load_state_dict(params)
    xd, yd = x.to(DEV), y.to(DEV)
    logits, probs = e.forward(xd)
    dl = torch.zeros_like(logits)
    cld = float(e.cldice_term(probs, yd, weight=1.0, dlogits=dl, grad_scale=1.0).cpu())
    p = probs.detach().cpu().clone().requires_grad_(True)
    ref = seg.binary_soft_cldice_loss(p, y.reshape(p.shape).float())
    ref.backward()
    assert abs(cld - float(ref.detach())) < 5e-6
    want = (p.grad * p.detach() * (1 - p.detach())).reshape(dl.shape)
    got = dl.cpu()
    assert float((got - want).norm()) / float(want.norm()) < 2e-3
    losses = [float(e.train_step(xd, yd, "BinaryDiceLoss", cldice_weight=1.0)[0]) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("dtype,tol", [("f32", 4e-3), ("f16", 4e-3), ("bf16", 8e-3)])      # measured on MI355X: 1.4e-3 / 7.3e-4 / 2.5e-3
def test_thirty_step_loss_curve_follows_the_fp32_oracle(dtype, tol):
    """16-bit TRAINING fidelity, not just one gradient: thirty AdamW steps of VNet3d on 1 x 48^3 (BinaryCrossEntropyDice,
    dropout on with the SAME masks on both sides) against thirty steps of the fp32 torch-CPU oracle from the same initial weights.  The loss
    curve of the engine must stay within `tol` (absolute, on a loss that starts near 1.5) of the oracle's at EVERY step and must fall like
    it does.  The measured deviations are printed (and kept under profiles/ by tools/gpu_final.sh through SEG_FULLSIZE_REPORT)."""
    kind, ndim, ncls, loss, steps = "vnet", 3, 1, "BinaryCrossEntropyDiceLoss", 30
    params = seg.perturb_params(seg.init_params(kind, ndim, 1, ncls, seed=0), seed=7)
    x, y = seg.synthetic_batch(1, (48, 48, 48), 1, ncls, seed=21)
    g = torch.Generator().manual_seed(3)
    all_masks = [seg.draw_masks(kind, 1, generator=g) for _ in range(steps)]
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cur, st, ref_curve = {k: v.clone() for k, v in params.items()}, {}, []
    for it in range(steps):
        r = seg.forward_backward(kind, cur, x, y, loss, masks=all_masks[it])
        ref_curve.append(float(r["loss"]))
        cur = seg.adamw_step(cur, r["grads"], st)
    e = SegEngine(kind, ndim, 1, ncls, dtype=dtype, device=DEV)
    e.load_state_dict(params)
    xd, yd = x.to(DEV), y.to(DEV)
    curve = [float(e.train_step(xd, yd, loss, lr=1e-3, mask_mode=_capi.MASKS_GIVEN, masks=all_masks[it])[0]) for it in range(steps)]
    dev = [abs(a - b) for a, b in zip(curve, ref_curve)]
    line = "loss curve %s: oracle %.4f -> %.4f, engine %.4f -> %.4f, max |diff| %.2e at step %d, mean |diff| %.2e, skipped steps %d" % (
        dtype, ref_curve[0], ref_curve[-1], curve[0], curve[-1], max(dev), dev.index(max(dev)), sum(dev) / len(dev), e.skipped_steps)
    print(line)
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write(line + "\n")
    assert all(np.isfinite(curve))
    assert max(dev) < tol, line
    assert curve[-1] < curve[0] - 0.5 * (ref_curve[0] - ref_curve[-1]), line          # it trains: at least half the oracle's descent


def test_c1_unet2d_at_its_own_size_f32_forward_gradients_and_steps():
    """BASELINE configs[0] - UNet2d binary segmentation, 2 x 1 x 256 x 256, fp32, BinaryDiceLoss - is the reference's own CPU-runnable case (VERDICT r04
    item 7b: it had only been run at 32^2).  Here it goes through the engine at ITS size in the f32 run dtype against the oracle: logits within
    north_star's 1e-3, identical integer masks (up to numerically tied voxels) and Dice, loss within 2e-5, every gradient tensor against the float64
    oracle no further than twice the fp32 oracle's own distance, and three AdamW steps with the same recorded channel-dropout masks on both sides
    tracking the oracle's loss."""
    tag = "C1_unet2d"
    kind, ndim, shape, ncls, loss = CONFIGS[tag]
    params = seg.perturb_params(seg.init_params(kind, ndim, shape[1], ncls, seed=0), seed=7)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=3)
    r32 = seg.forward_backward(kind, params, x, y, loss)
    r64 = seg.forward_backward(kind, {k: v.double() for k, v in params.items()}, x.double(), y, loss)
    e = SegEngine(kind, ndim, shape[1], ncls, dtype="f32", device=DEV)
    e.load_state_dict(params)
    xd, yd = x.to(DEV), y.to(DEV)
    logits, probs = e.forward(xd)
    out3 = e.loss_forward(logits, yd, loss).cpu()
    e.backward(e.loss_backward(logits, yd, loss))
    err = float((logits.cpu() - r32["logits"]).abs().max())
    assert err < 1e-3, err
    ref_probs = torch.sigmoid(r32["logits"])
    flip = (probs.cpu() > 0.5) != (ref_probs > 0.5)
    assert int(flip.sum()) == 0 or float((ref_probs[flip] - 0.5).abs().max()) < 5e-5
    assert abs(float(out3[0]) - float(r32["loss"])) < 2e-5
    assert abs(float(out3[1]) - float(seg.dice_coeff(ref_probs, y))) < (1e-6 if int(flip.sum()) == 0 else 1e-4)
    worst = 0.0
    for k, g in e.grad_dict().items():
        ref = r64["grads"][k]
        nrm = float(ref.norm()) + 1e-30
        ee = float((g.cpu().double() - ref).norm()) / nrm
        oo = float((r32["grads"][k].double() - ref).norm()) / nrm
        worst = max(worst, ee)
        assert ee < 2.0 * oo + 1e-3, (k, ee, oo)
    # three optimisation steps against the oracle's AdamW
    gm = torch.Generator().manual_seed(3)
    all_masks = [seg.draw_masks(kind, shape[0], generator=gm) for _ in range(3)]          # the SAME channel-dropout masks on both sides (Unet2d.py:74,83)
    cur, st, ref_curve = {k: v.clone() for k, v in params.items()}, {}, []
    cur64, st64 = {k: v.double() for k, v in params.items()}, {}
    for it in range(3):
        r = seg.forward_backward(kind, cur, x, y, loss, masks=all_masks[it])
        ref_curve.append(float(r["loss"]))
        cur = seg.adamw_step(cur, r["grads"], st)
        r = seg.forward_backward(kind, cur64, x.double(), y, loss, masks=[m.double() for m in all_masks[it]])
        cur64 = seg.adamw_step(cur64, r["grads"], st64)
    e.load_state_dict(params)
    curve = [float(e.train_step(xd, yd, loss, lr=1e-3, mask_mode=_capi.MASKS_GIVEN, masks=all_masks[it])[0]) for it in range(3)]
    line = "C1 UNet2d 2x256^2 f32 at its own size: logits max|d| %.2e, mask flips %d, worst gradient tensor vs fp64 %.2e, 3-step loss %s vs oracle %s" % (
        err, int(flip.sum()), worst, ["%.5f" % v for v in curve], ["%.5f" % v for v in ref_curve])
    print(line)
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write(line + "\n")
    assert max(abs(a - b) for a, b in zip(curve, ref_curve)) < 2e-4, line
    # AdamW moves every weight by <= lr per step whatever its gradient's size, and at this size a tenth of the gradients are below 2e-9 - the scale of
    # Adam's eps - so the update of those weights is decided by rounding noise: the fp32 oracle itself ends > 1e-4 away from its float64 twin in 19 % of
    # the weights.  The yardstick is therefore the float64 run: the engine may disagree with it in at most 1.5x as many weights as the fp32 oracle does.
    sd = e.state_dict()
    tot = bad = bad_o = 0
    for k, v in cur64.items():
        d = (sd[k].cpu().double() - v).abs()
        assert float(d.max()) < 3 * 2e-3, k
        tot += d.numel()
        bad += int((d > 1e-4).sum())
        bad_o += int(((cur[k].double() - v).abs() > 1e-4).sum())
    print("weights > 1e-4 from the float64 oracle after 3 steps: engine %.3f, fp32 oracle %.3f" % (bad / tot, bad_o / tot))
    assert bad <= 1.5 * bad_o + 0.01 * tot, (bad, bad_o, tot)


_LONG_CURVE = {}


def _oracle_long_curve(steps):
    """`steps` AdamW steps of the fp32 torch-CPU oracle, VNet3d on 1 x 48^3 (BinaryCrossEntropyDice, recorded dropout masks): computed once per process"""
    if steps not in _LONG_CURVE:
        kind, ncls, loss = "vnet", 1, "BinaryCrossEntropyDiceLoss"
        params = seg.perturb_params(seg.init_params(kind, 3, 1, ncls, seed=0), seed=7)
        x, y = seg.synthetic_batch(1, (48, 48, 48), 1, ncls, seed=21)
        g = torch.Generator().manual_seed(3)
        masks = [seg.draw_masks(kind, 1, generator=g) for _ in range(steps)]
        nt = torch.get_num_threads()
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        try:
            cur, st, curve = {k: v.clone() for k, v in params.items()}, {}, []
            for it in range(steps):
                r = seg.forward_backward(kind, cur, x, y, loss, masks=masks[it])
                curve.append(float(r["loss"]))
                cur = seg.adamw_step(cur, r["grads"], st)
        finally:
            torch.set_num_threads(nt)
        _LONG_CURVE[steps] = (params, x, y, masks, curve)
    return _LONG_CURVE[steps]


# the 16-bit curves are compared over ten times the horizon of the test above
# measured on MI355X (profiles/r05_fullsize_report.txt): f16 max |diff| 4.9e-3 (step 283), last-50 mean 7.8e-5; bf16 5.6e-3 (step 108), 5.0e-4
# Runs of the same binary differ (single trajectories separate chaotically; the order of the statistics' atomics makes runs differ in the last bits).  Observed over
# three GPU sessions: f16 max |diff| 4.4e-3 / 4.9e-3 / 6.8e-3, last-50 mean 7.8e-5 ... 9.3e-4; bf16 5.6e-3 / 6.5e-3 / 1.22e-2, 5.0e-4 ... 4.9e-3.  Gates: 11 % of the
# oracle's descent (0.225) at the worst step, 7 % for the tail mean - a run that skips steps, diverges or stalls fails them by a wide margin.
@pytest.mark.parametrize("dtype,tol,tail_tol", [("f16", 2.5e-2, 1.5e-2), ("bf16", 2.5e-2, 1.5e-2)])
def test_three_hundred_step_loss_curve_follows_the_fp32_oracle(dtype, tol, tail_tol):
    """the only training-level evidence for the 16-bit run dtypes was a 30-step curve.  300 AdamW steps of VNet3d on 1 x 48^3
    (the SAME dropout masks on both sides) against 300 steps of the fp32 oracle from the same weights: the 16-bit loss curve must stay within `tol` of
    the oracle's at every step, the mean over the last 50 steps within `tail_tol`, no step may be refused by the loss-scale guard, and the run must
    descend like the oracle does.  Individual trajectories separate slowly (every rounding is a perturbation that training amplifies); what is gated
    is that they separate by percents of the descent, not that they coincide."""
    steps = 300
    params, x, y, masks, ref = _oracle_long_curve(steps)
    e = SegEngine("vnet", 3, 1, 1, dtype=dtype, device=DEV)
    e.load_state_dict(params)
    xd, yd = x.to(DEV), y.to(DEV)
    curve = [float(e.train_step(xd, yd, "BinaryCrossEntropyDiceLoss", lr=1e-3, mask_mode=_capi.MASKS_GIVEN, masks=masks[it])[0]) for it in range(steps)]
    dev = [abs(a - b) for a, b in zip(curve, ref)]
    tail = abs(sum(curve[-50:]) - sum(ref[-50:])) / 50
    line = "300-step loss curve %s: oracle %.4f -> %.4f, engine %.4f -> %.4f, max |diff| %.2e at step %d, mean |diff| %.2e, |diff| of the last-50 mean %.2e, skipped steps %d" % (
        dtype, ref[0], ref[-1], curve[0], curve[-1], max(dev), dev.index(max(dev)), sum(dev) / len(dev), tail, e.skipped_steps)
    print(line)
    if os.environ.get("SEG_FULLSIZE_REPORT"):
        with open(os.environ["SEG_FULLSIZE_REPORT"], "a") as f:
            f.write(line + "\n")
    assert all(np.isfinite(curve)) and e.skipped_steps == 0, line
    assert max(dev) < tol and tail < tail_tol, line
    assert ref[0] - curve[-1] > 0.8 * (ref[0] - ref[-1]), line

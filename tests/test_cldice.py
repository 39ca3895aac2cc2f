"""soft-clDice: oracle vs the repaired-reference golden vectors (CPU), HIP kernels vs the oracle
(checker build on CPU, libsegengine on the GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import seg_oracle as seg
from pytorchdeeplearing_amd import lossescldice as cl

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cldice.npz"))


def T(name, device="cpu"):
    return torch.from_numpy(G[name]).to(device)


# ---- the oracle is pinned to the reference (with the documented repairs) -----------------------------------------
@pytest.mark.parametrize("tag", ["b3", "b2"])
def test_oracle_binary_matches_reference(tag):
    pred, target = T(tag + "_pred").requires_grad_(True), T(tag + "_target")
    assert torch.equal(seg.soft_skeletonize(pred.detach()), T(tag + "_skel_pred"))
    assert torch.equal(seg.soft_skeletonize(target), T(tag + "_skel_target"))
    loss = seg.binary_soft_cldice_loss(pred, target)
    loss.backward()
    assert abs(float(loss.detach()) - float(G[tag + "_loss"])) < 1e-7
    np.testing.assert_allclose(pred.grad.numpy(), G[tag + "_dpred"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("tag", ["m3", "m2"])
def test_oracle_multiclass_matches_reference(tag):
    pred, target = T(tag + "_pred").requires_grad_(True), T(tag + "_target")
    loss = seg.multi_soft_cldice_loss(pred, target, G[tag + "_alpha"])
    loss.backward()
    assert abs(float(loss.detach()) - float(G[tag + "_loss"])) < 1e-7
    np.testing.assert_allclose(pred.grad.numpy(), G[tag + "_dpred"], rtol=1e-6, atol=1e-9)


# ---- HIP path vs oracle / golden ------------------------------------------------------------------------------------
def _binary_case(tag, dev):
    pred, target = T(tag + "_pred", dev).requires_grad_(True), T(tag + "_target", dev)
    # forward skeletons are pure min/max/sub/relu in fp32: bit-exact
    assert torch.equal(cl.soft_skeletonize(pred.detach()).cpu(), T(tag + "_skel_pred"))
    assert torch.equal(cl.soft_skeletonize(target).cpu(), T(tag + "_skel_target"))
    loss = cl.Binary_Soft_cldice_loss()(pred, target)
    loss.backward()
    assert abs(float(loss.detach()) - float(G[tag + "_loss"])) < 2e-6
    # gradient routing sums +-2e-4 contributions with fp32 atomics (order varies run to run): elements that cancel to
    # ~1e-9 carry an absolute error of a few 1e-8, hence the absolute term tied to the gradient scale
    ref = G[tag + "_dpred"]
    np.testing.assert_allclose(pred.grad.cpu().numpy(), ref, rtol=2e-4, atol=3e-4 * float(np.abs(ref).max()))


def _multi_case(tag, dev):
    pred, target = T(tag + "_pred", dev).requires_grad_(True), T(tag + "_target", dev)
    loss = cl.Mutil_Soft_cldice_loss(torch.from_numpy(G[tag + "_alpha"]))(pred, target)
    loss.backward()
    assert abs(float(loss.detach()) - float(G[tag + "_loss"])) < 2e-6
    ref = G[tag + "_dpred"]
    np.testing.assert_allclose(pred.grad.cpu().numpy(), ref, rtol=2e-4, atol=3e-4 * float(np.abs(ref).max()))


@pytest.mark.parametrize("tag", ["b3", "b2"])
def test_cldice_binary(dev, tag):
    _binary_case(tag, dev)


@pytest.mark.parametrize("tag", ["m3", "m2"])
def test_cldice_multiclass(dev, tag):
    _multi_case(tag, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 1, 40, 48, 56), (3, 2, 96, 80)])
def test_cldice_vs_oracle_gpu(shape):
    """larger seeded case against the oracle itself (same inputs, oracle on the host)."""
    from oracle.make_golden import cldice_inputs
    pred, target = cldice_inputs(shape, 5)
    p0 = pred.clone().requires_grad_(True)
    l0 = seg.binary_soft_cldice_loss(p0, target)
    l0.backward()
    p1 = pred.cuda().requires_grad_(True)
    assert torch.equal(cl.soft_skeletonize(p1.detach()).cpu(), seg.soft_skeletonize(pred))
    l1 = cl.Binary_Soft_cldice_loss()(p1, target.cuda())
    l1.backward()
    assert abs(float(l1) - float(l0)) < 2e-6
    np.testing.assert_allclose(p1.grad.cpu().numpy(), p0.grad.numpy(), rtol=5e-4, atol=3e-4 * float(p0.grad.abs().max()))


def _first_class(pred, target, nd, dev, scale=1.0):
    """seg_cldice_binary on (planes = n*c) probabilities: returns (loss, d loss / d logit) for logit = logit(pred)."""
    from pytorchdeeplearing_amd import _capi
    lib, st = _capi.lib_for(dev), _capi.stream_for(dev)
    p = pred.detach().float().contiguous()
    planes = p.shape[0] * p.shape[1]
    sp = tuple(p.shape[2:])
    d, h, w = sp if nd == 3 else (1,) + sp
    t = target.to(torch.float32).contiguous()
    ws = torch.empty(lib.seg_cldice_ws_bytes(planes, d, h, w, nd, 10) // 4 + 64, dtype=torch.float32, device=p.device)
    out = torch.zeros(1, dtype=torch.float32, device=p.device)
    dl = torch.full_like(p, 0.25)                       # the entry point ADDS to what the companion loss left there
    lib.check(lib.seg_cldice_binary(p.data_ptr(), t.data_ptr(), _capi.LABEL_TYPES[str(t.dtype)], planes, d, h, w, nd, 10, float(scale),
                                    ws.data_ptr(), out.data_ptr(), dl.data_ptr(), 0, st), "seg_cldice_binary")
    return float(out.cpu()), (dl - 0.25).cpu()


@pytest.mark.parametrize("tag", ["b3", "b2"])
def test_cldice_first_class_call_matches_reference_golden(dev, tag):
    """the one-call loss of the engine (both skeletons, ratios, backward to the logits) against the repaired-reference golden values"""
    pred, target = T(tag + "_pred", dev), T(tag + "_target", dev)
    loss, dlogit = _first_class(pred, target, pred.dim() - 2, dev, scale=3.0)
    assert abs(loss - float(G[tag + "_loss"])) < 2e-6
    pr = T(tag + "_pred")
    ref = 3.0 * torch.from_numpy(G[tag + "_dpred"]) * pr * (1.0 - pr)
    np.testing.assert_allclose(dlogit.numpy(), ref.numpy(), rtol=3e-4, atol=3e-4 * float(ref.abs().max()))


def test_engine_train_step_with_cldice_term(dev):
    """SegEngine.train_step(cldice_weight=...) = Dice + clDice on the head's probabilities: loss value and d logits equal the sum of the
    fused Dice path and the autograd clDice module on the same probabilities"""
    from pytorchdeeplearing_amd import SegEngine
    nd, sp = (3, (16, 16, 16)) if dev.type == "cuda" else (2, (16, 16))       # the host checker runs the 2-D network (a minute less)
    e = SegEngine("vnet", nd, 1, 1, dtype="f32", device=dev)
    e.load_state_dict(seg.perturb_params(seg.init_params("vnet", nd, 1, 1, seed=0), seed=3))
    x, y = seg.synthetic_batch(1, sp, 1, 1, seed=5)
    x, y = x.to(dev), y.to(dev)
    from pytorchdeeplearing_amd import _capi
    logits, probs = e.forward(x, _capi.MASKS_EVAL)
    out3 = e.loss_forward(logits, y, "BinaryDiceLoss").clone()
    dl = e.loss_backward(logits, y, "BinaryDiceLoss").clone()
    dl2 = dl.clone()
    cld = e.cldice_term(probs, y, weight=0.5, dlogits=dl2).clone()
    p = probs.detach().clone().requires_grad_(True)
    ref = cl.Binary_Soft_cldice_loss()(p, y.reshape(p.shape).float())
    ref.backward()
    assert abs(float(cld) - float(ref.detach())) < 2e-6
    want = dl + 0.5 * float(e.loss_scale) * p.grad * probs * (1 - probs)
    np.testing.assert_allclose(dl2.cpu().numpy(), want.cpu().numpy(), rtol=3e-4, atol=3e-4 * float(want.abs().max()))
    o = e.train_step(x, y, "BinaryDiceLoss", cldice_weight=0.5, mask_mode=_capi.MASKS_EVAL).cpu()
    assert abs(float(o[0]) - (float(out3[0]) + 0.5 * float(cld))) < 5e-6


def test_cldice_cpu_tensors_raise():
    from pytorchdeeplearing_amd import _capi
    saved = _capi.lib_for
    _capi.lib_for = getattr(_capi, "product_lib_for", saved)          # the package's own routing (the test suite's checker routing put aside)
    try:
        with pytest.raises(RuntimeError):
            cl.soft_skeletonize(torch.rand(1, 1, 8, 8))
    finally:
        _capi.lib_for = saved


@pytest.mark.parametrize("shape,nd", [((2, 5, 9, 11), 3), ((3, 4, 10, 13), 2), ((1, 6, 20, 40), 3)])
def test_skel_iter_kernels_agree_with_two_kernel_forms(dev, shape, nd):
    """the fused LDS-tile iteration and its gather backward equal the simple per-voxel kernels (pool3 + skel_update forward,
    atomic scatter backward) on random and on tie-rich data."""
    from pytorchdeeplearing_amd import _capi
    lib, st = _capi.lib_for(dev), _capi.stream_for(dev)
    planes, d, h, w = shape
    g0 = torch.Generator().manual_seed(11)
    for ties in (False, True):
        x = torch.rand(shape, generator=g0)
        if ties:
            x = (x * 4).round() / 4                       # many equal values: exercises the first-extremum rule
        x = x.to(dev).contiguous()
        gup = torch.rand(shape, generator=g0).to(dev).contiguous()
        e1, y1, e2, y2 = (torch.empty_like(x) for _ in range(4))
        lib.check(lib.seg_op_pool3(x.data_ptr(), e1.data_ptr(), planes, d, h, w, nd, 1, st), "pool3")
        lib.check(lib.seg_op_skel_update(x.data_ptr(), e1.data_ptr(), y1.data_ptr(), planes, d, h, w, nd, st), "update")
        lib.check(lib.seg_op_skel_iter(x.data_ptr(), e2.data_ptr(), y2.data_ptr(), planes, d, h, w, nd, st), "iter")
        assert torch.equal(e1, e2) and torch.equal(y1, y2)
        dx1, de1 = torch.empty_like(x), torch.zeros_like(x)
        lib.check(lib.seg_op_skel_update_bwd(gup.data_ptr(), x.data_ptr(), e1.data_ptr(), dx1.data_ptr(), de1.data_ptr(), planes, d, h, w, nd, st), "a")
        lib.check(lib.seg_op_pool3_bwd(x.data_ptr(), de1.data_ptr(), dx1.data_ptr(), planes, d, h, w, nd, 1, st), "b")
        dx2, de2 = torch.empty_like(x), torch.empty_like(x)
        lib.check(lib.seg_op_skel_iter_bwd(gup.data_ptr(), x.data_ptr(), e1.data_ptr(), dx2.data_ptr(), de2.data_ptr(), planes, d, h, w, nd, st), "bwd")
        assert float((de1 - de2).abs().max()) <= 1e-5 and float((dx1 - dx2).abs().max()) <= 1e-5


@pytest.mark.parametrize("shape,nd,width", [((1, 5, 9, 33), 3, 10), ((2, 4, 6, 64), 3, 3), ((1, 3, 7, 65), 3, 10), ((1, 6, 12, 31), 3, 4),
                                            ((2, 3, 10, 70), 2, 10), ((1, 1, 16, 160), 2, 2),
                                            # volumes whose three bit images do not fit the fp32 scratch volume (rows of <= 2 voxels, a few hundred
                                            # voxels) take the fp32 iteration instead of spilling into the neighbouring volumes
                                            ((2, 4, 4, 4), 3, 3), ((1, 2, 8, 8), 3, 3), ((3, 5, 6, 2), 3, 2), ((1, 1, 9, 1), 2, 2)])
def test_target_skeleton_on_bits_equals_the_fp32_iteration(dev, shape, nd, width):
    """round 5: the label-only half of the one-call clDice term (seg_cldice_target) computes the skeleton of the BINARY mask (label != 0) on a bit
    image - erosion / dilation as AND / OR of shifted words, 32 voxels per word - and expands it to fp32 once.  It must equal, bit for bit, `width`
    iterations of the fp32 tile kernel (seg_op_skel_iter) on the float mask, for row lengths that end inside a word, on a word boundary and one
    past it, for 0/1 and 0/255 labels and every label type, in 3-D and per-slice 2-D pooling.  (White box: the workspace layout of
    csrc/cldice.hip:cld_layout - y first, then 2 * width work volumes, then the three target volumes.)"""
    from pytorchdeeplearing_amd import _capi
    lib, st = _capi.lib_for(dev), _capi.stream_for(dev)
    planes, d, h, w = shape
    if nd == 2:                                   # the one-call entry points take 2-D problems as (planes, 1, h, w): every slice is a plane
        planes, d = planes * d, 1
        shape = (planes, d, h, w)
    g0 = torch.Generator().manual_seed(5)
    # blobs, so that the skeleton is neither empty nor everything
    blob = torch.nn.functional.avg_pool3d(torch.rand((1, planes, d + 4, h + 4, w + 4), generator=g0), 5, stride=1)[0]
    mask = (blob > blob.median()).to(torch.uint8)
    n = planes * d * h * w
    vol = (n * 4 + 255) // 256 * 256
    for labels in (mask, mask * 255, mask.to(torch.int64) * 7, mask.to(torch.int32), mask.float() * 2.0):
        t = labels.contiguous().to(dev)
        ws = torch.zeros(lib.seg_cldice_ws_bytes(planes, d, h, w, nd, width) // 4 + 64, dtype=torch.float32, device=dev)
        lib.check(lib.seg_cldice_target(t.data_ptr(), _capi.LABEL_TYPES[str(t.dtype)], planes, d, h, w, nd, width, ws.data_ptr(), st), "seg_cldice_target")
        y = ws[:n].reshape(shape)
        off = (vol + 2 * width * vol + (1 + ((width - 1) & 1)) * vol) // 4
        got = ws[off:off + n].reshape(shape)
        cur = (labels != 0).float().to(dev).contiguous()
        assert torch.equal(y, cur)                                  # the float labels the sums use are the binarised ones
        e = torch.empty_like(cur)
        for _ in range(width):
            nx = torch.empty_like(cur)
            lib.check(lib.seg_op_skel_iter(cur.data_ptr(), e.data_ptr(), nx.data_ptr(), planes, d, h, w, nd, st), "iter")
            cur = nx
        assert torch.equal(got, cur)
        assert 0 < int(cur.sum()) < n or w <= 2

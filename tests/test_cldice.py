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


def test_cldice_cpu_tensors_raise():
    from pytorchdeeplearing_amd import _capi
    if _capi._injected is not None:
        pytest.skip("checker library injected in this process")
    with pytest.raises(RuntimeError):
        cl.soft_skeletonize(torch.rand(1, 1, 8, 8))


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

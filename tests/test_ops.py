"""Operator-level parity: each HIP kernel (implicit-GEMM conv in gather / scatter mode, data-gradient
forms, weight gradients) against torch.nn.functional on integer-valued data, where f32, f16 and bf16
arithmetic is exact -> bit-exact comparison for all three run dtypes."""
import pytest

import conftest
import torch
import torch.nn.functional as F

from pytorchdeeplearing_amd import _capi, ops

DT = ["f32", "f16", "bf16"]


def cl(x):      # NC[D]HW -> N D H W C (2-D: D = 1)
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x, ndim):
    x = x.permute(0, 4, 1, 2, 3).contiguous()
    return x.squeeze(2) if ndim == 2 else x


def ints(shape, lo, hi, g, density=1.0):
    t = torch.randint(lo, hi + 1, shape, generator=g).float()
    if density < 1.0:
        t = t * (torch.rand(shape, generator=g) < density).float()
    return t


def to_dev(t, dtype, dev):
    return ops.aligned_like(t.to(dev).to(ops.TORCH_DTYPE[dtype]))


def test_abi_struct_sizes(dev):
    ops.check_abi(_capi.lib_for(dev))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [
    # ndim, N, spatial, Cin(list = concat), Cout, k, stride, pad
    (3, 2, (6, 5, 7), [16], 16, 3, 1, 1),
    (3, 1, (4, 6, 8), [32], 64, 3, 1, 1),
    (3, 3, (4, 4, 4), [16], 32, 2, 2, 0),
    (3, 2, (5, 4, 6), [16, 16], 16, 1, 1, 0),
    (2, 2, (9, 11), [16], 16, 3, 1, 1),
    (2, 1, (8, 12), [64, 64], 64, 1, 1, 0),
    (3, 1, (3, 4, 5), [64], 128, 3, 1, 1),
    # short reductions on 16-row-aligned volumes -> register-resident streaming kernel (last field: expected kernel)
    (3, 2, (4, 4, 8), [16], 32, 2, 2, 0, 1),
    (3, 2, (2, 4, 6), [16, 16], 16, 1, 1, 0, 1),
    (3, 1, (2, 2, 4), [16], 16, 1, 1, 0, 1),
    (2, 2, (8, 8), [16], 32, 2, 2, 0, 1),
    (2, 1, (4, 8), [32, 32], 16, 1, 1, 0, 1),
    (2, 3, (8, 16), [32], 64, 2, 2, 0, 0),
])
def test_conv_gather_exact(dev, dtype, case):
    want_kernel = case[8] if len(case) > 8 else 0
    ndim, N, sp, cins, cout, k, stride, pad = case[:8]
    g = torch.Generator().manual_seed(sum(sp) * 7 + cout)
    cin = sum(cins)
    x = ints((N, cin) + sp, -2, 2, g)
    w = ints((cout, cin) + (k,) * ndim, -1, 1, g, density=0.2)
    b = ints((cout,), -3, 3, g)
    conv = F.conv3d if ndim == 3 else F.conv2d
    ref = conv(x, w, b, stride=stride, padding=pad)
    assert float(ref.abs().max()) <= 256
    xs = torch.split(x, cins, dim=1)
    x0 = to_dev(cl(xs[0]), dtype, dev)
    x1 = to_dev(cl(xs[1]), dtype, dev) if len(xs) > 1 else None
    wp = ops.pack(w.to(dev), "conv_fwd", dtype)
    out, stats = ops.conv(x0, wp, dtype, ndim, k, stride, pad, x1=x1, bias=ops.aligned_like(b.to(dev)), cout=cout, want_stats=True)
    assert ops.last_conv_kernel == want_kernel
    got = ncdhw(out.float().cpu(), ndim)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    rs = torch.stack([ref.double().flatten(2).sum(2), (ref.double() ** 2).flatten(2).sum(2)], dim=2)
    assert torch.equal(stats.cpu(), rs)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(3, 2, (3, 4, 5), 32, 16), (2, 2, (6, 7), 16, 16), (3, 1, (2, 2, 3), 128, 64),
                                  (3, 2, (2, 2, 4), 32, 16, 1), (2, 2, (4, 4), 32, 16, 1), (2, 1, (4, 8), 64, 16, 1), (2, 1, (4, 8), 64, 32, 0)])
def test_conv_transpose_scatter_exact(dev, dtype, case):
    want_kernel = case[5] if len(case) > 5 else 0
    ndim, N, sp, cin, cout = case[:5]
    g = torch.Generator().manual_seed(7)
    x = ints((N, cin) + sp, -2, 2, g)
    w = ints((cin, cout) + (2,) * ndim, -1, 1, g, density=0.3)
    b = ints((cout,), -3, 3, g)
    convT = F.conv_transpose3d if ndim == 3 else F.conv_transpose2d
    ref = convT(x, w, b, stride=2)
    assert float(ref.abs().max()) <= 256
    wp = ops.pack(w.to(dev), "convT_fwd", dtype)
    out, stats = ops.conv(to_dev(cl(x), dtype, dev), wp, dtype, ndim, 2, scatter=True, bias=ops.aligned_like(b.to(dev)),
                          cout=cout, want_stats=True)
    assert ops.last_conv_kernel == want_kernel
    assert torch.equal(ncdhw(out.float().cpu(), ndim), ref)
    rs = torch.stack([ref.double().flatten(2).sum(2), (ref.double() ** 2).flatten(2).sum(2)], dim=2)
    assert torch.equal(stats.cpu(), rs)


@pytest.mark.parametrize("dtype", DT)
def test_data_gradients_exact(dev, dtype):
    """dgrad of conv k3 (flipped gather), of conv k2s2 (scatter) and of convT k2s2 (strided gather)."""
    g = torch.Generator().manual_seed(11)
    # conv 3^3, 16 -> 32
    x = ints((2, 16, 4, 5, 6), -2, 2, g).requires_grad_(True)
    w = ints((32, 16, 3, 3, 3), -1, 1, g, density=0.15)
    dy = ints((2, 32, 4, 5, 6), -1, 1, g, density=0.5)
    F.conv3d(x, w, padding=1).backward(dy)
    assert float(x.grad.abs().max()) <= 256
    wp = ops.pack(w.to(dev), "conv_dgrad", dtype)
    got = ops.conv(to_dev(cl(dy), dtype, dev), wp, dtype, 3, 3, 1, 1, cout=16)
    assert torch.equal(ncdhw(got.float().cpu(), 3), x.grad)
    # conv 2^3 stride 2, 16 -> 32 : gradient is a scatter GEMM
    x = ints((2, 16, 4, 6, 4), -2, 2, g).requires_grad_(True)
    w = ints((32, 16, 2, 2, 2), -1, 1, g, density=0.3)
    dy = ints((2, 32, 2, 3, 2), -1, 1, g)
    F.conv3d(x, w, stride=2).backward(dy)
    wp = ops.pack(w.to(dev), "k2s2_dgrad", dtype)
    got = ops.conv(to_dev(cl(dy), dtype, dev), wp, dtype, 3, 2, scatter=True, cout=16)
    assert torch.equal(ncdhw(got.float().cpu(), 3), x.grad)
    # conv-transpose 2^3 stride 2, 32 -> 16 : gradient is a stride-2 gather
    x = ints((1, 32, 2, 3, 4), -2, 2, g).requires_grad_(True)
    w = ints((32, 16, 2, 2, 2), -1, 1, g, density=0.3)
    dy = ints((1, 16, 4, 6, 8), -1, 1, g)
    F.conv_transpose3d(x, w, stride=2).backward(dy)
    wp = ops.pack(w.to(dev), "convT_dgrad", dtype)
    got = ops.conv(to_dev(cl(dy), dtype, dev), wp, dtype, 3, 2, 2, 0, cout=32)
    assert torch.equal(ncdhw(got.float().cpu(), 3), x.grad)
    # the same two forms on 16-row-aligned volumes (streaming kernel)
    x = ints((2, 16, 4, 4, 8), -2, 2, g).requires_grad_(True)
    w = ints((32, 16, 2, 2, 2), -1, 1, g, density=0.3)
    dy = ints((2, 32, 2, 2, 4), -1, 1, g)
    F.conv3d(x, w, stride=2).backward(dy)
    wp = ops.pack(w.to(dev), "k2s2_dgrad", dtype)
    got = ops.conv(to_dev(cl(dy), dtype, dev), wp, dtype, 3, 2, scatter=True, cout=16)
    assert ops.last_conv_kernel == 1
    assert torch.equal(ncdhw(got.float().cpu(), 3), x.grad)
    x = ints((1, 32, 2, 2, 4), -2, 2, g).requires_grad_(True)
    w = ints((32, 16, 2, 2, 2), -1, 1, g, density=0.3)
    dy = ints((1, 16, 4, 4, 8), -1, 1, g)
    F.conv_transpose3d(x, w, stride=2).backward(dy)
    wp = ops.pack(w.to(dev), "convT_dgrad", dtype)
    got = ops.conv(to_dev(cl(dy), dtype, dev), wp, dtype, 3, 2, 2, 0, cout=32)
    assert ops.last_conv_kernel == 1
    assert torch.equal(ncdhw(got.float().cpu(), 3), x.grad)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [
    # ndim, N, spatial(in), Cin list, Cout, k, stride, pad
    (3, 2, (5, 4, 6), [16], 16, 3, 1, 1),
    (3, 1, (4, 4, 6), [32], 64, 3, 1, 1),
    (3, 2, (4, 6, 4), [16], 32, 2, 2, 0),
    (3, 2, (3, 5, 4), [16, 16], 16, 1, 1, 0),
    # 1^d convolutions on a concat: the multi-step streaming kernel of the 16-bit dtypes (wgrad_direct_kernel: 16 x 32, 32 x 64 and 64 x 64 tiles,
    # several 128-row steps per slice, a ragged last step)
    (3, 2, (6, 9, 8), [16, 16], 16, 1, 1, 0),
    (3, 1, (5, 8, 9), [32, 32], 32, 1, 1, 0),
    (2, 2, (13, 16), [64, 64], 64, 1, 1, 0),
    (3, 1, (96, 96, 96), [16, 16], 16, 1, 1, 0),      # seven steps per slice: the unrolled register ring wraps around (GPU run only)
    (2, 3, (7, 9), [16], 32, 3, 1, 1),
    (3, 1, (2, 3, 4), [128], 128, 3, 1, 1),
])
def test_wgrad_exact(dev, dtype, case):
    ndim, N, sp, cins, cout, k, stride, pad = case
    if sp[0] >= 96:
        conftest.checker_slow(dev, "885 k voxel rows on the host checker")
    g = torch.Generator().manual_seed(5)
    cin = sum(cins)
    x = ints((N, cin) + sp, -2, 2, g)
    w = torch.zeros((cout, cin) + (k,) * ndim, requires_grad=True)
    conv = F.conv3d if ndim == 3 else F.conv2d
    y = conv(x, w, stride=stride, padding=pad)
    dy = ints(tuple(y.shape), -2, 2, g)
    y.backward(dy)
    xs = torch.split(x, cins, dim=1)
    x1 = to_dev(cl(xs[1]), dtype, dev) if len(xs) > 1 else None
    got = ops.wgrad(to_dev(cl(dy), dtype, dev), to_dev(cl(xs[0]), dtype, dev), dtype, ndim, k, stride, pad, x1=x1)
    assert torch.equal(got.cpu(), w.grad), float((got.cpu() - w.grad).abs().max())


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", [(3, 2, (2, 4, 6), [16, 16], 16), (3, 3, (4, 4, 8), [32, 32], 32), (2, 2, (8, 16), [16, 16], 16),
                                  pytest.param((3, 2, (48, 48, 48), [32, 32], 32), marks=pytest.mark.gpu)])
def test_activation_on_load_exact(dev, dtype, case):
    """seg_conv_args / seg_wgrad_args act_scale, act_shift: the first concat source holds the RAW output r of a conv + GroupNorm unit and is read as
    relu(scale[n][c] * r + shift[n][c]) rounded to the run dtype (GroupNorm + channel dropout + ReLU of networks/VNet3d.py:72-74 applied by the reader).  Integer
    r, power-of-two scales (zero = a dropped channel) and small integer shifts: every product and sum is exact, so the 1^d conv and its weight gradient must
    equal torch on the activated tensor bit for bit.  More than one sample: a voxel slice of the weight gradient may span a sample boundary."""
    ndim, N, sp, cins, cout = case
    g = torch.Generator().manual_seed(sum(sp) + cout)
    r = ints((N, cins[0]) + sp, -4, 4, g)
    skip = ints((N, cins[1]) + sp, 0, 3, g)
    scale = torch.tensor([0.0, 0.5, 1.0, 2.0])[torch.randint(0, 4, (N, cins[0]), generator=g)]
    shift = ints((N, cins[0]), -2, 2, g)
    bc = (N, cins[0]) + (1,) * ndim
    a0 = torch.relu(scale.reshape(bc) * r + shift.reshape(bc))
    x = torch.cat([a0, skip], dim=1)
    w = ints((cout, sum(cins)) + (1,) * ndim, -1, 1, g, density=0.3).requires_grad_(True)
    conv = F.conv3d if ndim == 3 else F.conv2d
    ref = conv(x, w)
    assert float(ref.abs().max()) <= 512
    dy = ints(tuple(ref.shape), -1, 1, g, density=0.5)
    ref.backward(dy)
    act = (ops.aligned_like(scale.to(dev)), ops.aligned_like(shift.to(dev)))
    r0, x1 = to_dev(cl(r), dtype, dev), to_dev(cl(skip), dtype, dev)
    out = ops.conv(r0, ops.pack(w.detach().to(dev), "conv_fwd", dtype), dtype, ndim, 1, x1=x1, cout=cout, act=act)
    assert ops.last_conv_kernel == 1
    assert torch.equal(ncdhw(out.float().cpu(), ndim), ref.detach())
    got = ops.wgrad(to_dev(cl(dy), dtype, dev), r0, dtype, ndim, 1, x1=x1, act=act)
    assert torch.equal(got.cpu(), w.grad), float((got.cpu() - w.grad).abs().max())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(3, 2, (2, 4, 6), 16, [16, 16]), (3, 1, (4, 4, 8), 32, [32, 32]), (2, 2, (8, 16), 16, [16, 48])])
def test_conv_two_outputs_exact(dev, dtype, case):
    """seg_conv_args.out1 / Cout0: the data-gradients of BOTH sources of a virtual concat under a 1^d conv (networks/VNet3d.py:75-77 -> autograd) from one pass
    over d(raw): GEMM columns below Cout0 go to the first tensor, the others to the second."""
    ndim, N, sp, K, couts = case
    g = torch.Generator().manual_seed(sum(sp) + K)
    x = ints((N, K) + sp, -2, 2, g)
    w = ints((sum(couts), K) + (1,) * ndim, -1, 1, g, density=0.4)
    conv = F.conv3d if ndim == 3 else F.conv2d
    ref = conv(x, w)
    o0, o1 = ops.conv(to_dev(cl(x), dtype, dev), ops.pack(w.to(dev), "conv_fwd", dtype), dtype, ndim, 1, cout=sum(couts), split=couts[0])
    assert ops.last_conv_kernel == 1
    assert torch.equal(ncdhw(o0.float().cpu(), ndim), ref[:, :couts[0]]) and torch.equal(ncdhw(o1.float().cpu(), ndim), ref[:, couts[0]:])
    if couts[0] == couts[1]:
        # rq_*: the first output is the gradient dz of a = relu(scale * r + shift); the launch also delivers sum dz * [a > 0] and sum dz * [a > 0] * r per (n, c) -
        # the GroupNorm-backward reduction over (dz, r) (gn_bwd_reduce_kernel) - exactly on integer data
        r = ints((N, couts[0]) + sp, -3, 3, g)
        scale = torch.tensor([0.0, 0.5, 1.0, 2.0])[torch.randint(0, 4, (N, couts[0]), generator=g)]
        shift = ints((N, couts[0]), -2, 2, g)
        bc = (N, couts[0]) + (1,) * ndim
        gate = (scale.reshape(bc) * r + shift.reshape(bc) > 0).double()
        dz = ref[:, :couts[0]].double()
        want = torch.stack([(dz * gate).flatten(2).sum(2), (dz * gate * r.double()).flatten(2).sum(2)], dim=2)
        o0b, o1b, q = ops.conv(to_dev(cl(x), dtype, dev), ops.pack(w.to(dev), "conv_fwd", dtype), dtype, ndim, 1, cout=sum(couts), split=couts[0],
                               rq=(to_dev(cl(r), dtype, dev), ops.aligned_like(scale.to(dev)), ops.aligned_like(shift.to(dev))))
        assert torch.equal(o0b, o0) and torch.equal(o1b, o1)
        assert torch.equal(q.cpu(), want), float((q.cpu() - want).abs().max())


def test_activation_on_load_is_refused_where_no_kernel_applies_it(dev):
    lib = _capi.lib_for(dev)
    x = to_dev(torch.zeros(1, 3, 4, 6, 16), "f32", dev)                      # 72 voxel rows: not a multiple of 16 -> LDS-staged kernel
    w = ops.pack(torch.zeros(16, 16, 1, 1, 1).to(dev), "conv_fwd", "f32")
    act = (ops.aligned_like(torch.ones(1, 16).to(dev)), ops.aligned_like(torch.zeros(1, 16).to(dev)))
    with pytest.raises(RuntimeError, match="act_scale"):
        ops.conv(x, w, "f32", 3, 1, cout=16, act=act)
    with pytest.raises(RuntimeError, match="act_scale"):                        # fp32 tensors: no direct weight-gradient kernel
        ops.wgrad(to_dev(torch.zeros(1, 2, 4, 8, 16), "f32", dev), to_dev(torch.zeros(1, 2, 4, 8, 16), "f32", dev), "f32", 3, 1, act=act)


@pytest.mark.parametrize("dtype", DT)
def test_wgrad_conv_transpose_and_stem_exact(dev, dtype):
    g = torch.Generator().manual_seed(3)
    # ConvTranspose k2 s2 (32 -> 16): dW[ci][co][a] = sum_coarse x[m][ci] * dy[2m+a][co]
    x = ints((2, 32, 2, 3, 2), -2, 2, g)
    w = torch.zeros((32, 16, 2, 2, 2), requires_grad=True)
    y = F.conv_transpose3d(x, w, stride=2)
    dy = ints(tuple(y.shape), -2, 2, g)
    y.backward(dy)
    got = ops.wgrad(to_dev(cl(x), dtype, dev), to_dev(cl(dy), dtype, dev), dtype, 3, 2, 2, 0)
    assert torch.equal(got.cpu(), w.grad)
    # stem: 1 -> 16, 3^3 pad 1 (direct K = 27), and the 1^3 twin
    for k, pad in ((3, 1), (1, 0)):
        x = ints((2, 1, 5, 6, 4), -3, 3, g)
        w = torch.zeros((16, 1, k, k, k), requires_grad=True)
        y = F.conv3d(x, w, padding=pad)
        dy = ints(tuple(y.shape), -2, 2, g)
        y.backward(dy)
        got = ops.wgrad(to_dev(cl(dy), dtype, dev), to_dev(cl(x), dtype, dev), dtype, 3, k, 1, pad, stem=True)
        assert torch.equal(got.cpu(), w.grad)
    # 2-D RGB stem: 3 -> 16, 3x3
    x = ints((2, 3, 6, 7), -3, 3, g)
    w = torch.zeros((16, 3, 3, 3), requires_grad=True)
    y = F.conv2d(x, w, padding=1)
    dy = ints(tuple(y.shape), -2, 2, g)
    y.backward(dy)
    got = ops.wgrad(to_dev(cl(dy), dtype, dev), to_dev(cl(x), dtype, dev), dtype, 2, 3, 1, 1, stem=True)
    assert torch.equal(got.cpu(), w.grad)


HALO_CASES = [
    # ndim, N, spatial, Cin, Cout   (box shapes: W%16==0 or 12 -> 3x4x16 / 1x8x16, else 3x8x8 / 1x8x8; partial boxes masked)
    (3, 2, (6, 8, 16), 16, 16),
    (3, 1, (3, 5, 12), 32, 32),
    (3, 2, (4, 6, 6), 64, 64),
    (3, 1, (6, 8, 24), 16, 32),
    (3, 1, (3, 4, 16), 32, 16),
    (2, 2, (16, 32), 16, 16),
    (2, 1, (12, 24), 64, 128),
    (3, 1, (2, 3, 5), 128, 64),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3_halo_exact(dev, dtype, case):
    ndim, N, sp, cin, cout = case
    g = torch.Generator().manual_seed(sum(sp) + cin)
    x = ints((N, cin) + sp, -2, 2, g)
    w = ints((cout, cin) + (3,) * ndim, -1, 1, g, density=0.15)
    b = ints((cout,), -3, 3, g)
    conv = F.conv3d if ndim == 3 else F.conv2d
    xr = x.clone().requires_grad_(True)
    ref = conv(xr, w, b, padding=1)
    assert float(ref.abs().max()) <= 256
    out, stats = ops.conv3(to_dev(cl(x), dtype, dev), ops.pack(w.to(dev), "conv_fwd", dtype), dtype, ndim, cout,
                           bias=ops.aligned_like(b.to(dev)), want_stats=True)
    assert torch.equal(ncdhw(out.float().cpu(), ndim), ref.detach())
    rs = torch.stack([ref.detach().double().flatten(2).sum(2), (ref.detach().double() ** 2).flatten(2).sum(2)], dim=2)
    assert torch.equal(stats.cpu(), rs)
    # data-gradient through the same kernel with the flipped layout
    dy = ints(tuple(ref.shape), -1, 1, g, density=0.4)
    ref.backward(dy)
    assert float(xr.grad.abs().max()) <= 256
    got = ops.conv3(to_dev(cl(dy), dtype, dev), ops.pack(w.to(dev), "conv_dgrad", dtype), dtype, ndim, cin)
    assert torch.equal(ncdhw(got.float().cpu(), ndim), xr.grad)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3_halo_exact(dev, dtype, case):
    ndim, N, sp, cin, cout = case
    g = torch.Generator().manual_seed(sum(sp) + cin)
    x = ints((N, cin) + sp, -2, 2, g)
    w = ints((cout, cin) + (3,) * ndim, -1, 1, g, density=0.15)
    b = ints((cout,), -3, 3, g)
    conv = F.conv3d if ndim == 3 else F.conv2d
    xr = x.clone().requires_grad_(True)
    ref = conv(xr, w, b, padding=1)
    assert float(ref.abs().max()) <= 256
    out, stats = ops.conv3(to_dev(cl(x), dtype, dev), ops.pack(w.to(dev), "conv_fwd", dtype), dtype, ndim, cout,
                           bias=ops.aligned_like(b.to(dev)), want_stats=True)
    assert torch.equal(ncdhw(out.float().cpu(), ndim), ref.detach())
    rs = torch.stack([ref.detach().double().flatten(2).sum(2), (ref.detach().double() ** 2).flatten(2).sum(2)], dim=2)
    assert torch.equal(stats.cpu(), rs)
    # data-gradient through the same kernel with the flipped layout
    dy = ints(tuple(ref.shape), -1, 1, g, density=0.4)
    ref.backward(dy)
    assert float(xr.grad.abs().max()) <= 256
    got = ops.conv3(to_dev(cl(dy), dtype, dev), ops.pack(w.to(dev), "conv_dgrad", dtype), dtype, ndim, cin)
    assert torch.equal(ncdhw(got.float().cpu(), ndim), xr.grad)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", HALO_CASES)
def test_wgrad3_halo_exact(dev, dtype, case):
    ndim, N, sp, cin, cout = case
    g = torch.Generator().manual_seed(sum(sp) + cout)
    x = ints((N, cin) + sp, -2, 2, g)
    w = torch.zeros((cout, cin) + (3,) * ndim, requires_grad=True)
    conv = F.conv3d if ndim == 3 else F.conv2d
    y = conv(x, w, padding=1)
    dy = ints(tuple(y.shape), -2, 2, g)
    y.backward(dy)
    got = ops.wgrad3(to_dev(cl(dy), dtype, dev), to_dev(cl(x), dtype, dev), dtype, ndim)
    assert torch.equal(got.cpu(), w.grad), float((got.cpu() - w.grad).abs().max())



@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("sp,N", [((4, 8, 16), 1), ((9, 11, 32), 2), ((8, 16, 12), 1)])
def test_wgrad3_big_box_16_channels_exact(dev, dtype, sp, N, monkeypatch):
    """the 4 x 8 x 16-box instantiation of wgrad3_kernel that serves the 16 -> 16 channel convs of the finest level (Wgrad3Big16, conv3.hip; autograd of
    networks/VNet3d.py:8 at 96^3): whole boxes, ragged boxes on every axis, the 12-wide row of the deep levels; SEG_W3_BOX16=2 forces it
    onto these small volumes."""
    if dtype == "bf16":
        conftest.checker_slow(dev, "the bf16 twin of every case runs on the GPU; the f16 cases run on the host checker")
    monkeypatch.setenv("SEG_WGRAD3X", "0")
    monkeypatch.setenv("SEG_W3_BOX16", "2")
    g = torch.Generator().manual_seed(sum(sp) + N)
    x = ints((N, 16) + sp, -2, 2, g)
    w = torch.zeros((16, 16, 3, 3, 3), requires_grad=True)
    y = F.conv3d(x, w, padding=1)
    dy = ints(tuple(y.shape), -2, 2, g)
    y.backward(dy)
    got = ops.wgrad3(to_dev(cl(dy), dtype, dev), to_dev(cl(x), dtype, dev), dtype, 3)
    assert torch.equal(got.cpu(), w.grad), float((got.cpu() - w.grad).abs().max())
    # 32 -> 16 over a virtual concat of two 16-channel tensors (the UNet decoder's first conv at the finest level): two q-tiles on the same boxes
    x2 = ints((N, 32) + sp, -2, 2, g)
    w2 = torch.zeros((16, 32, 3, 3, 3), requires_grad=True)
    y2 = F.conv3d(x2, w2, padding=1)
    y2.backward(dy)
    xs = torch.split(x2, [16, 16], dim=1)
    got = ops.wgrad3(to_dev(cl(dy), dtype, dev), to_dev(cl(xs[0]), dtype, dev), dtype, 3, x1=to_dev(cl(xs[1]), dtype, dev))
    assert torch.equal(got.cpu(), w2.grad), float((got.cpu() - w2.grad).abs().max())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(3, 1, (3, 8, 16), [16, 16], 16), (3, 2, (4, 6, 8), [32, 32], 32), (2, 1, (8, 16), [64, 64], 64),
                                  (3, 1, (7, 9, 17), [32, 32], 64), (2, 2, (11, 13), [16, 16], 32)])
def test_wgrad3_concat_exact(dev, dtype, case):
    """x = virtual concat of two tensors (UNet decoder blocks): q-tiles never straddle the sources; partial boxes; several
    boxes per workgroup."""
    ndim, N, sp, cins, cout = case
    if dtype == "bf16" and N * sp[0] * sp[1] * (sp[2] if ndim == 3 else 1) * sum(cins) * cout > 2_000_000:
        conftest.checker_slow(dev, "big bf16 case: 20 s per kernel on the host checker (the f16 twin runs there)")
    g = torch.Generator().manual_seed(sum(sp) + cout + 5)
    x = ints((N, sum(cins)) + sp, -2, 2, g)
    w = torch.zeros((cout, sum(cins)) + (3,) * ndim, requires_grad=True)
    conv = F.conv3d if ndim == 3 else F.conv2d
    y = conv(x, w, padding=1)
    dy = ints(tuple(y.shape), -2, 2, g, density=0.6)
    y.backward(dy)
    assert float(w.grad.abs().max()) < 2 ** 20
    xs = torch.split(x, cins, dim=1)
    got = ops.wgrad3(to_dev(cl(dy), dtype, dev), to_dev(cl(xs[0]), dtype, dev), dtype, ndim, x1=to_dev(cl(xs[1]), dtype, dev))
    assert torch.equal(got.cpu(), w.grad), float((got.cpu() - w.grad).abs().max())

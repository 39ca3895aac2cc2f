"""Fused input block (csrc/stemx.hip) against torch on integer-valued data: GroupNorm partial sums, the normalised / activated
sum of the two stem branches, the backward reduction and the stem weight gradients - all without raw / d(raw) tensors.
Reference ops: networks/VNet3d.py:25-43 (InputTransition), networks/Unet3d.py:64-86 (first conv of a block)."""
import pytest
import torch
import torch.nn.functional as F

from pytorchdeeplearing_amd import ops
from test_ops import cl, ncdhw, ints, to_dev

DT = ["f32", "f16", "bf16"]
# ndim, N, spatial, Cimg, two branches
CASES = [(3, 2, (4, 16, 16), 1, True), (3, 1, (3, 10, 20), 1, True), (3, 2, (2, 8, 32), 1, False),
         (2, 2, (16, 32), 1, True), (2, 1, (19, 24), 3, True), (2, 2, (32, 16), 2, False)]


def dyadic(shape, g, vals):
    idx = torch.randint(0, len(vals), shape, generator=g)
    return torch.tensor(vals)[idx]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CASES)
def test_stemx_modes_exact(dev, dtype, case):
    ndim, N, sp, cimg, two = case
    tdt = ops.TORCH_DTYPE[dtype]
    g = torch.Generator().manual_seed(sum(sp) * 5 + cimg + (7 if two else 0))
    conv = F.conv3d if ndim == 3 else F.conv2d
    x = ints((N, cimg) + sp, -2, 2, g)
    w3 = ints((16, cimg) + (3,) * ndim, -1, 1, g, density=0.6).requires_grad_(True)
    w1 = ints((16, cimg) + (1,) * ndim, -2, 2, g).requires_grad_(True)
    b3, b1 = ints((16,), -2, 2, g), ints((16,), -1, 1, g)
    r3 = conv(x, w3, b3, padding=1)
    r1 = conv(x, w1, b1)
    assert float(r3.abs().max()) <= 64
    img = to_dev(cl(x), dtype, dev)
    w3p = ops.pack(w3.detach().to(dev), "conv_fwd", dtype)
    w1p = ops.pack(w1.detach().to(dev), "conv_fwd", dtype) if two else None
    kw = dict(w1p=w1p, bias3=ops.aligned_like(b3.to(dev)), bias1=ops.aligned_like(b1.to(dev)) if two else None)
    moments = lambda r: torch.stack([r.detach().double().flatten(2).sum(2), (r.detach().double() ** 2).flatten(2).sum(2)], dim=2)
    # ---- mode 0: GroupNorm partial sums of both branches
    s3, s1 = ops.stemx(0, img, w3p, dtype, ndim, **kw)
    assert torch.equal(s3.cpu(), moments(r3))
    if two:
        assert torch.equal(s1.cpu(), moments(r1))
    # ---- mode 1: y = relu(s3*r3+t3) + relu(s1*r1+t1), power-of-two scales (every product exact)
    sc = [dyadic((N, 16), g, [0.5, 1.0, 2.0, -1.0, -0.5]) for _ in range(2)]
    sh = [ints((N, 16), -3, 3, g) * 0.5 for _ in range(2)]
    bc = lambda t: t.reshape((N, 16) + (1,) * ndim)
    y = (bc(sc[0]) * r3 + bc(sh[0])).clamp_min(0)
    if two:
        y = y + (bc(sc[1]) * r1 + bc(sh[1])).clamp_min(0)
    dsc = [ops.aligned_like(t.to(dev)) for t in sc]
    dsh = [ops.aligned_like(t.to(dev)) for t in sh]
    out = ops.stemx(1, img, w3p, dtype, ndim, scale=dsc, shift=dsh, **kw)
    want = y.detach().to(tdt).float()
    got = ncdhw(out.float().cpu(), ndim)
    assert torch.equal(got, want), float((got - want).abs().max())
    # ---- mode 2: Q = {sum dz, sum dz * r}, dz = (sum of the gradient sources) * [scale*r + shift > 0]
    ndy = 1 + (sum(sp) % 3)
    dys = [ints((N, 16) + sp, -1, 1, g, density=0.5) for _ in range(ndy)]
    dysum = sum(dys)
    ddys = [to_dev(cl(d), dtype, dev) for d in dys]
    q3, q1 = ops.stemx(2, img, w3p, dtype, ndim, scale=dsc, shift=dsh, dys=ddys, **kw)
    dz3 = dysum * ((bc(sc[0]) * r3 + bc(sh[0])) > 0)
    dz1 = dysum * ((bc(sc[1]) * r1 + bc(sh[1])) > 0)
    qq = lambda dz, r: torch.stack([dz.detach().double().flatten(2).sum(2), (dz.detach().double() * r.detach().double()).flatten(2).sum(2)], dim=2)
    assert torch.equal(q3.cpu(), qq(dz3, r3))
    if two:
        assert torch.equal(q1.cpu(), qq(dz1, r1))
    # ---- mode 3: weight gradients from d(raw) = A*dz + B*r + C (rounded to the run dtype), never stored
    cf = [torch.stack([dyadic((N, 16), g, [1.0, 0.5, -1.0, 2.0]), dyadic((N, 16), g, [0.0, 0.25, -0.25, 0.5]),
                       dyadic((N, 16), g, [0.0, 0.5, -0.5, 1.0])], dim=2) for _ in range(2)]
    draw = lambda c, dz, r: (bc(c[..., 0]) * dz + bc(c[..., 1]) * r + bc(c[..., 2])).detach().to(tdt).float()
    d3, d1 = draw(cf[0], dz3, r3), draw(cf[1], dz1, r1)
    r3.backward(d3)
    r1.backward(d1)
    assert float(w3.grad.abs().max()) < 2 ** 20
    dcf = [ops.aligned_like(t.contiguous().to(dev)) for t in cf]
    dw3, dw1 = ops.stemx(3, img, w3p, dtype, ndim, scale=dsc, shift=dsh, dys=ddys, coef=dcf, **kw)
    assert torch.equal(dw3.cpu(), w3.grad), float((dw3.cpu() - w3.grad).abs().max())
    if two:
        assert torch.equal(dw1.cpu(), w1.grad), float((dw1.cpu() - w1.grad).abs().max())


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("f16", 2e-2), ("bf16", 1.5e-1)])
def test_stemx_real_valued_data_repeated_launches(dev, dtype, tol):
    """Real-valued image / weights, every mode launched twice on a box grid with several tiles per wave: padding k slots must read
    zeros whatever an earlier launch (or another kernel) left in LDS - a finite-garbage x 0 product hides on integer data."""
    torch.manual_seed(3)
    N, sp = 2, (4, 16, 32)
    x = torch.randn((N, 1) + sp)
    w3, w1 = torch.randn(16, 1, 3, 3, 3) * 0.3, torch.randn(16, 1, 1, 1, 1)
    b3, b1 = torch.randn(16) * 0.1, torch.randn(16) * 0.1
    tdt = ops.TORCH_DTYPE[dtype]
    q = lambda t: t.to(tdt).float()                          # operands as the kernel sees them
    r3 = q(F.conv3d(q(x), q(w3), b3, padding=1))
    r1 = q(F.conv3d(q(x), q(w1), b1))
    img = to_dev(cl(x), dtype, dev)
    w3p, w1p = ops.pack(w3.to(dev), "conv_fwd", dtype), ops.pack(w1.to(dev), "conv_fwd", dtype)
    kw = dict(w1p=w1p, bias3=ops.aligned_like(b3.to(dev)), bias1=ops.aligned_like(b1.to(dev)))
    sc = [ops.aligned_like((torch.rand(N, 16) + 0.5).to(dev)) for _ in range(2)]
    sh = [ops.aligned_like((torch.randn(N, 16) * 0.2).to(dev)) for _ in range(2)]
    bc = lambda t: t.cpu().reshape(N, 16, 1, 1, 1)
    y = (bc(sc[0]) * r3 + bc(sh[0])).clamp_min(0) + (bc(sc[1]) * r1 + bc(sh[1])).clamp_min(0)
    dy = torch.randn((N, 16) + sp)
    ddy = [to_dev(cl(dy), dtype, dev)]
    for rep in range(2):
        s3, s1 = ops.stemx(0, img, w3p, dtype, 3, **kw)
        assert torch.isfinite(s3).all() and torch.isfinite(s1).all()
        ref = r3.double().flatten(2).sum(2)
        assert float((s3[..., 0].cpu() - ref).abs().max()) <= tol * float(r3.abs().double().flatten(2).sum(2).max())
        out = ops.stemx(1, img, w3p, dtype, 3, scale=sc, shift=sh, **kw)
        got = ncdhw(out.float().cpu(), 3)
        assert torch.isfinite(got).all() and float((got - y).abs().max()) <= tol * float(y.abs().max())
        q3, q1 = ops.stemx(2, img, w3p, dtype, 3, scale=sc, shift=sh, dys=ddy, **kw)
        assert torch.isfinite(q3).all() and torch.isfinite(q1).all()
        cf = [ops.aligned_like(torch.stack([torch.ones(N, 16), torch.zeros(N, 16), torch.zeros(N, 16)], dim=2).contiguous().to(dev)) for _ in range(2)]
        dw3, dw1 = ops.stemx(3, img, w3p, dtype, 3, scale=sc, shift=sh, dys=ddy, coef=cf, **kw)
        # with coef = (1, 0, 0): d(raw) = dz, so dw3 is the plain conv weight gradient of the masked gradient
        dz3 = q(q(dy) * ((bc(sc[0]) * r3 + bc(sh[0])) > 0))
        w = q(w3).clone().requires_grad_(True)
        F.conv3d(q(x), w, None, padding=1).backward(dz3)
        assert torch.isfinite(dw3).all() and float((dw3.cpu() - w.grad).abs().max()) <= tol * float(w.grad.abs().max()) + 1e-3

"""Register-blocked halo conv (csrc/conv3x.hip) against torch.nn.functional on integer-valued data: every tiling of the
kernel, forward (+ bias + GroupNorm partial sums), data-gradient (flipped fragment-major weights), virtual concat inputs
(aligned and straddling a 32-channel chunk), partial boxes, several chunk groups.  Bit-exact for f16 and bf16."""
import os

import pytest

import conftest
import torch
import torch.nn.functional as F

from pytorchdeeplearing_amd import ops
from test_ops import cl, ncdhw, ints, to_dev

# ndim, N, spatial, Cin (list = concat sources), Cout, tiling ids to run (None: the default pick)
CASES = [
    (3, 1, (3, 9, 18), [32], 32, [0, 1, 13, 14, 20, 21, 22, 23]),          # partial boxes in every direction
    (3, 2, (2, 8, 16), [64], 64, [2, 15]),
    (3, 1, (4, 8, 10), [64], 64, [3, 4, 5, 6, 11]),
    (3, 1, (3, 5, 12), [128], 128, [7, 8, 9, 12]),
    (3, 1, (2, 4, 6), [256], 64, [11, 7]),                  # two chunk groups of four resident chunks
    (3, 1, (2, 8, 16), [32], 16, [10]),
    (3, 1, (2, 6, 16), [16, 16], 32, [0]),                  # concat straddling one 32-channel chunk
    (3, 1, (2, 5, 8), [32, 32], 64, [3]),                   # concat on a chunk boundary
    (3, 1, (2, 4, 8), [64, 64], 64, [None]),
    (3, 1, (3, 9, 18), [16], 16, [24, 25]),                 # Cin == 16: two taps per MFMA step, flat-K weights
    (3, 2, (2, 8, 32), [16], 32, [26, 27]),
    (3, 1, (3, 9, 18), [16], 16, [28, 29, None]),           # Cin == 16, halo fragments reused across the kh taps (weight layout 3); partial boxes
    (3, 2, (5, 8, 32), [16], 16, [29, None]),
    (2, 1, (19, 24), [16], 16, [56]),
    (2, 2, (16, 16), [16], 32, [57, None]),
    (2, 2, (17, 20), [32], 32, [32, 39]),
    (2, 1, (16, 32), [64], 64, [33, 34, 35, 38]),
    (2, 1, (9, 16), [128], 128, [36]),
    (2, 1, (8, 16), [32], 16, [37]),
    (2, 1, (12, 24), [16, 16], 64, [38]),
]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_conv3x_exact(dev, dtype, case):
    ndim, N, sp, cins, cout, cfgs = case
    cin = sum(cins)
    vox = N * sp[0] * sp[1] * (sp[2] if ndim == 3 else 1)
    if dtype == "bf16" and vox * cin * cout * len(cfgs) > 3_000_000:       # bf16 arithmetic is the slow part of the host checker (15-40 s)
        conftest.checker_slow(dev, "big bf16 case (the f16 twin runs here)")
    g = torch.Generator().manual_seed(sum(sp) * 3 + cin + cout)
    x = ints((N, cin) + sp, -2, 2, g)
    w = ints((cout, cin) + (3,) * ndim, -1, 1, g, density=0.1)
    b = ints((cout,), -3, 3, g)
    conv = F.conv3d if ndim == 3 else F.conv2d
    xr = x.clone().requires_grad_(True)
    ref = conv(xr, w, b, padding=1)
    assert float(ref.abs().max()) <= 256
    dy = ints(tuple(ref.shape), -1, 1, g, density=0.3)
    ref.backward(dy)
    assert float(xr.grad.abs().max()) <= 256
    xs = torch.split(x, cins, dim=1)
    x0 = to_dev(cl(xs[0]), dtype, dev)
    x1 = to_dev(cl(xs[1]), dtype, dev) if len(xs) > 1 else None
    wf = ops.pack(w.to(dev), "conv_fwd", dtype, frag="all")          # every layout of the shape; conv3x picks the one the tiling reads
    rs = torch.stack([ref.detach().double().flatten(2).sum(2), (ref.detach().double() ** 2).flatten(2).sum(2)], dim=2)
    known = {c["id"]: c for c in ops.conv3x_cfgs(dev)}
    for cfg in cfgs:
        if cfg is not None:
            assert cfg in known and known[cfg]["ndim"] == ndim and cout % known[cfg]["bn"] == 0, (cfg, known.get(cfg))
        out, stats = ops.conv3x(x0, wf, dtype, ndim, cout, bias=ops.aligned_like(b.to(dev)), want_stats=True, x1=x1,
                                cfg=-1 if cfg is None else cfg)
        got = ncdhw(out.float().cpu(), ndim)
        assert torch.equal(got, ref.detach()), (cfg, float((got - ref.detach()).abs().max()))
        assert torch.equal(stats.cpu(), rs), cfg
    # data-gradient: K = Cout (a multiple of 32, or 16), one launch per concat source with its own flipped weights
    if cout % 32 == 0 or cout == 16:
        dyd = to_dev(cl(dy), dtype, dev)
        c_lo = 0
        for ci in cins:
            if ci % 16:
                continue
            wd = ops.pack(w[:, c_lo:c_lo + ci].contiguous().to(dev), "conv_dgrad", dtype, frag="all")
            got = ops.conv3x(dyd, wd, dtype, ndim, ci)
            assert torch.equal(ncdhw(got.float().cpu(), ndim), xr.grad[:, c_lo:c_lo + ci]), ("dgrad", ci)
            c_lo += ci


def test_conv3x_rejects_what_it_cannot_run(dev):
    x = to_dev(torch.zeros(1, 2, 4, 8, 16), "f16", dev)
    w = to_dev(torch.zeros(16, 27 * 16), "f16", dev)
    x1 = to_dev(torch.zeros(1, 2, 4, 8, 8), "f16", dev)
    with pytest.raises(RuntimeError):
        ops.conv3x(x1, w, "f16", 3, 16, x1=x1)      # Cin = 16 as a concat of 8 + 8
    x = to_dev(torch.zeros(1, 2, 4, 8, 32), "f16", dev)
    w = to_dev(torch.zeros(32, 27 * 32), "f16", dev)
    with pytest.raises(RuntimeError):
        ops.conv3x(x, w, "f16", 3, 32, cfg=2)       # tiling 2 produces 64 output channels per workgroup
    with pytest.raises(RuntimeError):
        ops.conv3x(x, w, "f16", 3, 32, cfg=33)      # a 2-D tiling

"""CPU: pin oracle/seg_oracle.py against (a) golden vectors produced by the REAL reference
(tests/golden, made by oracle/make_golden.py) and (b) the live reference when /root/reference is
importable.  The reference ships no tests of its own."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_loader, seg_oracle as seg

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))

NETS = [
    ("vnet3d_bin_16", "vnet", 3, (2, 1, 16, 16, 16), 1, "BinaryDiceLoss"),
    ("unet3d_mc4_16", "unet", 3, (1, 1, 16, 16, 16), 4, "MutilDiceLoss"),
    ("vnet2d_mc2_32", "vnet", 2, (2, 1, 32, 32), 2, "MutilCrossEntropyLoss"),
    ("unet2d_bin_32", "unet", 2, (2, 1, 32, 32), 1, "BinaryCrossEntropyDiceLoss"),
]


def grad_summary(g):
    f = g.detach().double().reshape(-1)
    idx = (torch.arange(16, dtype=torch.int64) * 2654435761 + 12345) % f.numel()
    return np.concatenate([[float(f.sum()), float(f.norm())], f[idx].numpy()])


def test_losses_metrics_vs_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "losses_metrics.npz"))
    z, y = torch.from_numpy(G["z"]), torch.from_numpy(G["y"])
    z4, y4 = torch.from_numpy(G["z4"]), torch.from_numpy(G["y4"])
    a = torch.ones(4)
    cases = [("BinaryDiceLoss", seg.binary_dice_loss, (z, y)),
             ("BinaryCrossEntropyLoss", seg.binary_ce_loss, (z, y)),
             ("BinaryFocalLoss", seg.binary_focal_loss, (z, y)),
             ("BinaryCrossEntropyDiceLoss", seg.binary_ce_dice_loss, (z, y)),
             ("MutilDiceLoss", lambda l, t: seg.multi_dice_loss(l, t, a), (z4, y4)),
             ("MutilCrossEntropyLoss", seg.multi_ce_loss, (z4, y4)),
             ("MutilFocalLoss_g2", lambda l, t: seg.multi_focal_loss(l, t, a, 2), (z4, y4)),
             ("MutilFocalLoss_g3", lambda l, t: seg.multi_focal_loss(l, t, a, 3), (z4, y4))]
    for name, f, (l, t) in cases:
        l = l.clone().requires_grad_(True)
        v = f(l, t)
        v.backward()
        assert abs(float(v) - float(G["loss_" + name])) < 1e-6, name
        np.testing.assert_allclose(l.grad.numpy(), G["grad_" + name], rtol=1e-5, atol=1e-9, err_msg=name)
    assert abs(float(seg.dice_coeff(torch.sigmoid(z), y)) - float(G["dice_coeff"])) < 1e-7
    assert abs(float(seg.iou_coeff(torch.sigmoid(z), y)) - float(G["iou_coeff"])) < 1e-7
    assert abs(float(seg.multiclass_dice_coeff(torch.softmax(z4, 1), y4)) - float(G["multiclass_dice_coeff"])) < 1e-7
    # SURVEY.md §8(c) known-answer spot values (torch-2.10-CPU)
    assert abs(float(G["loss_BinaryDiceLoss"]) - 0.6185579) < 1e-6
    assert abs(float(G["loss_MutilDiceLoss"]) + 0.2545772) < 1e-6
    assert abs(float(G["multiclass_dice_coeff"]) - 0.1862026) < 1e-6


@pytest.mark.parametrize("tag,kind,ndim,shape,numclass,loss", NETS)
def test_net_restatement_vs_golden(golden_dir, tag, kind, ndim, shape, numclass, loss):
    G = np.load(os.path.join(golden_dir, tag + ".npz"))
    params = seg.perturb_params(seg.init_params(kind, ndim, shape[1], numclass, seed=0), seed=7)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], numclass, seed=1)
    assert abs(float(x.double().sum()) - float(G["x_sum"])) < 1e-9 and int(y.sum()) == int(G["y_sum"])
    alpha = torch.ones(numclass)
    r = seg.forward_backward(kind, params, x, y, loss, masks=None, alpha=alpha)
    np.testing.assert_allclose(r["logits"].numpy(), G["eval_logits"], rtol=0, atol=2e-5)
    assert abs(float(r["loss"]) - float(G["eval_loss"])) < 1e-6
    names = list(G["grad_names"])
    assert names == list(r["grads"].keys())
    gs = np.stack([grad_summary(r["grads"][k]) for k in names])
    np.testing.assert_allclose(gs, G["grad_summary"], rtol=2e-3, atol=2e-6)
    # train mode with the recorded channel-dropout multipliers
    ch = G["train_mask_channels"]
    assert list(ch) == seg.dropout_channels(kind)
    masks = [torch.from_numpy(G["train_masks"][i, :, :c].astype(np.float32)) for i, c in enumerate(ch)]
    rt = seg.forward_backward(kind, params, x, y, loss, masks=masks, alpha=alpha)
    np.testing.assert_allclose(rt["logits"].numpy(), G["train_logits"], rtol=0, atol=2e-5)
    assert abs(float(rt["loss"]) - float(G["train_loss"])) < 1e-6
    gs = np.stack([grad_summary(rt["grads"][k]) for k in names])
    np.testing.assert_allclose(gs, G["train_grad_summary"], rtol=2e-3, atol=2e-6)


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree only exists in the build container")
def test_restatement_vs_live_reference():
    nets, losses, metric = ref_loader.load()
    m = nets.VNet3d(1, 1)
    torch.manual_seed(3)
    m.apply(nets.initialize_weights)
    assert list(m.state_dict().keys()) == list(seg.vnet_param_shapes(3, 1, 1).keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == seg.vnet_param_shapes(3, 1, 1)[k], k
    u = nets.UNet3d(1, 4)
    assert list(u.state_dict().keys()) == list(seg.unet_param_shapes(3, 1, 4).keys())
    x, y = seg.synthetic_batch(1, (16, 16, 16), 1, 1, seed=5)
    m.eval()
    with torch.no_grad():
        lg, pr = m(x)
    lo, po = seg.vnet_forward(dict(m.state_dict()), x)
    assert float((lg - lo).abs().max()) < 1e-5
    assert float(metric.dice_coeff(pr, y)) == float(seg.dice_coeff(po, y))
    # dropout draw order == oracle.draw_masks order
    torch.manual_seed(11)
    mk = seg.draw_masks("vnet", 1)
    torch.manual_seed(11)
    m.train()
    with torch.no_grad():
        lt, _ = m(x)
    lo2, _ = seg.vnet_forward(dict(m.state_dict()), x, masks=mk)
    assert float((lt - lo2).abs().max()) < 1e-5


def test_adamw_restatement_vs_torch():
    torch.manual_seed(0)
    p0 = {"a": torch.randn(5, 3), "b": torch.randn(7)}
    for decoupled in (True, False):
        ps = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
        opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ps.values(), lr=1e-3)
        cur, st = {k: v.clone() for k, v in p0.items()}, {}
        for it in range(3):
            gr = {k: torch.randn_like(v) for k, v in p0.items()}
            for k in ps:
                ps[k].grad = gr[k].clone()
            opt.step()
            cur = seg.adamw_step(cur, gr, st, weight_decay=0.01 if decoupled else 0.0, decoupled=decoupled)
        for k in ps:
            assert float((ps[k].detach() - cur[k]).abs().max()) < 1e-6

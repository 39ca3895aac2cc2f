"""B2 / B3 boundary: nn.Module networks with reference state_dict layout, autograd
through the engine, `model.apply(initialize_weights)`, torch optimisers on the parameter views;
losses / metrics modules against the golden vectors produced by the real reference."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_loader, seg_oracle as seg
from pytorchdeeplearing_amd import losses, metric, networks


def test_losses_metrics_modules_vs_reference_golden(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "losses_metrics.npz"))
    z, y = torch.from_numpy(G["z"]).to(dev), torch.from_numpy(G["y"]).to(dev)
    z4, y4 = torch.from_numpy(G["z4"]).to(dev), torch.from_numpy(G["y4"]).to(dev)
    a = torch.ones(4)
    cases = [("BinaryDiceLoss", losses.BinaryDiceLoss(), z, y), ("BinaryCrossEntropyLoss", losses.BinaryCrossEntropyLoss(), z, y),
             ("BinaryFocalLoss", losses.BinaryFocalLoss(), z, y), ("BinaryCrossEntropyDiceLoss", losses.BinaryCrossEntropyDiceLoss(), z, y),
             ("MutilDiceLoss", losses.MutilDiceLoss(a), z4, y4), ("MutilCrossEntropyLoss", losses.MutilCrossEntropyLoss(a), z4, y4),
             ("MutilFocalLoss_g2", losses.MutilFocalLoss(a, gamma=2), z4, y4), ("MutilFocalLoss_g3", losses.MutilFocalLoss(a, gamma=3), z4, y4)]
    for name, f, l, t in cases:
        l = l.clone().requires_grad_(True)
        v = f(l, t)
        v.backward()
        assert abs(float(v) - float(G["loss_" + name])) < 2e-6, name
        np.testing.assert_allclose(l.grad.cpu().numpy(), G["grad_" + name], rtol=2e-4, atol=1e-9, err_msg=name)
    assert abs(float(metric.dice_coeff(torch.sigmoid(z), y)) - float(G["dice_coeff"])) < 1e-7
    assert abs(float(metric.iou_coeff(torch.sigmoid(z), y)) - float(G["iou_coeff"])) < 1e-7
    assert abs(float(metric.multiclass_dice_coeff(torch.softmax(z4, 1), y4)) - float(G["multiclass_dice_coeff"])) < 1e-7
    # label dtypes the datasets may hand over
    for dt in (torch.uint8, torch.int32, torch.float32):
        assert abs(float(losses.BinaryDiceLoss()(z, y.to(dt))) - float(G["loss_BinaryDiceLoss"])) < 2e-6


def test_multiclass_iou_coeff_vs_reference_golden(dev, golden_dir):
    """M4 (model/metric.py:204-215): golden values come from the reference's own function text with its unsatisfiable size
    assert removed (oracle/make_golden.py:METRIC_REPAIRS); 3-D with every class present, with an absent class, and 2-D."""
    G = np.load(os.path.join(golden_dir, "metric_extra.npz"))
    z4, z3 = torch.from_numpy(G["z4"]), torch.from_numpy(G["z3"])
    for key, z, y in (("miou_a", z4, G["y4"]), ("miou_b", z4, G["y4b"]), ("miou_2d", z3, G["y3"])):
        y = torch.from_numpy(y)
        p = torch.softmax(z, 1)
        assert abs(float(seg.multiclass_iou_coeff(p, y)) - float(G[key])) < 1e-7, key                 # the restatement
        assert abs(float(metric.multiclass_iou_coeff(p.to(dev), y.to(dev))) - float(G[key])) < 1e-7, key      # the kernel
        for dt in (torch.uint8, torch.int32):
            assert abs(float(metric.multiclass_iou_coeff(p.to(dev), y.to(dev).to(dt))) - float(G[key])) < 1e-7
    assert abs(float(metric.multiclass_dice_coeff(torch.softmax(z4, 1).to(dev), torch.from_numpy(G["y4"]).to(dev))) - float(G["mdice_a"])) < 1e-7
    assert float(G["miou_a"]) > 0.05          # the fixture is not the trivial all-background case


EXTRA_LOSSES = ["BinaryJaccardLoss", "BinaryELDiceLoss", "BinaryTverskyLoss", "BinarySSLoss", "MutilCrossEntropyDiceLoss", "MutilELDiceLoss"]


def test_oracle_extra_losses_equal_reference_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "losses_extra.npz"))
    a = torch.from_numpy(G["alpha"])
    for name in EXTRA_LOSSES:
        l = torch.from_numpy(G["z4" if name.startswith("Mutil") else "z"]).clone().requires_grad_(True)
        t = torch.from_numpy(G["y4" if name.startswith("Mutil") else "y"])
        v = seg.loss_fn(name, a)(l, t)
        v.backward()
        assert abs(float(v) - float(G["loss_" + name])) < 1e-6, name
        np.testing.assert_allclose(l.grad.numpy(), G["grad_" + name], rtol=1e-4, atol=1e-10, err_msg=name)


@pytest.mark.parametrize("name", EXTRA_LOSSES)
def test_extra_loss_modules_vs_reference_golden(dev, golden_dir, name):
    """the model/losses.py classes outside the wrappers' loss_name list, from the same reduction sums; the
    multi-class case has an absent class and non-uniform alpha."""
    G = np.load(os.path.join(golden_dir, "losses_extra.npz"))
    mc = name.startswith("Mutil")
    l = torch.from_numpy(G["z4" if mc else "z"]).to(dev).clone().requires_grad_(True)
    t = torch.from_numpy(G["y4" if mc else "y"]).to(dev)
    f = getattr(losses, name)(torch.from_numpy(G["alpha"])) if mc else getattr(losses, name)()
    v = f(l, t)
    v.backward()
    assert abs(float(v) - float(G["loss_" + name])) < 2e-6, name
    np.testing.assert_allclose(l.grad.cpu().numpy(), G["grad_" + name], rtol=3e-4, atol=1e-9, err_msg=name)
    from model import losses as shim                     # the script-facing package exports them too
    assert hasattr(shim, name)


REST_LOSSES = ["MutilTverskyLoss", "MutilSSLoss", "MCC_Loss", "BinaryLovaszLoss", "LovaszLoss"]


def _rest_case(G, name, dev=None):
    """inputs of the round-3 golden cases (oracle/make_golden.py:make_losses_extra, second block)"""
    if name == "MCC_Loss":
        x, t = torch.sigmoid(torch.from_numpy(G["z"])), torch.from_numpy(G["y"]).unsqueeze(1).float()
    elif name == "BinaryLovaszLoss":
        x, t = torch.from_numpy(G["lov_z"]), torch.from_numpy(G["y"])
    else:
        x, t = torch.from_numpy(G["z4"]), torch.from_numpy(G["y4"])
    if dev is not None:
        x, t = x.to(dev), t.to(dev)
    return x.clone().requires_grad_(True), t


def test_oracle_remaining_losses_equal_reference_golden(golden_dir):
    """the oracle restatements of the model/losses.py classes built in round 3 against goldens made by the reference classes themselves
    (with the instance attributes of make_golden.LOSS_REPAIRS)"""
    G = np.load(os.path.join(golden_dir, "losses_extra.npz"))
    a = torch.from_numpy(G["alpha"])
    for name in REST_LOSSES:
        x, t = _rest_case(G, name)
        v = seg.loss_fn(name, a)(x, t)
        v.backward()
        assert abs(float(v) - float(G["loss_" + name])) < 2e-6, name
        np.testing.assert_allclose(x.grad.numpy(), G["grad_" + name], rtol=1e-4, atol=1e-9, err_msg=name)


@pytest.mark.parametrize("name", REST_LOSSES)
def test_remaining_loss_modules_vs_reference_golden(dev, golden_dir, name):
    """SURVEY 8f N4, the rest of model/losses.py: MutilTversky / MutilSS / MCC from the reduction sums, the two Lovasz losses through the
    sort-based kernels (continuous random scores: no ties, so the sub-gradient is unique)."""
    G = np.load(os.path.join(golden_dir, "losses_extra.npz"))
    x, t = _rest_case(G, name, dev)
    a = torch.from_numpy(G["alpha"])
    if name == "MutilTverskyLoss":
        f = losses.MutilTverskyLoss(a)
        with pytest.raises(AttributeError):
            f(x, t)                                       # like the reference: beta is not defined until the caller sets it
        f.beta = 0.7
    elif name == "MutilSSLoss":
        f = losses.MutilSSLoss(a)
        f.r = 0.1
    else:
        f = getattr(losses, name)()
    v = f(x, t)
    v.backward()
    assert abs(float(v) - float(G["loss_" + name])) < 3e-6 * max(1.0, abs(float(G["loss_" + name]))), name
    np.testing.assert_allclose(x.grad.cpu().numpy(), G["grad_" + name], rtol=3e-4, atol=2e-9, err_msg=name)
    from model import losses as shim
    assert hasattr(shim, name)


def test_lovasz_larger_case_vs_oracle(dev):
    """a volume that needs several radix-sort passes and scan blocks (2 x 3 x 40 x 36 x 33), one class absent"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 40, 36, 33, generator=g)
    t = torch.randint(0, 2, (2, 40, 36, 33), generator=g)            # class 2 never occurs
    xr = x.clone().requires_grad_(True)
    ref = seg.multi_lovasz_loss(xr, t)
    ref.backward()
    xd = x.to(dev).requires_grad_(True)
    v = losses.LovaszLoss()(xd, t.to(dev))
    v.backward()
    assert abs(float(v) - float(ref)) < 1e-5 * abs(float(ref))
    # the Jaccard increments are differences of float32 values near 1: absolute resolution ~6e-8 (one ulp), whatever their size
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-4, atol=2e-7)
    assert float(xd.grad[:, 2].abs().max()) == 0.0
    zb = torch.randn(2, 40, 36, 33, generator=g)
    zr = zb.clone().requires_grad_(True)
    refb = seg.binary_lovasz_loss(zr, t)
    refb.backward()
    zd = zb.to(dev).requires_grad_(True)
    vb = losses.BinaryLovaszLoss()(zd, t.to(dev))
    vb.backward()
    assert abs(float(vb) - float(refb)) < 1e-5 * abs(float(refb))
    np.testing.assert_allclose(zd.grad.cpu().numpy(), zr.grad.numpy(), rtol=2e-4, atol=2e-7)


@pytest.mark.parametrize("cls,kind,ndim,args,shape", [
    ("VNet2d", "vnet", 2, (1, 1), (2, 1, 16, 16)),
    ("UNet2d", "unet", 2, (1, 2), (1, 1, 16, 32)),
])
def test_module_boundary(dev, cls, kind, ndim, args, shape):
    torch.manual_seed(0)
    net = getattr(networks, cls)(*args, dtype="f32")
    ref_shapes = seg.param_shapes(kind, ndim, args[0], args[1])
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref_shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref_shapes[k], k
    net.apply(networks.initialize_weights)
    net = net.to(dev)
    params = seg.perturb_params({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, seed=3)
    net.load_state_dict(params)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], args[1], seed=2)
    x, y = x.to(dev), y.to(dev)
    net.eval()
    logits, probs = net(x)
    r = seg.forward_backward(kind, params, x.cpu(), y.cpu(), "BinaryDiceLoss" if args[1] == 1 else "MutilDiceLoss", alpha=torch.ones(args[1]))
    assert float((logits.detach().cpu() - r["logits"]).abs().max()) < 1e-4
    # reference-style step: loss module -> backward -> torch optimiser acting on the parameter views
    lossf = losses.BinaryDiceLoss() if args[1] == 1 else losses.MutilDiceLoss(torch.ones(args[1]))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    loss = lossf(logits, y)
    opt.zero_grad()
    loss.backward()
    assert abs(float(loss) - float(r["loss"])) < 1e-5
    for k, p in net.named_parameters():
        ref = r["grads"][k]
        assert float((p.grad.cpu() - ref).norm()) <= 5e-3 * float(ref.norm()) + 1e-9, k
    before = net.engine.params.clone()
    opt.step()
    assert not torch.equal(before, net.engine.params)        # the optimiser wrote straight into the flat buffer
    # state_dict round trip into a fresh module (what loading an old .pth does)
    net2 = getattr(networks, cls)(*args, dtype="f32").to(dev)
    net2.load_state_dict(net.state_dict())
    net2.eval()
    l1, _ = net(x)
    l2, _ = net2(x)
    assert torch.equal(l1, l2)
    # training mode draws dropout masks inside the engine: outputs differ from eval, and between calls
    net.train()
    t1, _ = net(x)
    t2, _ = net(x)
    assert not torch.equal(t1, l1) and not torch.equal(t1, t2)


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree only exists in the build container")
def test_module_tree_matches_live_reference():
    nets, _, _ = ref_loader.load()
    from tests.conftest import emu_library
    emu_library()
    for ours, theirs in ((networks.VNet3d(1, 1), nets.VNet3d(1, 1)), (networks.UNet3d(1, 4), nets.UNet3d(1, 4)),
                         (networks.VNet2d(1, 2), nets.VNet2d(1, 2)), (networks.UNet2d(3, 1), nets.UNet2d(3, 1))):
        a, b = ours.state_dict(), theirs.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape for k in a)
        # same leaf module types in the same order -> model.apply(initialize_weights) visits them identically
        ta = [type(m).__name__ for m in ours.modules() if not list(m.children())]
        tb = [type(m).__name__ for m in theirs.modules() if not list(m.children()) and type(m).__name__ not in ("ReLU", "Dropout3d", "Dropout2d", "MaxPool3d", "MaxPool2d")]
        assert ta == tb


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree only exists in the build container")
def test_initialize_weights_draws_the_reference_values():
    """A9 (networks/__init__.py:11-26) at the VALUE level: the same seed followed by `apply(initialize_weights)` gives the live reference
    module and this package's module identical state_dicts (same leaf order, same fan computation, same RNG consumption) - what makes a
    training run started here reproduce a run started on the reference."""
    nets, _, _ = ref_loader.load()
    from tests.conftest import emu_library
    emu_library()
    for seed, (ours, theirs) in enumerate(((networks.VNet3d(1, 1, dtype="f32"), nets.VNet3d(1, 1)), (networks.UNet3d(1, 4, dtype="f32"), nets.UNet3d(1, 4)),
                                           (networks.VNet2d(3, 2, dtype="f32"), nets.VNet2d(3, 2)), (networks.UNet2d(1, 1, dtype="f32"), nets.UNet2d(1, 1)))):
        torch.manual_seed(100 + seed)
        theirs.apply(nets.initialize_weights)
        torch.manual_seed(100 + seed)
        ours.apply(networks.initialize_weights)
        a, b = ours.state_dict(), theirs.state_dict()
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert torch.equal(a[k].cpu(), b[k]), k
        # and the values really are in the engine's flat buffer (the views alias it), not in detached copies
        w0 = next(iter(a))
        assert torch.equal(ours.engine.param_view(w0).cpu().reshape(b[w0].shape), b[w0])


@pytest.mark.parametrize("shape,c", [((2, 1, 5, 6, 7), 1), ((3, 4, 9, 11), 4), ((1, 3, 4, 4, 6), 3)])
def test_predict_mask_exact(dev, shape, c):
    """device threshold / argmax (predict post-processing) == numpy on the same probabilities, ties included."""
    import numpy as np
    from pytorchdeeplearing_amd.metric import predict_mask
    g = torch.Generator().manual_seed(3)
    p = torch.rand(shape, generator=g)
    p[0, :, 0] = 0.5                                      # exact ties: threshold is strict, argmax takes the first maximum
    got = predict_mask(p.to(dev), 0.5, 255).cpu().numpy()
    if c == 1:
        want = ((p[:, 0].numpy() > 0.5) * 255).astype(np.uint8)
    else:
        want = np.argmax(p.numpy(), axis=1).astype(np.uint8)
    assert got.dtype == np.uint8 and np.array_equal(got, want)


def test_product_library_exports_every_symbol_the_header_declares():
    """The gfx950 library loads on the GPU-less build box and exports every entry point of include/segengine.h; the ctypes
    table (_capi.SIGNATURES) binds exactly that set (no compute calls here)."""
    import ctypes
    import re
    from pytorchdeeplearing_amd import _capi, build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "segengine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(seg_[a-z0-9_]+)\s*\(", text))
    assert len(declared) > 40
    dll = ctypes.CDLL(build.build())
    missing = [n for n in sorted(declared) if not hasattr(dll, n)]
    assert not missing, missing
    unbound = sorted(declared - set(_capi.SIGNATURES))
    undeclared = sorted(set(_capi.SIGNATURES) - declared)
    assert not unbound and not undeclared, (unbound, undeclared)


def test_bench_and_product_do_not_touch_the_oracle_outside_the_cpu_baseline_leg():
    """oracle/ is test infrastructure: the package never imports it, and bench.py only inside cpu_baseline()."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _dirs, files in os.walk(os.path.join(root, "pytorchdeeplearing_amd")):
        for f in files:
            if f.endswith(".py"):
                tree = ast.parse(open(os.path.join(dirpath, f)).read())
                for node in ast.walk(tree):
                    names = [a.name for a in node.names] if isinstance(node, ast.Import) else \
                            [node.module or ""] if isinstance(node, ast.ImportFrom) else []
                    assert not any(n == "oracle" or n.startswith("oracle.") for n in names), (f, names)
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mod = node.module if isinstance(node, ast.ImportFrom) else node.names[0].name
                if mod and mod.split(".")[0] == "oracle":
                    assert fn.name == "cpu_baseline", fn.name
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            mod = node.module if isinstance(node, ast.ImportFrom) else node.names[0].name
            assert not (mod and mod.split(".")[0] == "oracle")


def test_oracle_ssim_equals_reference_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "ssim.npz"))
    for tag, ka, kb, win in (("ssim2d", "a2", "b2", 11), ("ssim2d_w7", "a2", "b2", 7), ("ssim3d", "a3", "b3", 11)):
        a, b = torch.from_numpy(G[ka]).requires_grad_(True), torch.from_numpy(G[kb]).requires_grad_(True)
        v = seg.ssim_oracle(a, b, win)
        v.backward()
        assert abs(float(v) - float(G["val_" + tag])) < 1e-6, tag
        np.testing.assert_allclose(a.grad.numpy(), G["g1_" + tag], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(b.grad.numpy(), G["g2_" + tag], rtol=1e-4, atol=1e-9)


def test_ssim_modules_vs_reference_golden(dev, golden_dir):
    """model/lossesSSIM.py (SSIM, SSIM3D, ssim with another window, size_average=False) on the device against goldens written by the reference
    module itself: values 1e-5, gradients with respect to BOTH images (separable windows: rounding order differs from the 11^d window)."""
    from pytorchdeeplearing_amd import lossesSSIM as S
    from model import lossesSSIM as shim
    assert shim.SSIM3D is S.SSIM3D and shim.ssim is S.ssim
    G = np.load(os.path.join(golden_dir, "ssim.npz"))
    for tag, f, ka, kb in (("ssim2d", S.SSIM(window_size=11, size_average=True), "a2", "b2"),
                           ("ssim2d_w7", lambda p, q: S.ssim(p, q, window_size=7), "a2", "b2"),
                           ("ssim3d", S.SSIM3D(window_size=11), "a3", "b3")):
        a, b = torch.from_numpy(G[ka]).to(dev).requires_grad_(True), torch.from_numpy(G[kb]).to(dev).requires_grad_(True)
        v = f(a, b)
        v.backward()
        assert abs(float(v) - float(G["val_" + tag])) < 1e-5, (tag, float(v), float(G["val_" + tag]))
        for got, want in ((a.grad, G["g1_" + tag]), (b.grad, G["g2_" + tag])):
            err = np.abs(got.cpu().numpy() - want).max()
            assert err < 2e-3 * np.abs(want).max() + 1e-9, (tag, err, np.abs(want).max())
    a = torch.from_numpy(G["a2"]).to(dev).requires_grad_(True)
    v = S.SSIM(window_size=11, size_average=False)(a, torch.from_numpy(G["b2"]).to(dev))
    np.testing.assert_allclose(v.detach().cpu().numpy(), G["val_ssim2d_persample"], rtol=0, atol=1e-5)
    (v * torch.tensor([1.0, -2.0], device=dev)).sum().backward()
    want = G["g1_ssim2d_persample"]
    assert np.abs(a.grad.cpu().numpy() - want).max() < 2e-3 * np.abs(want).max()
    # ssim3D(size_average=False): the reference's (N, W) result (means over channel, depth, height), weighted sum differentiated w.r.t. both volumes
    a = torch.from_numpy(G["a3"]).to(dev).requires_grad_(True)
    b = torch.from_numpy(G["b3"]).to(dev).requires_grad_(True)
    v = S.ssim3D(a, b, window_size=11, size_average=False)
    assert tuple(v.shape) == tuple(G["val_ssim3d_cols"].shape)
    np.testing.assert_allclose(v.detach().cpu().numpy(), G["val_ssim3d_cols"], rtol=0, atol=1e-5)
    tot = (v * torch.from_numpy(G["w_ssim3d_cols"]).to(dev)).sum()
    tot.backward(retain_graph=True)
    for got, want in ((a.grad, G["g1_ssim3d_cols"]), (b.grad, G["g2_ssim3d_cols"])):
        assert np.abs(got.cpu().numpy() - want).max() < 2e-3 * np.abs(want).max() + 1e-9
    # a second traversal of the same node (retain_graph) returns the same gradient: the backward pass rebuilds the maps it consumed
    g_first = a.grad.clone()
    a.grad = None
    tot.backward()
    assert float((a.grad - g_first).abs().max()) <= 1e-6 * float(g_first.abs().max())


@pytest.mark.parametrize("name", ["BinaryDiceLoss", "BinaryCrossEntropyDiceLoss", "BinaryFocalLoss", "BinaryTverskyLoss", "BinarySSLoss", "BinaryJaccardLoss"])
def test_binary_losses_accept_multi_label_heads(dev, golden_dir, name):
    """the reference's Binary* classes accept a multi-label [N, C > 1, ...] prediction with a same-shaped target
    (`view(bs, num_classes, -1)`, model/losses.py:43-53) and reduce over every element; so do the modules here (goldens from the reference classes)."""
    G = np.load(os.path.join(golden_dir, "losses_extra.npz"))
    z = torch.from_numpy(G["zml"]).to(dev).requires_grad_(True)
    t = torch.from_numpy(G["yml"]).to(dev)
    v = getattr(losses, name)()(z, t)
    v.backward()
    assert abs(float(v) - float(G["mlloss_" + name])) < 2e-6, name
    np.testing.assert_allclose(z.grad.cpu().numpy(), G["mlgrad_" + name], rtol=3e-4, atol=1e-9, err_msg=name)
    assert z.grad.shape == z.shape

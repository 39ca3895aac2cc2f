"""EXPERIMENT (not part of the package or of the `tests/` suite; run with `python -m pytest tools/experiments/test_lanes.py -m gpu -p conftest --rootdir tests`
or simply with tests/ on PYTHONPATH).  Round 1 measured batch lanes 15-50 % SLOWER than one engine (7.6 / 10 ms vs 6.6 ms per step, DESIGN history):
Intra-GPU batch lanes (tools/experiments/lanes.py): splitting the batch into independent
lanes with shared parameters, a full-batch loss and summed gradients is the SAME optimisation step as
one engine at the full batch size."""
import pytest
import torch

from oracle import seg_oracle as seg
from pytorchdeeplearing_amd import SegEngine, _capi
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import dev  # noqa: F401,E402  (the fixture)
from lanes import LaneEngine  # noqa: E402


@pytest.mark.parametrize("kind,ndim,shape,ncls,loss", [("vnet", 2, (4, 1, 16, 16), 1, "BinaryDiceLoss"),
                                                       pytest.param("unet", 2, (3, 1, 16, 16), 3, "MutilDiceLoss", marks=pytest.mark.gpu)])
def test_lanes_equal_single_engine(dev, kind, ndim, shape, ncls, loss):
    if dev.type == "cpu":
        pytest.skip("batch lanes are an opt-in (measured slower) feature: checked on the GPU run only, the host checker needs ~1 min for it")
    params = seg.perturb_params(seg.init_params(kind, ndim, shape[1], ncls, seed=0), seed=7)
    x, y = seg.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=1)
    x, y = x.to(dev), y.to(dev)
    alpha = torch.ones(ncls, device=dev)
    single = SegEngine(kind, ndim, shape[1], ncls, dtype="f32", device=dev)
    lanes = LaneEngine(kind, ndim, shape[1], ncls, dtype="f32", device=dev, lanes=2)
    single.load_state_dict(params)
    lanes.load_state_dict(params)
    on_gpu = torch.device(dev).type == "cuda"
    for it in range(2 if on_gpu else 1):          # the host-side checker is ~1000x slower: one step there
        g = torch.Generator().manual_seed(40 + it)
        masks = seg.draw_masks(kind, shape[0], generator=g)
        o1 = single.train_step(x, y, loss, class_alpha=alpha, mask_mode=_capi.MASKS_GIVEN, masks=masks).clone()
        o2 = lanes.train_step(x, y, loss, class_alpha=alpha, mask_mode=_capi.MASKS_GIVEN, masks=masks).clone()
        assert abs(float(o1[0]) - float(o2[0])) < 1e-5 and abs(float(o1[1]) - float(o2[1])) < 1e-6
    a, b = single.state_dict(), lanes.state_dict()
    tot = bad = 0
    for k in a:
        d = (a[k] - b[k]).abs()
        tot += d.numel()
        bad += int((d > 1e-5).sum())        # Adam turns fp32-noise-level gradients into O(lr) moves; those are rare
        assert float(d.max()) < 4e-3, k
    assert bad <= 0.005 * tot
    # random-mask mode runs and keeps the lanes' parameters shared
    if on_gpu:
        lanes.train_step(x, y, loss, class_alpha=alpha)
    assert lanes.engines[1].params.data_ptr() == lanes.engines[0].params.data_ptr()

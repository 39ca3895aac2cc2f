"""Pre/post-processing around predict: HIP kernels through the C-ABI vs the CPU restatement
(oracle/prepost_oracle.py) and the golden vectors produced by the reference's own `normalize` (tests/golden/prepost.npz).
Every test runs on the host-side execution-model checker ('emu') and, under -m gpu, on the real library."""
import os

import numpy as np
import pytest
import torch

from oracle import prepost_oracle as po
from pytorchdeeplearing_amd import prepost as pp


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "prepost.npz"))


def test_oracle_normalize_equals_reference_golden(golden):
    for k in "abcde":
        r = po.normalize(golden["normalize_in_" + k].copy())
        assert r.dtype == golden["normalize_out_" + k].dtype
        np.testing.assert_array_equal(r, golden["normalize_out_" + k])


def test_oracle_linear_resample_agrees_with_scipy_inside_the_buffer():
    from scipy import ndimage
    rs = np.random.RandomState(0)
    v = rs.randn(7, 9, 11).astype(np.float32)
    out_size, step = (5, 12, 8), (7 / 5, 9 / 12, 11 / 8)
    r = po.itk_resample(v, out_size, step)
    coords = np.meshgrid(*[np.arange(o) * s for o, s in zip(out_size, step)], indexing="ij")
    ref = ndimage.map_coordinates(v.astype(np.float64), coords, order=1, mode="nearest")
    inside = np.ones(out_size, bool)
    for c, n in zip(coords, v.shape):
        inside &= c < n - 0.5
    np.testing.assert_allclose(r[inside], ref[inside], rtol=0, atol=1e-5)
    assert (r[~inside] == 0).all()


@pytest.mark.parametrize("k", list("abcde"))
def test_normalize_percentile_vs_reference_golden(dev, golden, k):
    x = golden["normalize_in_" + k]
    got = pp.normalize_percentile(_t(x, dev)).cpu().numpy()
    ref = golden["normalize_out_" + k]
    # mean / std are accumulated in fp64 here and pairwise in fp32 by numpy: a few fp32 ulps on values of magnitude <= ~3
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    if k in "cd":                     # early-return branches hand the clipped volume back untouched: bit-exact
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (37, 2), (4099, 3), (70001, 4)])
def test_percentile_bounds_are_exact_order_statistics(dev, n, seed):
    """constant-free volumes of awkward sizes, heavy ties and negative values: the clip bounds found by the radix select
    must be np.percentile's, bit for bit (checked through the clipped extremes of the early-return branch)."""
    rs = np.random.RandomState(seed)
    x = np.round(rs.randn(n) * 3.0).astype(np.float32)          # ties everywhere, +-0
    x3 = x.reshape(1, 1, n)
    got = pp.normalize_percentile(_t(x3, dev)).cpu().numpy().ravel()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = po.normalize(x3.copy()).ravel()
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6)
    t, b = np.percentile(x, 5), np.percentile(x, 95)
    if t == b:
        np.testing.assert_array_equal(got, np.clip(x, t, b))


def test_normalize_percentile_full_size_properties(dev):
    if dev.type == "cpu":
        n = (24, 24, 24)
    else:
        n = (160, 160, 160)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, generator=g) * 200.0
    x[:, : n[1] // 4] = 0.0
    got = pp.normalize_percentile(x.to(dev)).cpu().numpy()
    xn = x.numpy()
    t, b = np.percentile(xn, 5), np.percentile(xn, 95)
    c = np.clip(xn, t, b)
    nz = c[c != 0].astype(np.float64)
    ref = (c - nz.mean()) / nz.std()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)
    # the z-scored non-zero voxels have zero mean / unit variance
    z = got[c != 0].astype(np.float64)
    assert abs(z.mean()) < 1e-5 and abs(z.std() - 1.0) < 1e-5


@pytest.mark.parametrize("clip", [True, False])
def test_normalize_meanstd_vs_oracle(dev, clip):
    rs = np.random.RandomState(1)
    x = (rs.randn(11, 13, 17) * 400.0 - 300.0).astype(np.float32)
    lo, hi = (-100.0, 100.0) if clip else (None, None)
    got = pp.normalize_meanstd(_t(x, dev), lo, hi).cpu().numpy()
    ref = po.truncated_meanstd(x, hi, lo)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    assert abs(float(got.astype(np.float64).mean())) < 1e-6
    assert abs(float(got.astype(np.float64).std(ddof=1)) - 1.0) < 1e-6


@pytest.mark.parametrize("in_size,out_size", [((7, 9, 11), (5, 12, 8)), ((6, 6, 6), (6, 6, 6)), ((5, 4, 3), (16, 16, 16)),
                                              ((20, 18, 16), (3, 2, 1)), ((1, 8, 8), (4, 4, 4))])
def test_resample_linear_and_nearest_vs_oracle(dev, in_size, out_size):
    rs = np.random.RandomState(sum(in_size))
    v = (rs.randn(*in_size) * 100.0).astype(np.float32)
    step = tuple(i / o for i, o in zip(in_size, out_size))
    lin = pp.resample3d(_t(v, dev), out_size).cpu().numpy()
    np.testing.assert_allclose(lin, po.itk_resample(v, out_size, step), rtol=0, atol=1e-4)
    near = pp.resample3d(_t(v, dev), out_size, mode=pp.NEAREST).cpu().numpy()
    np.testing.assert_array_equal(near, po.itk_resample(v, out_size, step, nearest=True))
    m = (rs.rand(*in_size) > 0.5).astype(np.uint8) * 255
    mk = pp.resample3d(_t(m, dev), out_size, mode=pp.NEAREST).cpu().numpy()
    assert mk.dtype == np.uint8
    np.testing.assert_array_equal(mk, po.itk_resample(m, out_size, step, nearest=True))
    if in_size == out_size:            # identity resample returns the input
        np.testing.assert_array_equal(lin, v)
        np.testing.assert_array_equal(near, v)


def test_resample_by_spacing_and_mask_round_trip(dev):
    """resize_image_itk sizes (utils.py:123-145) and the mask's way back: down by an integer factor then up again with
    nearest-neighbour reproduces a block-constant mask exactly."""
    size, step = pp.spacing_resample_size((12, 16, 20), (2.0, 2.0, 2.0), (1.0, 1.0, 1.0))
    assert size == (6, 8, 10) and step == (2.0, 2.0, 2.0)
    rs = np.random.RandomState(4)
    small = (rs.rand(6, 8, 10) > 0.5).astype(np.uint8)
    big = np.kron(small, np.ones((2, 2, 2), np.uint8))
    down = pp.resample3d(_t(big, dev), size, step, mode=pp.NEAREST)
    np.testing.assert_array_equal(down.cpu().numpy(), small)
    up = pp.resample3d(down, (12, 16, 20), (0.5, 0.5, 0.5), mode=pp.NEAREST).cpu().numpy()
    ref = po.itk_resample(small, (12, 16, 20), (0.5, 0.5, 0.5), nearest=True)
    np.testing.assert_array_equal(up, ref)


def test_resample_rejects_bad_arguments(dev):
    v = torch.zeros((4, 4, 4), dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError):
        pp.resample3d(v, (2, 2, 2), mode=pp.LINEAR)          # linear needs float volumes
    with pytest.raises(RuntimeError):
        pp.resample3d(v.float(), (2, 2, 2), step=(0.0, 1.0, 1.0))


@pytest.mark.parametrize("vol_shape,patch", [((20, 24, 28), (8, 8, 8)), ((8, 8, 8), (8, 8, 8)), ((9, 30, 17), (8, 16, 16))])
def test_patch_origins_gather_stitch_equal_the_reference_loop(dev, vol_shape, patch):
    rs = np.random.RandomState(7)
    vol = rs.randn(*vol_shape).astype(np.float32)
    thr = 0.8

    def predict(p):                       # stand-in network: a voxel-wise rule, so windows are comparable
        return ((p[0] > thr) * 255).astype(np.uint8)

    ref = po.patch_loop(vol[None], patch, predict)
    origins = pp.patch_origins(vol_shape, patch)
    o = torch.tensor(origins, dtype=torch.int32, device=dev)
    batch = pp.gather_patches(_t(vol, dev), o, patch)
    assert tuple(batch.shape) == (len(origins), 1) + tuple(patch)
    for i, (z, y, x) in enumerate(origins):
        np.testing.assert_array_equal(batch[i, 0].cpu().numpy(), vol[z:z + patch[0], y:y + patch[1], x:x + patch[2]])
    masks = ((batch[:, 0] > thr) * 255).to(torch.uint8)
    out = torch.zeros(vol_shape, dtype=torch.uint8, device=dev)
    pp.stitch_mask(masks, o, out)
    np.testing.assert_array_equal(out.cpu().numpy(), ref.astype(np.uint8))


def test_patch_origins_rejects_small_volumes():
    with pytest.raises(ValueError):
        pp.patch_origins((4, 20, 20), (8, 8, 8))


def test_prepost_rejects_empty_volumes_and_oversized_patches(dev):
    """error behaviour of the C-ABI entry points (no kernel is launched): int return < 0 -> RuntimeError with seg_last_error()."""
    empty = torch.zeros((0, 4, 4), dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError):
        pp.normalize_meanstd(empty, -1.0, 1.0)
    with pytest.raises(RuntimeError):
        pp.normalize_percentile(empty)
    v = torch.zeros((4, 4, 4), dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError):
        pp.normalize_meanstd(v, 1.0, -1.0)                      # lower > upper
    o = torch.zeros((1, 3), dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        pp.gather_patches(v, o, (8, 8, 8))                      # patch larger than the volume
    with pytest.raises(RuntimeError):
        pp.stitch_mask(torch.zeros((1, 8, 8, 8), dtype=torch.uint8, device=dev), o, torch.zeros((4, 4, 4), dtype=torch.uint8, device=dev))

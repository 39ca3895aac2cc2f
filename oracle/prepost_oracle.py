"""TEST INFRASTRUCTURE ONLY (never imported by the product path): CPU restatement of the pre/post-processing around
`predict`.

* `normalize`, the clip + mean/std of `ConvertitkTrunctedValue` and the patch loop are plain numpy in the reference
  (dataprocess/utils.py:148-204, model/modelUnet.py:718-746): `normalize` is pinned by tests/golden/prepost.npz, generated
  by executing the reference's own function source (oracle/make_golden.py:prepost_golden).
* The resampling itself lives in a third-party dependency that is ABSENT here: SimpleITK (un-pinned; README.md:13 lists
  "SimpleITK").  `itk_resample` restates the published algorithm of itk::ResampleImageFilter + Linear/NearestNeighbor
  InterpolateImageFunction for the identity transform the reference uses; **parity unpinned** for this piece (no ITK in
  the image to generate vectors from) — cross-checked against scipy.ndimage.map_coordinates inside the buffer.
* itk::NormalizeImageFilter: (x - mean) / sigma with the unbiased (N-1) variance, double accumulators."""
import numpy as np


def itk_resample(vol, out_size, step, nearest=False):
    """vol (D,H,W); output voxel i along an axis samples continuous input index i*step (dataprocess/utils.py:99-145)."""
    vol = np.asarray(vol)
    idx = [np.arange(o, dtype=np.float64) * float(s) for o, s in zip(out_size, step)]
    cz, cy, cx = np.meshgrid(*idx, indexing="ij")
    inside = np.ones(cz.shape, bool)
    for c, n in zip((cz, cy, cx), vol.shape):
        inside &= (c >= -0.5) & (c < n - 0.5)            # itk::ImageFunction::IsInsideBuffer
    out = np.zeros(cz.shape, vol.dtype)
    if nearest:
        i = [np.clip(np.floor(c + 0.5).astype(np.int64), 0, n - 1) for c, n in zip((cz, cy, cx), vol.shape)]   # RoundHalfIntegerUp
        out[inside] = vol[i[0], i[1], i[2]][inside]
        return out
    b0 = [np.clip(np.floor(c).astype(np.int64), 0, n - 1) for c, n in zip((cz, cy, cx), vol.shape)]
    d = [np.maximum(c - b, 0.0) for c, b in zip((cz, cy, cx), b0)]
    b1 = [np.minimum(b + 1, n - 1) for b, n in zip(b0, vol.shape)]
    v = vol.astype(np.float64)
    g = lambda z, y, x: v[z, y, x]
    a00 = g(b0[0], b0[1], b0[2]) + d[2] * (g(b0[0], b0[1], b1[2]) - g(b0[0], b0[1], b0[2]))
    a01 = g(b0[0], b1[1], b0[2]) + d[2] * (g(b0[0], b1[1], b1[2]) - g(b0[0], b1[1], b0[2]))
    a10 = g(b1[0], b0[1], b0[2]) + d[2] * (g(b1[0], b0[1], b1[2]) - g(b1[0], b0[1], b0[2]))
    a11 = g(b1[0], b1[1], b0[2]) + d[2] * (g(b1[0], b1[1], b1[2]) - g(b1[0], b1[1], b0[2]))
    c0 = a00 + d[1] * (a01 - a00)
    c1 = a10 + d[1] * (a11 - a10)
    res = c0 + d[0] * (c1 - c0)
    out[inside] = res[inside].astype(vol.dtype)
    return out


def truncated_meanstd(vol, upper=None, lower=None):
    """ConvertitkTrunctedValue(image, upper, lower, 'meanstd') (dataprocess/utils.py:148-179)."""
    a = np.asarray(vol, np.float32).copy()
    if upper is not None:
        a[a > upper] = upper
        a[a < lower] = lower
    a64 = a.astype(np.float64)
    mean, sigma = a64.mean(), a64.std(ddof=1)
    return ((a64 - mean) * (1.0 / sigma)).astype(np.float32)


def normalize(slice, bottom=95, down=5):
    """dataprocess/utils.py:182-204, restated."""
    b = np.percentile(slice, bottom)
    t = np.percentile(slice, down)
    slice = np.clip(slice, t, b)
    image_nonzero = slice[np.nonzero(slice)]
    if np.std(slice) == 0 or np.std(image_nonzero) == 0:
        return slice
    return (slice - np.mean(image_nonzero)) / np.std(image_nonzero)


def patch_loop(vol, patch_shape, predict):
    """model/modelUnet.py:718-746 statement by statement; predict(patch (1,d,h,w)) -> uint8 mask (d,h,w)."""
    _, D, H, W = vol.shape
    pd, ph, pw = patch_shape
    out_mask = np.zeros((D, H, W))
    for z in range(0, D, pd // 2):
        for y in range(0, H, ph // 2):
            for x in range(0, W, pw // 2):
                x_min = x * pw
                x_max = (x + 1) * pw
                if x_max > W:
                    x_max = W
                    x_min = W - pw
                y_min = y * ph
                y_max = (y + 1) * ph
                if y_max > H:
                    y_max = H
                    y_min = H - ph
                z_min = z * pd
                z_max = (z + 1) * pd
                if z_max > D:
                    z_max = D
                    z_min = D - pd
                patch = vol[:, z_min:z_max, y_min:y_max, x_min:x_max]
                out_mask[z_min:z_max, y_min:y_max, x_min:x_max] = out_mask[z_min:z_max, y_min:y_max, x_min:x_max] + predict(patch).copy()
    out_mask[out_mask != 0] = 1
    return out_mask

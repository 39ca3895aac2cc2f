"""Device-side pre/post-processing either side of `predict`: the SimpleITK resampling, the two
intensity normalisations and the sliding-window crop / stitch of the reference's `inference` / `inference_patch`
(model/modelVNet.py:678-698, model/modelUnet.py:684-763, dataprocess/utils.py:99-204), as HIP kernels behind the C-ABI
(`seg_op_resample3d`, `seg_op_normalize_*`, `seg_op_gather_patches`, `seg_op_stitch_mask`).  Volumes are (D, H, W)
torch tensors on the engine's device; no CPU fallback (a CPU tensor raises unless the test-only checker is injected)."""
import numpy as np
import torch

from . import _capi
from .engine import aligned_empty

LINEAR, NEAREST = 0, 1


def _vol(t):
    assert t.dim() == 3, "expected a (D, H, W) volume"
    return t.contiguous()


def resample3d(vol, out_size, step=None, mode=LINEAR):
    """ITK ResampleImageFilter with the identity transform (dataprocess/utils.py:99-145).  out_size = (D, H, W) of the
    result; step = continuous-index step per axis (output spacing / input spacing); default = in_size / out_size, which
    is `resize_image_itkwithsize`.  float32 volumes (linear or nearest) or uint8 masks (nearest)."""
    vol = _vol(vol)
    if vol.dtype not in (torch.float32, torch.uint8):
        vol = vol.float()
    out_size = tuple(int(s) for s in out_size)
    if step is None:
        step = tuple(float(i) / float(o) for i, o in zip(vol.shape, out_size))
    lib = _capi.lib_for(vol.device)
    out = torch.empty(out_size, dtype=vol.dtype, device=vol.device)
    lib.check(lib.seg_op_resample3d(vol.data_ptr(), out.data_ptr(), 0 if vol.dtype == torch.float32 else 1, *vol.shape, *out_size,
                                    float(step[0]), float(step[1]), float(step[2]), int(mode), _capi.stream_for(vol.device)),
              "seg_op_resample3d")
    return out


def spacing_resample_size(in_size, new_spacing, origin_spacing):
    """`resize_image_itk` (dataprocess/utils.py:123-145): newSize = (originSize / (newSpacing / originSpacing)).astype(int);
    returns (out_size, step) per axis in the order the sizes are given."""
    factor = np.array(new_spacing, float) / np.array(origin_spacing, float)
    new_size = (np.array(in_size) / factor).astype(int)
    return tuple(int(s) for s in new_size), tuple(float(f) for f in factor)


def _ws(device):
    lib = _capi.lib_for(device)
    return lib, aligned_empty(lib.seg_op_normalize_ws_bytes(), device)


def normalize_meanstd(vol, lower=None, upper=None):
    """ConvertitkTrunctedValue(image, upper, lower, 'meanstd') (dataprocess/utils.py:148-179)."""
    x = vol.float().contiguous()
    lib, ws = _ws(x.device)
    out = torch.empty_like(x)
    clip = lower is not None and upper is not None
    lib.check(lib.seg_op_normalize_meanstd(x.data_ptr(), out.data_ptr(), x.numel(), 1 if clip else 0, float(lower if clip else 0.0),
                                           float(upper if clip else 0.0), ws.data_ptr(), _capi.stream_for(x.device)), "seg_op_normalize_meanstd")
    return out


def normalize_percentile(vol, bottom=95, down=5):
    """normalize(slice, bottom=95, down=5) (dataprocess/utils.py:182-204)."""
    x = vol.float().contiguous()
    lib, ws = _ws(x.device)
    out = torch.empty_like(x)
    lib.check(lib.seg_op_normalize_percentile(x.data_ptr(), out.data_ptr(), x.numel(), float(down), float(bottom), ws.data_ptr(),
                                              _capi.stream_for(x.device)), "seg_op_normalize_percentile")
    return out


def patch_origins(vol_shape, patch_shape):
    """Window origins of the reference's sliding loop, statement by statement (model/modelUnet.py:718-743).  The loop
    variables already advance in half-patch steps and are then multiplied by the patch size once more, so every window
    after the first is clamped to the far border: the set of windows is {0, size - patch} per axis.  Reproduced as is."""
    D, H, W = (int(s) for s in vol_shape)
    pd, ph, pw = (int(s) for s in patch_shape)
    if pd > D or ph > H or pw > W:
        raise ValueError("inference_patch: the resampled volume %s is smaller than the network patch %s" % ((D, H, W), (pd, ph, pw)))
    seen, out = set(), []
    for z in range(0, D, pd // 2):
        for y in range(0, H, ph // 2):
            for x in range(0, W, pw // 2):
                x_min, x_max = x * pw, (x + 1) * pw
                if x_max > W:
                    x_max, x_min = W, W - pw
                y_min, y_max = y * ph, (y + 1) * ph
                if y_max > H:
                    y_max, y_min = H, H - ph
                z_min, z_max = z * pd, (z + 1) * pd
                if z_max > D:
                    z_max, z_min = D, D - pd
                o = (z_min, y_min, x_min)
                if o not in seen:            # the stitch is idempotent (|=), repeated windows add nothing
                    seen.add(o)
                    out.append(o)
    return out


def gather_patches(vol, origins, patch_shape):
    """(D,H,W) float32 volume + int32 origins (nb, 3) on the device -> (nb, 1, pd, ph, pw) network batch."""
    vol = _vol(vol.float())
    lib = _capi.lib_for(vol.device)
    nb = origins.shape[0]
    out = torch.empty((nb, 1) + tuple(patch_shape), dtype=torch.float32, device=vol.device)
    lib.check(lib.seg_op_gather_patches(vol.data_ptr(), *vol.shape, origins.data_ptr(), nb, *patch_shape, out.data_ptr(),
                                        _capi.stream_for(vol.device)), "seg_op_gather_patches")
    return out


def stitch_mask(masks, origins, out):
    """out[window] = 1 where the window's uint8 mask is non-zero (`out_mask += patch; out_mask[out_mask != 0] = 1`)."""
    masks = masks.contiguous()
    assert masks.dtype == torch.uint8 and out.dtype == torch.uint8 and out.is_contiguous()
    lib = _capi.lib_for(out.device)
    nb = origins.shape[0]
    pshape = tuple(masks.shape[-3:])
    lib.check(lib.seg_op_stitch_mask(masks.data_ptr(), origins.data_ptr(), nb, *pshape, out.data_ptr(), *out.shape,
                                     _capi.stream_for(out.device)), "seg_op_stitch_mask")
    return out

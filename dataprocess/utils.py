"""Host-side pre/post-processing helpers the wrappers and scripts call (dataprocess/utils.py:99-233 of
the reference).  SimpleITK is optional in this image; the helpers that need it import it lazily."""
import os

import numpy as np


def file_name_path(file_dir, dir=True, file=False):
    """sub-directories (dir=True) or files (file=True) of the first non-empty level (utils.py:221-233)."""
    for root, dirs, files in os.walk(file_dir):
        if len(dirs) and dir:
            print("sub_dirs:", dirs)
            return dirs
        if len(files) and file:
            print("files:", files)
            return files


def normalize(slice, bottom=95, down=5):
    """percentile clip + z-score over the non-zero voxels (utils.py:182-204; the `tmp == tmp.min() -> -9` line is
    commented out in the reference and is not applied).  Host-side numpy form; the wrappers use the device kernel
    (pytorchdeeplearing_amd.prepost.normalize_percentile)."""
    b, t = np.percentile(slice, bottom), np.percentile(slice, down)
    slice = np.clip(slice, t, b)
    image_nonzero = slice[np.nonzero(slice)]
    if np.std(slice) == 0 or np.std(image_nonzero) == 0:
        return slice
    return (slice - np.mean(image_nonzero)) / np.std(image_nonzero)


def _sitk():
    import SimpleITK as sitk
    return sitk


def resize_image_itkwithsize(itkimage, newSize, originSize, resamplemethod=None):
    """resample to a fixed grid size keeping the physical extent (utils.py:99-120)."""
    sitk = _sitk()
    resampler = sitk.ResampleImageFilter()
    originSize, newSize = np.array(originSize), np.array(newSize)
    factor = originSize / newSize
    newSpacing = np.array(itkimage.GetSpacing()) * factor
    resampler.SetReferenceImage(itkimage)
    resampler.SetOutputSpacing(newSpacing.tolist())
    resampler.SetSize(newSize.astype(int).tolist())
    resampler.SetTransform(sitk.Transform(3, sitk.sitkIdentity))
    resampler.SetInterpolator(sitk.sitkNearestNeighbor if resamplemethod is None else resamplemethod)
    out = resampler.Execute(itkimage)
    return sitk.GetArrayFromImage(out), out


def ConvertitkTrunctedValue(image, upper=200, lower=-200, normalize="maxmin"):
    """clip to [lower, upper] then max-min or mean-std normalise (utils.py:148-179)."""
    sitk = _sitk()
    arr = np.clip(sitk.GetArrayFromImage(image).astype(np.float64), lower, upper)
    if normalize == "maxmin":
        arr = (arr - arr.min()) / max(arr.max() - arr.min(), 1e-12)
    elif normalize == "meanstd":
        arr = (arr - arr.mean()) / max(arr.std(ddof=1), 1e-12)      # itk::NormalizeImageFilter: unbiased variance
    out = sitk.GetImageFromArray(arr.astype(np.float32))
    out.SetSpacing(image.GetSpacing()); out.SetOrigin(image.GetOrigin()); out.SetDirection(image.GetDirection())
    return out

"""Image / volume I/O helpers for the model wrappers.  OpenCV, SimpleITK, TensorBoard and
torchsummary are optional: the reference imports them unconditionally (model/modelVNet.py:17-22);
this image has none of them, so every use is soft — PIL stands in for cv2's grey-scale
read / bilinear resize / BMP write, and logging simply skips what is missing."""
import numpy as np

try:
    import cv2
except Exception:  # pragma: no cover
    cv2 = None
try:
    import SimpleITK as sitk
except Exception:  # pragma: no cover
    sitk = None
try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # pragma: no cover
    SummaryWriter = None


def imread_gray(path):
    if cv2 is not None:
        return cv2.imread(path, 0)
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"))


def resize(img, size_wh, nearest=False):
    """cv2.resize(img, (w, h)) semantics."""
    if cv2 is not None:
        return cv2.resize(img, size_wh, interpolation=cv2.INTER_NEAREST if nearest else cv2.INTER_LINEAR)
    from PIL import Image
    a = np.asarray(img)
    mode = "F" if a.dtype.kind == "f" else "L"
    im = Image.fromarray(a.astype(np.float32) if mode == "F" else a.astype(np.uint8), mode=mode)
    out = im.resize(tuple(int(s) for s in size_wh), Image.NEAREST if nearest else Image.BILINEAR)
    return np.asarray(out).astype(a.dtype if a.dtype.kind == "f" else np.uint8)


def imwrite(path, img_u8):
    if cv2 is not None:
        cv2.imwrite(path, img_u8)
        return
    from PIL import Image
    Image.fromarray(np.asarray(img_u8, dtype=np.uint8)).save(path)

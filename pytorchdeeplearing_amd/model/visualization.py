"""Per-epoch picture dumps (model/visualization.py:9-49 of the reference): loss/accuracy curves and
the predicted / ground-truth mask montages."""
import os

import numpy as np

from . import _io


def plot_result(model_dir, H_train, H_validation, H_train_name, H_validation_name, labelname):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # pragma: no cover
        return
    plt.figure()
    plt.plot(H_train, label=H_train_name)
    plt.plot(H_validation, label=H_validation_name)
    plt.title(H_train_name + "," + H_validation_name + " on Dataset")
    plt.xlabel("Epoch #")
    plt.ylabel(labelname)
    plt.legend(loc="lower left")
    plt.savefig(os.path.sep.join([model_dir, H_train_name + "_" + H_validation_name + "plot.png"]))
    plt.close()


def _montage(vol, size):
    h, w = vol.shape[1], vol.shape[2]
    out = np.zeros((h * size[0], w * size[1]))
    for idx, sl in enumerate(vol):
        i, j = idx % size[1], idx // size[1]
        if j >= size[0]:
            break
        out[j * h:j * h + h, i * w:i * w + w] = sl
    return out


def save_images3d(pdmask, gtmask, size, path, pixelvalue=255.0):
    pd = pdmask.detach().cpu().squeeze().numpy()
    gt = gtmask.detach().cpu().squeeze().numpy()
    if pd.ndim == 4:          # multi-class probabilities: show the arg-max label map
        pd = pd.argmax(0)
    _io.imwrite(path + "pdmask.bmp", np.clip(_montage(pd, size) * pixelvalue, 0, 255).astype("uint8"))
    _io.imwrite(path + "gtmask.bmp", np.clip(_montage(gt, size) * pixelvalue, 0, 255).astype("uint8"))


def save_images2d(pdmask, gtmask, path, pixelvalue=255.0):
    pd = pdmask.detach().cpu().squeeze().numpy()
    gt = gtmask.detach().cpu().squeeze().numpy()
    if pd.ndim == 3:
        pd = pd.argmax(0).astype(np.float32)
    elif np.max(gt) == 1:
        pd = (pd > 0.5).astype(np.float32)
    _io.imwrite(path + "pdmask.bmp", np.clip(pd * pixelvalue, 0, 255).astype("uint8"))
    _io.imwrite(path + "gtmask.bmp", np.clip(gt * pixelvalue, 0, 255).astype("uint8"))

"""Script-facing model wrappers: Binary/Mutil x VNet/UNet x 2d/3d `...Model` classes
with the reference's constructor keywords and methods (model/modelVNet.py:24-935, modelUnet.py:24-1001):
`trainprocess`, `predict`, `inference`, `clear_GPU_cache`, `_dataloder`, `_loss_function`,
`_accuracy_function`.  The per-batch body of `trainprocess` (modelVNet.py:570-596) runs as ONE engine
train step — forward, loss + Dice metric, zero_grad, backward, fused Adam(W) — through libsegengine.

Deliberate differences from the reference, all outside the arithmetic:
  * cv2 / SimpleITK / TensorBoard / torchsummary are optional (not installed in this image);
  * weights are saved whenever the epoch-mean validation Dice improves for EVERY wrapper (the
    reference's MutilVNet2dModel saves every epoch because of an indentation slip, modelVNet.py:413-417);
  * `MutilVNet3dModel` feeds index labels straight to the metric (the reference's
    `torch.argmax(y[0], 0)` at modelVNet.py:819 assumes one-hot labels and fails on index labels).
"""
import os
import threading
import time
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader

from .. import _capi, losses as L, metric as M, networks
from . import _io
from .dataset import datasetModelSegwithnpy, datasetModelSegwithopencv
from .pipeline import DevicePrefetcher
from .visualization import plot_result, save_images2d, save_images3d

_BINARY_LOSSES = {"BinaryCrossEntropyLoss", "BinaryDiceLoss", "BinaryCrossEntropyDiceLoss", "BinaryFocalLoss"}
_MULTI_LOSSES = {"MutilCrossEntropyLoss", "MutilFocalLoss", "MutilDiceLoss"}


class _SegModel(object):
    _net = None            # networks class
    _ndim = 3
    _binary = True
    _pth = "model.pth"
    _adamw = True          # modelUnet.py:849 uses Adam for MutilUNet3dModel, AdamW everywhere else
    _focal_gamma = 2
    _mask_scale = 255      # BinaryUNet3dModel.predict returns *1 (modelUnet.py:678)
    _norm3d = "meanstd"    # VNet wrappers z-score (modelVNet.py:681); UNet wrappers use percentile normalize

    def _init(self, dims, image_channel, numclass, batch_size, loss_name, inference, model_path, use_cuda):
        self.batch_size, self.loss_name, self.accuracyname = batch_size, loss_name, "dice"
        if self._ndim == 3:
            self.image_depth, self.image_height, self.image_width = dims
        else:
            self.image_height, self.image_width = dims
        self.image_channel, self.numclass = image_channel, numclass
        self.alpha = 0.25 if self._binary else [1.0] * numclass
        self.gamma = self._focal_gamma
        self.use_cuda = use_cuda
        self.device = torch.device("cuda" if use_cuda else "cpu")
        self.model = self._net(image_channel, numclass)
        self.model.to(device=self.device)
        self._lock = threading.Lock()      # flask_app.py serves one shared model object from request threads
        if not self._binary:
            self.alpha = torch.as_tensor(self.alpha).contiguous().to(self.device)
        if inference:
            print(f"Loading model {model_path}")
            print(f"Using device {self.device}")
            self.model.load_state_dict(torch.load(model_path, map_location=self.device))
            print("Model loaded!")

    # ---- reference helper methods ---------------------------------------------------------------
    def _dataloder(self, images, labels, shuffle=False):
        if self._ndim == 3:
            ds = datasetModelSegwithnpy(images, labels, targetsize=(self.image_channel, self.image_depth, self.image_height, self.image_width))
        else:
            ds = datasetModelSegwithopencv(images, labels, targetsize=(self.image_channel, self.image_height, self.image_width))
        # num_workers=0 as in the reference (modelVNet.py:508); SEGENGINE_LOADER_WORKERS > 0 adds reader processes behind the prefetcher
        workers = int(os.environ.get("SEGENGINE_LOADER_WORKERS", "0"))
        return DataLoader(ds, shuffle=shuffle, batch_size=self.batch_size, num_workers=workers, pin_memory=self.device.type == "cuda")

    def _loss_function(self, lossname):
        if lossname == "BinaryCrossEntropyLoss":
            return L.BinaryCrossEntropyLoss()
        if lossname == "BinaryDiceLoss":
            return L.BinaryDiceLoss()
        if lossname == "BinaryCrossEntropyDiceLoss":
            return L.BinaryCrossEntropyDiceLoss()
        if lossname == "BinaryFocalLoss":
            return L.BinaryFocalLoss()
        if lossname == "MutilCrossEntropyLoss":
            return L.MutilCrossEntropyLoss(alpha=self.alpha)
        if lossname == "MutilFocalLoss":
            return L.MutilFocalLoss(alpha=self.alpha, gamma=self.gamma)
        if lossname == "MutilDiceLoss":
            return L.MutilDiceLoss(alpha=self.alpha)

    def _accuracy_function(self, accuracyname, input, target):
        if accuracyname == "dice":
            return M.dice_coeff(input, target) if self.numclass == 1 else M.multiclass_dice_coeff(input, target)
        if accuracyname == "iou":
            return M.iou_coeff(input, target) if self.numclass == 1 else M.multiclass_iou_coeff(input, target)

    # ---- training -------------------------------------------------------------------------------
    def trainprocess(self, trainimage, trainmask, validationimage, validationmask, model_dir, epochs=50, lr=1e-3, showwind=[8, 8]):
        print("[INFO] training the network...")
        Path(model_dir).mkdir(parents=True, exist_ok=True)
        MODEL_PATH = os.path.join(model_dir, self._pth)
        valid = _BINARY_LOSSES if self._binary else _MULTI_LOSSES
        if self.loss_name not in valid:
            raise ValueError("loss_name %r is not available for this wrapper (choose from %s)" % (self.loss_name, sorted(valid)))
        showpixelvalue = 255.0
        if self.numclass > 1:
            showpixelvalue = showpixelvalue // (self.numclass - 1)
        self.model.apply(networks.initialize_weights)          # always re-initialises (modelVNet.py:546)
        eng = self.model.engine
        eng.init_optimizer()
        eng.binarize_labels = bool(self._binary)               # `label[label != 0] = 1` (modelVNet.py:576) is done by the kernels that read the labels
        train_loader = self._dataloder(trainimage, trainmask, True)
        val_loader = self._dataloder(validationimage, validationmask, True)
        H = {"train_loss": [], "train_accuracy": [], "valdation_loss": [], "valdation_accuracy": []}
        startTime = time.time()
        best_validation_dsc = 0.0
        writer = _io.SummaryWriter(log_dir=model_dir) if _io.SummaryWriter is not None else None
        class_alpha = None if self._binary else self.alpha.float()
        fgamma = float(self.gamma)
        wd = 0.01 if self._adamw else 0.0
        metric_slot = 1 if self.accuracyname == "dice" else 2
        try:
            from tqdm import tqdm
        except Exception:  # pragma: no cover
            tqdm = lambda it: it
        for e in tqdm(range(epochs)):
            self.model.train()
            tl, ta, vl, va = [], [], [], []
            trainshow = True
            # reader thread + pinned staging + copy stream; labels binarised / narrowed to uint8 on the host (model/pipeline.py)
            for x, y in DevicePrefetcher(train_loader, self.device, self._binary):
                out3 = eng.train_step(x, y, self.loss_name, lr=lr, weight_decay=wd, decoupled=self._adamw, focal_alpha=0.25,
                                      focal_gamma=fgamma, class_alpha=class_alpha, mask_mode=_capi.MASKS_RANDOM).clone()
                if trainshow:
                    self._show(eng, y, model_dir + "/" + str(e + 1) + "_Train_EPOCH_", showwind, showpixelvalue)
                    trainshow = False
                tl.append(out3[0])
                ta.append(out3[metric_slot])
            self.model.eval()
            with torch.no_grad():
                for x, y in DevicePrefetcher(val_loader, self.device, self._binary):
                    logits, probs = eng.forward(x, _capi.MASKS_EVAL)
                    out3 = eng.loss_forward(logits, y, self.loss_name, 0.25, fgamma, class_alpha).clone()
                    self._last = (probs, y)
                    self._show(eng, y, model_dir + "/" + str(e + 1) + "_Val_EPOCH_", showwind, showpixelvalue, probs=probs)
                    vl.append(out3[0])
                    va.append(out3[metric_slot])
            avgTrainLoss, avgValidationLoss = torch.mean(torch.stack(tl)), torch.mean(torch.stack(vl))
            avgTrainAccu, avgValidationAccu = torch.mean(torch.stack(ta)), torch.mean(torch.stack(va))
            H["train_loss"].append(avgTrainLoss.cpu().numpy())
            H["valdation_loss"].append(avgValidationLoss.cpu().numpy())
            H["train_accuracy"].append(avgTrainAccu.cpu().numpy())
            H["valdation_accuracy"].append(avgValidationAccu.cpu().numpy())
            print("[INFO] EPOCH: {}/{}".format(e + 1, epochs))
            print("Train loss: {:.5f}, Train accu: {:.5f}，validation loss: {:.5f}, validation accu: {:.5f}".format(
                float(avgTrainLoss), float(avgTrainAccu), float(avgValidationLoss), float(avgValidationAccu)))
            if writer is not None:
                writer.add_scalar("Train/Loss", float(avgTrainLoss), e + 1)
                writer.add_scalar("Train/accu", float(avgTrainAccu), e + 1)
                writer.add_scalar("Valid/loss", float(avgValidationLoss), e + 1)
                writer.add_scalar("Valid/accu", float(avgValidationAccu), e + 1)
                writer.flush()
            if float(avgValidationAccu) > best_validation_dsc:
                best_validation_dsc = float(avgValidationAccu)
                torch.save(self.model.state_dict(), MODEL_PATH)
        print("[INFO] total time taken to train the model: {:.2f}s".format(time.time() - startTime))
        plot_result(model_dir, H["train_loss"], H["valdation_loss"], "train_loss", "valdation_loss", "loss")
        plot_result(model_dir, H["train_accuracy"], H["valdation_accuracy"], "train_accuracy", "valdation_accuracy", "accuracy")
        self.history = H
        self.clear_GPU_cache()

    def _show(self, eng, y, path, showwind, pixelvalue, probs=None):
        try:
            if probs is None:
                probs = eng._last_probs if hasattr(eng, "_last_probs") else None
            if probs is None:
                return
            if self._ndim == 3:
                save_images3d(probs[0], y[0], showwind, path, pixelvalue=pixelvalue)
            else:
                save_images2d(probs[0], y[0], path, pixelvalue=pixelvalue)
        except Exception as ex:  # picture dumps must never kill a training run
            print("[WARN] could not write preview images:", ex)

    # ---- inference ------------------------------------------------------------------------------
    def _predict_device(self, img, out_threshold=0.5):
        """img: float32 device tensor (B, C, [D,] H, W) -> uint8 device mask (B, [D,] H, W); eval forward + the
        threshold / argmax of modelVNet.py:670-676 on the device."""
        self.model.eval()
        with torch.no_grad():
            _, output = self.model(img)
            return M.predict_mask(output.detach(), out_threshold, self._mask_scale)

    def predict(self, full_img, out_threshold=0.5):
        """full_img: ndarray (C,[D,]H,W) -> uint8 mask (modelVNet.py:655-676)."""
        with self._lock:
            self.clear_GPU_cache()
            img = torch.as_tensor(full_img).float().contiguous().unsqueeze(0).to(device=self.device, dtype=torch.float32)
            # one byte per voxel crosses PCIe instead of 4 x numclass
            out_mask = self._predict_device(img, out_threshold)[0].cpu().numpy()
        return np.squeeze(out_mask).astype(np.uint8)

    @staticmethod
    def _volume_of(image, spacing=None):
        """SimpleITK image or plain (D, H, W) ndarray -> (array (z,y,x), spacing (x,y,z), sitk image or None)."""
        if _io.sitk is not None and isinstance(image, _io.sitk.Image):
            return _io.sitk.GetArrayFromImage(image), tuple(image.GetSpacing()), image
        arr = np.asarray(image)
        if arr.ndim != 3:
            raise ValueError("3-D inference expects a SimpleITK image or a (D, H, W) array, got shape %s" % (arr.shape,))
        return arr, tuple(spacing) if spacing is not None else (1.0, 1.0, 1.0), None

    @staticmethod
    def _like(image_sitk, arr):
        """hand the mask back the way the reference does: a SimpleITK image carrying the source geometry, or the array."""
        if image_sitk is None:
            return arr
        out = _io.sitk.GetImageFromArray(arr)
        out.SetOrigin(image_sitk.GetOrigin()); out.SetSpacing(image_sitk.GetSpacing()); out.SetDirection(image_sitk.GetDirection())
        return out

    def inference(self, image, newSize=(96, 96, 96)):
        if self._ndim == 2:
            # modelVNet.py:231-242: resize, /255, predict, resize the mask back
            imageresize = _io.resize(image, (self.image_width, self.image_height)) / 255.0
            h, w = imageresize.shape[0], imageresize.shape[1]
            out_mask = self.predict(np.reshape(imageresize, (1, h, w)))
            return _io.resize(out_mask, image.shape[:2], nearest=False)
        # modelVNet.py:678-698 / modelUnet.py:684-705, 976-997: linear resample to newSize -> normalise -> predict ->
        # nearest-neighbour resample of the mask back to the source grid.  The whole chain runs on the device
        # (pytorchdeeplearing_amd/prepost.py): the volume goes up once, one byte per source voxel comes back.
        # `image`: SimpleITK image (as in the reference) or a (D, H, W) ndarray; newSize is (x, y, z) as in SimpleITK.
        from .. import prepost as PP
        arr, _, image_sitk = self._volume_of(image)
        grid = tuple(int(s) for s in reversed(newSize))
        with self._lock:
            self.clear_GPU_cache()
            vol = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
            resized = PP.resample3d(vol, grid, mode=PP.LINEAR)
            if self._norm3d == "meanstd":
                resized = PP.normalize_meanstd(resized, -100.0, 100.0)        # ConvertitkTrunctedValue(., 100, -100, 'meanstd')
            else:
                resized = PP.normalize_percentile(resized)                    # normalize(.)
            mask = self._predict_device(resized[None, None])[0]
            final = PP.resample3d(mask.reshape(grid), arr.shape, mode=PP.NEAREST).cpu().numpy()
        return self._like(image_sitk, final)

    def inference_patch(self, image, newSpacing=(0.5, 0.5, 0.5), spacing=None):
        """modelUnet.py:707-763: resample to `newSpacing`, clip [-1024, -800] + z-score, run the network over the
        reference's window list in batches of `batch_size`, OR the window masks together, resample the mask back to the
        source spacing and paste it into an array of the source shape.  `spacing` (x, y, z): source spacing when
        `image` is a plain (D, H, W) array."""
        if self._ndim != 3:
            raise AttributeError("inference_patch is a 3-D method (modelUnet.py:707)")
        from .. import prepost as PP
        arr, src_spacing, image_sitk = self._volume_of(image, spacing)
        patch = (self.image_depth, self.image_height, self.image_width)
        size, step = PP.spacing_resample_size(arr.shape, tuple(reversed(newSpacing)), tuple(reversed(src_spacing)))
        with self._lock:
            self.clear_GPU_cache()
            vol = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
            resized = PP.normalize_meanstd(PP.resample3d(vol, size, step, PP.LINEAR), -1024.0, -800.0)
            origins = torch.tensor(PP.patch_origins(size, patch), dtype=torch.int32, device=self.device)
            out_mask = torch.zeros(size, dtype=torch.uint8, device=self.device)
            for i in range(0, origins.shape[0], max(1, int(self.batch_size))):
                o = origins[i:i + max(1, int(self.batch_size))].contiguous()
                masks = self._predict_device(PP.gather_patches(resized, o, patch))
                PP.stitch_mask(masks, o, out_mask)
            back_size, back_step = PP.spacing_resample_size(size, tuple(reversed(src_spacing)), tuple(reversed(newSpacing)))
            back = PP.resample3d(out_mask, back_size, back_step, PP.NEAREST)
            final = torch.zeros(arr.shape, dtype=torch.uint8, device=self.device)
            m = [min(a, b) for a, b in zip(arr.shape, back_size)]
            final[:m[0], :m[1], :m[2]] = back[:m[0], :m[1], :m[2]]
            final = final.cpu().numpy().astype(arr.dtype)          # np.zeros_like(source array) (modelUnet.py:752)
        return self._like(image_sitk, final)

    def clear_GPU_cache(self):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()


def _make2d(name, net, binary, pth, default_loss, gamma=2):
    class Model(_SegModel):
        __doc__ = "model/%s (2-D)" % name
        _net, _ndim, _binary, _pth, _focal_gamma = net, 2, binary, pth, gamma

        def __init__(self, image_height, image_width, image_channel, numclass, batch_size, loss_name=default_loss, inference=False,
                     model_path=None, use_cuda=True):
            self._init((image_height, image_width), image_channel, numclass, batch_size, loss_name, inference, model_path, use_cuda)

        def trainprocess(self, trainimage, trainmask, validationimage, validationmask, model_dir, epochs=50, lr=1e-3):
            return _SegModel.trainprocess(self, trainimage, trainmask, validationimage, validationmask, model_dir, epochs, lr)
    Model.__name__ = Model.__qualname__ = name
    return Model


def _make3d(name, net, binary, pth, default_loss, gamma=2, adamw=True, mask_scale=255, norm="meanstd"):
    class Model(_SegModel):
        __doc__ = "model/%s (3-D)" % name
        _net, _ndim, _binary, _pth, _focal_gamma, _adamw, _mask_scale, _norm3d = net, 3, binary, pth, gamma, adamw, mask_scale, norm

        def __init__(self, image_depth, image_height, image_width, image_channel, numclass, batch_size, loss_name=default_loss,
                     inference=False, model_path=None, use_cuda=True):
            self._init((image_depth, image_height, image_width), image_channel, numclass, batch_size, loss_name, inference, model_path, use_cuda)
    Model.__name__ = Model.__qualname__ = name
    return Model


# constructor signatures / defaults: modelVNet.py:30-31,254-255,473-474,711-712 ; modelUnet.py:30-31,254-255,473-474,785-786
BinaryVNet2dModel = _make2d("BinaryVNet2dModel", networks.VNet2d, True, "BinaryVNet2dModel.pth", "BinaryDiceLoss")
MutilVNet2dModel = _make2d("MutilVNet2dModel", networks.VNet2d, False, "MutilVNet2d.pth", "MutilFocalLoss", gamma=2)
BinaryVNet3dModel = _make3d("BinaryVNet3dModel", networks.VNet3d, True, "BinaryVNet3d.pth", "BinaryDiceLoss")
MutilVNet3dModel = _make3d("MutilVNet3dModel", networks.VNet3d, False, "MutilVNet3d.pth", "MutilFocalLoss", gamma=3)
BinaryUNet2dModel = _make2d("BinaryUNet2dModel", networks.UNet2d, True, "BinaryUNet2d.pth", "BinaryDiceLoss")
MutilUNet2dModel = _make2d("MutilUNet2dModel", networks.UNet2d, False, "MutilUNet2d.pth", "MutilFocalLoss", gamma=2)
BinaryUNet3dModel = _make3d("BinaryUNet3dModel", networks.UNet3d, True, "BinaryUNet3d.pth", "BinaryDiceLoss", mask_scale=1, norm="normalize")
MutilUNet3dModel = _make3d("MutilUNet3dModel", networks.UNet3d, False, "MutilUNet3d.pth", "MutilFocalLoss", gamma=3, adamw=False,
                           norm="normalize")

"""B1 boundary: the script-facing `model` package — constructor keywords, trainprocess on
.npy path arrays, checkpoint file names, predict() output conventions — driven exactly like train.py /
inference.py drive the reference (`from model import *`)."""
import os

import numpy as np
import pytest
import torch


def _make_npy(tmp, n, shape, numclass, seed):
    g = np.random.RandomState(seed)
    imgs, labs = [], []
    for i in range(n):
        ip, lp = os.path.join(tmp, "img%d_%d.npy" % (seed, i)), os.path.join(tmp, "lab%d_%d.npy" % (seed, i))
        np.save(ip, g.randn(*shape).astype(np.float32))
        lab = (g.rand(*shape) > 0.7).astype(np.uint8) * (255 if numclass == 1 else 1)
        if numclass > 1:
            lab = g.randint(0, numclass, shape).astype(np.uint8)
        np.save(lp, lab)
        imgs.append(ip); labs.append(lp)
    return np.array(imgs), np.array(labs)


def test_from_model_import_star_surface():
    ns = {}
    exec("from model import *", ns)
    for name in ("BinaryVNet2dModel", "BinaryVNet3dModel", "MutilVNet2dModel", "MutilVNet3dModel", "BinaryUNet2dModel",
                 "BinaryUNet3dModel", "MutilUNet2dModel", "MutilUNet3dModel", "BinaryResNet3dModel"):
        assert name in ns
    import networks
    from dataprocess.utils import file_name_path  # noqa: F401  (inference.py:4)
    assert hasattr(networks, "VNet3d") and hasattr(networks, "initialize_weights")
    from model.losses import BinaryDiceLoss  # noqa: F401
    from model.metric import dice_coeff  # noqa: F401


@pytest.mark.parametrize("cls,numclass,loss,pth", [("BinaryVNet3dModel", 1, "BinaryDiceLoss", "BinaryVNet3d.pth"),
                                                    ("MutilUNet3dModel", 3, "MutilDiceLoss", "MutilUNet3d.pth")])
def test_trainprocess_and_predict(dev, tmp_path, monkeypatch, cls, numclass, loss, pth):
    import model
    monkeypatch.setenv("SEGENGINE_DTYPE", "f32")
    tmp = str(tmp_path)
    shape = (16, 16, 16)
    if dev.type == "cpu" and cls != "MutilUNet3dModel":
        pytest.skip("host-checker run keeps one wrapper (the GPU run covers both)")
    epochs = 1 if dev.type == "cpu" else 2
    tr_i, tr_l = _make_npy(tmp, 1 if dev.type == "cpu" else 2, shape, numclass, 1)
    va_i, va_l = _make_npy(tmp, 1, shape, numclass, 2)
    m = getattr(model, cls)(image_depth=16, image_height=16, image_width=16, image_channel=1, numclass=numclass, batch_size=1,
                            loss_name=loss, use_cuda=dev.type == "cuda")
    log = os.path.join(tmp, "log")
    m.trainprocess(tr_i, tr_l, va_i, va_l, model_dir=log, epochs=epochs, showwind=[4, 4])
    assert os.path.isfile(os.path.join(log, pth))
    assert os.path.isfile(os.path.join(log, "1_Train_EPOCH_pdmask.bmp")) and os.path.isfile(os.path.join(log, "%d_Val_EPOCH_gtmask.bmp" % epochs))
    assert len(m.history["train_loss"]) == epochs and np.isfinite(m.history["train_loss"]).all()
    m.model.load_state_dict(torch.load(os.path.join(log, pth)))       # the checkpoint is the best-validation epoch
    out = m.predict(np.load(tr_i[0]).reshape((1,) + shape))
    assert out.dtype == np.uint8 and out.shape == shape
    if numclass == 1:
        assert set(np.unique(out)) <= {0, 255}
    else:
        assert out.max() < numclass
    # predict() against the oracle's eval-mode forward with the same weights: identical mask (modelVNet.py:655-676:
    # sigmoid > 0.5 -> 255 / argmax of the soft-max), differences tolerated only at numerically tied voxels
    from oracle import seg_oracle as seg
    kind = "vnet" if "VNet" in cls else "unet"
    sd = {k: v.detach().cpu().float() for k, v in m.model.state_dict().items()}
    xin = torch.from_numpy(np.load(tr_i[0]).reshape((1, 1) + shape)).float()
    _, probs = seg.net_forward(kind, sd, xin)
    if numclass == 1:
        want = ((probs[0, 0] > 0.5).numpy() * 255).astype(np.uint8)
        tied = (probs[0, 0] - 0.5).abs().numpy() < 5e-5
    else:
        want = probs[0].argmax(0).numpy().astype(np.uint8)
        top2 = probs[0].topk(2, dim=0).values
        tied = (top2[0] - top2[1]).numpy() < 1e-4
    assert np.array_equal(out[~tied], want[~tied]) and tied.mean() < 0.01
    # the inference=True constructor path loads the checkpoint written above (inference.py:15-17)
    m2 = getattr(model, cls)(image_depth=16, image_height=16, image_width=16, image_channel=1, numclass=numclass, batch_size=1,
                             loss_name=loss, inference=True, model_path=os.path.join(log, pth), use_cuda=dev.type == "cuda")
    assert np.array_equal(m2.predict(np.load(tr_i[0]).reshape((1,) + shape)), out)
    with pytest.raises(ValueError):
        bad = getattr(model, cls)(16, 16, 16, 1, numclass, 1, loss_name="NoSuchLoss", use_cuda=dev.type == "cuda")
        bad.trainprocess(tr_i, tr_l, va_i, va_l, model_dir=log, epochs=1)

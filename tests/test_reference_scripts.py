"""The reference's entry scripts AS FILES (/root/reference/train.py, inference.py) executed against this repo's `model` / `dataprocess`
packages (the wrapper tests ran a re-typed script body; a drift between that paraphrase and the shipped file - e.g.
inference.py building `MutilUNet3dModel(numclass=1, inference=True, model_path=...)` - would not have been caught).

The file text is compiled and run unmodified with `__name__ == "__main__"`.  What the harness supplies is the outside world the scripts
hard-code: `pandas.read_csv` answers the three CSV paths from a temporary data set, `SimpleITK` is a small stand-in (not installable in this
image), `file_name_path` maps the author's F:\\ drive to a temporary directory, and the wrapper class is entered through a shim that binds the
script's constructor / method arguments against THIS repo's signatures (a keyword the wrappers do not take raises TypeError) and then
shrinks the volume (128 x 112 x 112 -> 16^3) and the epoch count so the run fits the GPU-less build box.
Every test takes the `dev` fixture (round 6): its "emu" leg runs the files against the host checker, its "gpu" leg (`-m gpu`) runs the
same files with `use_cuda=True` on cuda:0 against the product library.  Needs the reference tree (SEG_REFERENCE_ROOT, default /root/reference): both legs
are skipped where it does not exist - the driver's GPU box carries no reference files (they may not be copied into this repo), so the GPU leg runs
wherever a maintainer has the reference checked out next to a GPU."""
import inspect
import os
import sys
import types

import numpy as np
import pytest
import torch

import conftest

REF = os.environ.get("SEG_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="the reference tree is not present on this box")


def _shim(real_cls, record, small=(16, 16, 16), model_path=None, use_cuda=False):
    """enter `real_cls` the way the script does, check the call against the real signatures, run it small on the host checker (or cuda:0)"""
    class Shim(real_cls):
        def __init__(self, *args, **kw):
            bound = inspect.signature(real_cls.__init__).bind(self, *args, **kw)        # TypeError: the script passes something the wrapper lacks
            record["ctor"] = dict(bound.arguments)
            kw = dict(kw)
            kw.update(image_depth=small[0], image_height=small[1], image_width=small[2], use_cuda=use_cuda)
            if kw.get("inference"):
                kw["model_path"] = model_path
            real_cls.__init__(self, *args, **kw)

        def trainprocess(self, *args, **kw):
            bound = inspect.signature(real_cls.trainprocess).bind(self, *args, **kw)
            record["trainprocess"] = {k: v for k, v in bound.arguments.items() if k != "self"}
            kw = dict(kw)
            kw["epochs"] = 1
            return real_cls.trainprocess(self, *args, **kw)

        def inference(self, image, newSize=(96, 96, 96)):
            record.setdefault("inference", []).append(tuple(newSize))
            return real_cls.inference(self, image, (small[2], small[1], small[0]))
    Shim.__name__ = real_cls.__name__
    return Shim


def _run_file(name, monkeypatch):
    path = os.path.join(REF, name)
    with open(path) as f:
        text = f.read()
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", ""))      # the scripts overwrite both; restored afterwards
    monkeypatch.setenv("CUDA_LAUNCH_BLOCKING", os.environ.get("CUDA_LAUNCH_BLOCKING", ""))
    ns = {"__name__": "__main__", "__file__": path}
    exec(compile(text, path, "exec"), ns)
    return ns


def test_train_py_file_runs_against_this_package(dev, tmp_path, monkeypatch):
    cuda = dev.type == "cuda"
    import pandas as pd
    import model
    from pytorchdeeplearing_amd.model import seg_models
    monkeypatch.setenv("SEGENGINE_DTYPE", "f32")
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    rows = {}
    for tag, n in (("traindata", 1), ("trainaugdata", 1), ("validata", 1)):          # (one epoch over two 16^3 volumes: a minute less on the host checker)
        r = []
        for i in range(n):
            ip, mp = str(tmp_path / ("%s_img%d.npy" % (tag, i))), str(tmp_path / ("%s_msk%d.npy" % (tag, i)))
            np.save(ip, rng.standard_normal((16, 16, 16)).astype(np.float32))
            np.save(mp, rng.integers(0, 5, (16, 16, 16)).astype(np.int64))
            r.append((ip, mp))
        rows[tag] = pd.DataFrame(r, columns=["image", "mask"])
    asked = []

    def read_csv(p, *a, **k):
        asked.append(p)
        return rows[os.path.splitext(os.path.basename(str(p).replace("\\", "/")))[0]]
    monkeypatch.setattr(pd, "read_csv", read_csv)
    record = {}
    monkeypatch.setattr(model, "MutilUNet3dModel", _shim(seg_models.MutilUNet3dModel, record, use_cuda=cuda))
    ns = _run_file("train.py", monkeypatch)
    assert [os.path.basename(str(p).replace("\\", "/")) for p in asked] == ["traindata.csv", "trainaugdata.csv", "validata.csv"]
    # the arguments are the file's own (train.py:34-37) ...
    c = record["ctor"]
    assert (c["image_depth"], c["image_height"], c["image_width"], c["image_channel"], c["numclass"], c["batch_size"], c["loss_name"]) == \
        (128, 112, 112, 1, 5, 1, "MutilDiceLoss")
    t = record["trainprocess"]
    assert t["model_dir"] == "log/MutilUNet3d/dice" and t["epochs"] == 100 and list(t["showwind"]) == [16, 8]
    assert len(t["trainimage"]) == 2 and len(t["validationimage"]) == 1          # source + augmented rows concatenated and shuffled together
    # ... and the run left what the reference loop leaves: the checkpoint and the training curves under model_dir
    assert os.path.isfile(tmp_path / "log" / "MutilUNet3d" / "dice" / "MutilUNet3d.pth")
    assert callable(ns["trainmutilunet3d"])


class _FakeImage:
    def __init__(self, arr, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
        self.arr, self.spacing, self.origin, self.direction = np.asarray(arr), tuple(spacing), tuple(origin), tuple(direction)

    def GetSpacing(self): return self.spacing
    def GetOrigin(self): return self.origin
    def GetDirection(self): return self.direction
    def GetSize(self): return tuple(reversed(self.arr.shape))
    def SetSpacing(self, s): self.spacing = tuple(s)
    def SetOrigin(self, o): self.origin = tuple(o)
    def SetDirection(self, d): self.direction = tuple(d)


def test_inference_py_file_runs_against_this_package(dev, tmp_path, monkeypatch):
    cuda = dev.type == "cuda"
    import model
    import dataprocess.utils as DU
    from pytorchdeeplearing_amd import networks
    from pytorchdeeplearing_amd.model import _io, seg_models
    monkeypatch.setenv("SEGENGINE_DTYPE", "f32")
    monkeypatch.chdir(tmp_path)
    # the author's test directory with two volumes, behind a stand-in for SimpleITK (ReadImage / WriteImage / array conversion)
    data = tmp_path / "image"
    data.mkdir()
    rng = np.random.default_rng(1)
    for i in range(2):
        np.save(data / ("case%d.npy" % i), (rng.standard_normal((20, 18, 22)) * 200).astype(np.float32))
    written = []
    sitk = types.ModuleType("SimpleITK")
    sitk.Image = _FakeImage
    sitk.ReadImage = lambda p: _FakeImage(np.load(p), spacing=(0.8, 0.8, 1.5), origin=(1.0, 2.0, 3.0))
    sitk.WriteImage = lambda img, p: written.append((img, p))
    sitk.GetArrayFromImage = lambda img: img.arr
    sitk.GetImageFromArray = lambda a: _FakeImage(a)
    monkeypatch.setitem(sys.modules, "SimpleITK", sitk)
    monkeypatch.setattr(_io, "sitk", sitk)
    real_fnp = DU.file_name_path
    monkeypatch.setattr(DU, "file_name_path", lambda d, *a, **k: real_fnp(str(data), *a, **k))
    # the checkpoint the script names (a one-class UNet3d state_dict, as its constructor arguments imply)
    pth = str(tmp_path / "unet3d.pth")
    torch.save(networks.UNet3d(1, 1).state_dict(), pth)
    record = {}
    monkeypatch.setattr(model, "MutilUNet3dModel", _shim(seg_models.MutilUNet3dModel, record, model_path=pth, use_cuda=cuda))
    # the script concatenates "F:\..." + "/" + file name: reads and writes go to the temporary directory instead
    real_read = sitk.ReadImage
    sitk.ReadImage = lambda p: real_read(str(data / os.path.basename(str(p).replace("\\", "/"))))
    _run_file("inference.py", monkeypatch)
    c = record["ctor"]
    assert c["numclass"] == 1 and c["inference"] is True and c["loss_name"] == "MutilFocalLoss" and c["model_path"].endswith("BinaryVNet2dSegModel.pth")
    assert record["inference"] == [(112, 112, 128)] * 2                     # newSize as the file passes it (x, y, z)
    assert len(written) == 2
    for img, p in written:
        assert os.path.basename(str(p).replace("\\", "/")).startswith("case")
        assert img.arr.shape == (20, 18, 22) and img.arr.dtype == np.uint8          # mask on the SOURCE grid ...
        assert img.GetSpacing() == (0.8, 0.8, 1.5) and img.GetOrigin() == (1.0, 2.0, 3.0)      # ... carrying the source geometry (modelUnet.py:990-997)


def test_flask_app_py_file_serves_predictions_from_this_package(dev, tmp_path, monkeypatch):
    """north_star names flask_app.py beside train.py / inference.py: the file text is executed unmodified as a module
    (not as __main__, so the server is not started), then driven through `app.test_client()`: POST /predict with a volume file, GET /getresult
    for the mask.  The harness supplies only the outside world of /root/reference/flask_app.py:16-18,30-41: SimpleITK (stand-in, as above), the
    `D:/` upload directories (created relative to a temporary cwd), the checkpoint path, and `send_file`'s `attachment_filename` keyword that
    Flask >= 2.2 renamed to `download_name`.  Two requests are in flight at once to exercise the lock around the shared model object."""
    cuda = dev.type == "cuda"
    flask = pytest.importorskip("flask")
    import io
    import threading
    import model
    from pytorchdeeplearing_amd import networks
    from pytorchdeeplearing_amd.model import _io, seg_models
    monkeypatch.setenv("SEGENGINE_DTYPE", "f32")
    monkeypatch.chdir(tmp_path)
    sitk = types.ModuleType("SimpleITK")
    sitk.Image = _FakeImage
    sitk.ReadImage = lambda p: _FakeImage(np.load(p), spacing=(0.8, 0.8, 1.5), origin=(1.0, 2.0, 3.0))
    written = {}

    def write_image(img, p):
        written[os.path.basename(p)] = img
        with open(p, "wb") as f:                       # (np.save on a file object: no ".npy" is appended to the name the app chose)
            np.save(f, img.arr)
    sitk.WriteImage = write_image
    sitk.GetArrayFromImage = lambda img: img.arr
    sitk.GetImageFromArray = lambda a: _FakeImage(a)
    monkeypatch.setitem(sys.modules, "SimpleITK", sitk)
    monkeypatch.setattr(_io, "sitk", sitk)
    pth = str(tmp_path / "unet3d.pth")
    torch.save(networks.UNet3d(1, 1).state_dict(), pth)
    record = {}
    active, peak = [0], [0]
    Base = _shim(seg_models.MutilUNet3dModel, record, model_path=pth, use_cuda=cuda)

    class Counting(Base):                              # how many request threads are inside the network at once (must be 1: the lock)
        def _predict_device(self, *a, **k):
            import time
            active[0] += 1
            peak[0] = max(peak[0], active[0])
            try:
                time.sleep(0.05)                        # long enough for the other request thread to arrive
                return Base._predict_device(self, *a, **k)
            finally:
                active[0] -= 1
    Counting.__name__ = "MutilUNet3dModel"
    monkeypatch.setattr(model, "MutilUNet3dModel", Counting)
    real_send = flask.send_file

    def send_file(path, *a, attachment_filename=None, **k):
        if attachment_filename is not None:
            k["download_name"] = attachment_filename
        return real_send(os.path.abspath(path), *a, **k)
    monkeypatch.setattr(flask, "send_file", send_file)
    path = os.path.join(REF, "flask_app.py")
    with open(path) as f:
        text = f.read()
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", ""))
    monkeypatch.setenv("CUDA_LAUNCH_BLOCKING", os.environ.get("CUDA_LAUNCH_BLOCKING", ""))
    ns = {"__name__": "flask_app_under_test", "__file__": path}
    exec(compile(text, path, "exec"), ns)
    app = ns["app"]
    c = record["ctor"]                                 # flask_app.py:16-18
    assert (c["image_depth"], c["image_height"], c["image_width"], c["numclass"], c["inference"], c["loss_name"]) == (128, 112, 112, 1, True, "MutilFocalLoss")
    assert os.path.isdir("D:/uploads/Image") and os.path.isdir("D:/uploads/Mask")
    rng = np.random.default_rng(3)
    vols = {"case%d.npy" % i: (rng.standard_normal((20, 18, 22)) * 200).astype(np.float32) for i in range(2)}
    replies = {}

    def post(name):
        buf = io.BytesIO()
        np.save(buf, vols[name])
        buf.seek(0)
        replies[name] = app.test_client().post("/predict", data={"file": (buf, name)}, content_type="multipart/form-data")
    threads = [threading.Thread(target=post, args=(n,)) for n in vols]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for name in vols:
        assert replies[name].status_code == 200 and replies[name].get_data(as_text=True) == "Segmentation Success!"
    assert peak[0] == 1, "two requests were inside the shared model at once"
    assert record["inference"] == [(112, 112, 128)] * 2                      # newSize of flask_app.py:15
    cl = app.test_client()
    assert cl.post("/predict").get_data(as_text=True) == "No file uploaded"
    assert cl.get("/getresult").get_data(as_text=True) == "Missing parameter: file"
    assert cl.get("/getresult?file=absent.npy").get_data(as_text=True) == "The file does not exist"
    for name in vols:
        r = cl.get("/getresult?file=" + name)
        assert r.status_code == 200 and name in r.headers.get("Content-Disposition", "")
        mask = np.load(io.BytesIO(r.get_data()))
        assert mask.shape == (20, 18, 22) and mask.dtype == np.uint8 and np.array_equal(mask, written[name].arr)
        assert written[name].GetSpacing() == (0.8, 0.8, 1.5)
    # the served mask is what the wrapper's own inference gives for the same volume (one model object, deterministic eval forward)
    again = ns["Unet3d"].inference(_FakeImage(vols["case0.npy"], spacing=(0.8, 0.8, 1.5), origin=(1.0, 2.0, 3.0)), (112, 112, 128))
    assert np.array_equal(again.arr, written["case0.npy"].arr)

"""`model` package surface of the reference (model/__init__.py:1-3).  The eight segmentation wrappers
are backed by the HIP engine; the four ResNet classification wrappers are outside the hot path named
by BASELINE.json and are exported as explicit stubs so `from model import *`
in the reference scripts keeps working."""
from .seg_models import (BinaryUNet2dModel, BinaryUNet3dModel, BinaryVNet2dModel, BinaryVNet3dModel, MutilUNet2dModel,
                         MutilUNet3dModel, MutilVNet2dModel, MutilVNet3dModel)


def _out_of_scope(name):
    class _Stub(object):
        def __init__(self, *a, **k):
            raise NotImplementedError(name + " (classification) is outside the segmentation hot path this engine implements")
    _Stub.__name__ = name
    return _Stub


BinaryResNet2dModel = _out_of_scope("BinaryResNet2dModel")
BinaryResNet3dModel = _out_of_scope("BinaryResNet3dModel")
MutilResNet2dModel = _out_of_scope("MutilResNet2dModel")
MutilResNet3dModel = _out_of_scope("MutilResNet3dModel")

__all__ = ["BinaryVNet2dModel", "BinaryVNet3dModel", "MutilVNet2dModel", "MutilVNet3dModel", "BinaryUNet2dModel", "BinaryUNet3dModel",
           "MutilUNet2dModel", "MutilUNet3dModel", "BinaryResNet2dModel", "BinaryResNet3dModel", "MutilResNet2dModel", "MutilResNet3dModel"]

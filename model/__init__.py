"""Top-level `model` package name the reference scripts import (`from model import *` in train.py:4,
inference.py:3, flask_app.py:2): everything lives in pytorchdeeplearing_amd.model."""
from pytorchdeeplearing_amd.model import *  # noqa: F401,F403
from pytorchdeeplearing_amd.model import __all__  # noqa: F401

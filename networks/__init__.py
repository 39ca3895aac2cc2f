"""Top-level `networks` package name of the reference (networks/__init__.py:1-11)."""
from pytorchdeeplearing_amd.networks import UNet2d, UNet3d, VNet2d, VNet3d, initialize_weights  # noqa: F401

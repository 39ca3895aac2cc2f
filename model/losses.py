from pytorchdeeplearing_amd.losses import *  # noqa: F401,F403

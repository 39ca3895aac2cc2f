from pytorchdeeplearing_amd.metric import *  # noqa: F401,F403

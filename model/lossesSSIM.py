from pytorchdeeplearing_amd.lossesSSIM import *  # noqa: F401,F403
from pytorchdeeplearing_amd.lossesSSIM import ssim, ssim3D, SSIM, SSIM3D  # noqa: F401

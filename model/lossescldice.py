from pytorchdeeplearing_amd.lossescldice import *  # noqa: F401,F403
from pytorchdeeplearing_amd.lossescldice import (Binary_Soft_cldice_loss, Mutil_Soft_cldice_loss, norm_intersection,  # noqa: F401
                                                 soft_skeletonize)

"""MI355X-native (gfx950) segmentation engine behind the API surface of
junqiangchen/PytorchDeepLearing's VNet/UNet hot path.  See DESIGN.md."""
import os as _os

# Kernel arguments in device memory: with the host-memory path every one of the ~240 dependent launches of a train step pays a fabric round
# trip for its argument segment (measured on MI355X: 819 vs 886 volumes/s, profiles/r02_runtime_knobs_ab.log).  It is the runtime's default
# on this ROCm; pinned here (before the first HIP call) so an inherited HIP_FORCE_DEV_KERNARG=0 does not silently cost 8 %.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _capi  # noqa: E402,F401
from .engine import SegEngine  # noqa: E402,F401

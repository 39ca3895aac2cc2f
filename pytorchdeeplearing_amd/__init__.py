"""MI355X-native (gfx950) segmentation engine behind the API surface of
junqiangchen/PytorchDeepLearing's VNet/UNet hot path.  See DESIGN.md."""
from . import _capi  # noqa: F401
from .engine import SegEngine  # noqa: F401

"""deepchopper_amd -- MI355X-native `deepchopper predict` hot path (tokenise -> HyenaDNA
token-classification forward -> label smoothing / intervals) behind the reference's own API.

All hot-path arithmetic lives in libdeepchopper_hip.so (hand-written HIP for gfx950) behind the
C ABI of include/deepchopper_hip.h; there is no CPU fallback.
"""
__version__ = "0.1.0"

from .model import DeepChopper  # noqa: E402,F401

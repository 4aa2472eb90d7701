"""portrayer_amd — MI355X-native ray-cast/shade path of the portrayer ray tracer.

The compute path is hand-written HIP for gfx950 behind a C ABI (include/portrayer_hip.h,
portrayer_amd/libportrayer_hip.so); there is no CPU fallback."""
from . import _hip  # noqa: F401

__all__ = ["_hip"]

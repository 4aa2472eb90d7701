"""portrayer_amd — MI355X-native ray-cast/shade path of the portrayer ray tracer.

Layers: hand-written HIP kernels for gfx950 behind a C ABI (include/portrayer_hip.h,
libportrayer_hip.so, bound by `_hip`), and the C++ host layer mirroring the portrayer crate's API
above the pixel loop (portrayer_amd/host/portrayer.hpp, libportrayer_host.so, bound by `host`).
There is no CPU fallback: without the built libraries and an MI355X the calls raise."""
from . import _hip, host  # noqa: F401

__all__ = ["_hip", "host"]

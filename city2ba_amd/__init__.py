"""city2ba_amd -- MI355X-native hot path of city2ba behind the reference's BAProblem surface.

The compute lives in csrc/ (hand-written HIP for gfx950) behind the C ABI of include/city2ba_hip.h;
this package is the thin host-side mirror.  No CPU fallback exists."""
from ._lib import City2baError, device_count, lib  # noqa: F401
from .baproblem import BAProblem, reset_default_options, set_default_options, set_host_io_threads  # noqa: F401
from . import camera, generate, noise, synthetic  # noqa: F401   (device / dist import torch: import them explicitly)

__all__ = ["BAProblem", "City2baError", "device_count", "lib", "camera", "generate", "noise", "synthetic",
           "set_default_options", "reset_default_options", "set_host_io_threads"]

"""sonar-mi355x: MI355X-native drop-in for the procedural-noise / momentum-step / wavelet-split hot
path of blepping/ComfyUI-sonar.  ComfyUI loads this directory as a custom-node pack; tests and the
bench import it through ``sonar_pkg.load()``.

The arithmetic lives in ``csrc/*.hip`` (libsonar_hip.so, C ABI in include/sonar_hip.h) and is reached
through ``hip_lib`` (ctypes).  There is no non-HIP implementation: importing works anywhere (so the
node registry can be inspected), running anything requires the library and a ROCm device.
"""
from . import hip_lib  # noqa: F401

__all__ = ["hip_lib"]

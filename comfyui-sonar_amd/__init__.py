"""sonar-mi355x: MI355X-native drop-in for the procedural-noise / momentum-step / wavelet-split hot
path of blepping/ComfyUI-sonar.  ComfyUI loads this directory as a custom-node pack (it exports
``NODE_CLASS_MAPPINGS`` and registers the three Sonar samplers); tests and the bench import it through
``sonar_pkg.load()``.

The arithmetic lives in ``csrc/*.hip`` (libsonar_hip.so, C ABI in include/sonar_hip.h) and is reached
through ``hip_lib`` (ctypes).  There is no non-HIP implementation: importing works anywhere (so the
node registry can be inspected), running anything requires the library and a ROCm device.
"""
import sys

from . import hip_lib  # noqa: F401
from .py import nodes, sonar

NODE_CLASS_MAPPINGS = nodes.NODE_CLASS_MAPPINGS
NODE_DISPLAY_NAME_MAPPINGS = nodes.NODE_DISPLAY_NAME_MAPPINGS


def blep_init():
    """Publish the pack for sibling packs (reference __init__.py:7-12)."""
    registry = sys.modules.get("_blepping_integrations", {})
    if "sonar" not in registry:
        registry["sonar"] = sys.modules[__name__]
        sys.modules["_blepping_integrations"] = registry


sonar.add_samplers()  # no-op outside ComfyUI
blep_init()

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS", "hip_lib"]

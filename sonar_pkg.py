"""Imports the hyphen-named package directory ``comfyui-sonar_amd/`` under the importable alias
``comfyui_sonar_amd`` (ComfyUI loads custom-node packs by path the same way)."""
from __future__ import annotations

import importlib.util
import os
import sys

ALIAS = "comfyui_sonar_amd"
PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "comfyui-sonar_amd")


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod

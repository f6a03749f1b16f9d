"""Registers `face-recognition-platform_amd/` (not a valid identifier) as package `frp_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "face-recognition-platform_amd")


def load():
    if "frp_amd" in sys.modules:
        return sys.modules["frp_amd"]
    spec = importlib.util.spec_from_file_location(
        "frp_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["frp_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


load()

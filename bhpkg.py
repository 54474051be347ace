"""Import helper: the package directory is named `nbody-barnes-hut-cuda_amd` (not a valid
Python identifier), so it is loaded under the module name `nbody_barnes_hut_cuda_amd`."""
import importlib.util
import os
import sys

NAME = "nbody_barnes_hut_cuda_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "nbody-barnes-hut-cuda_amd")


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(
        NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules[NAME]
        raise
    return mod

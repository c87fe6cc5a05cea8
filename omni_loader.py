"""Registers the hyphenated package directory ``omni-swarm_amd/`` as the importable module ``omni_swarm_amd``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "omni_swarm_amd" in sys.modules:
        return sys.modules["omni_swarm_amd"]
    pkg_dir = os.path.join(_ROOT, "omni-swarm_amd")
    spec = importlib.util.spec_from_file_location(
        "omni_swarm_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["omni_swarm_amd"] = mod
    spec.loader.exec_module(mod)
    return mod

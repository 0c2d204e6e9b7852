"""Loads the package directory `s-rack_amd/` (not a valid Python identifier) as module `srack_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "s-rack_amd")


def load():
    if "srack_amd" in sys.modules:
        return sys.modules["srack_amd"]
    spec = importlib.util.spec_from_file_location(
        "srack_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["srack_amd"] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules["srack_amd"]
        raise
    return mod


def load_workloads():
    """workloads.py has no native dependency; usable before the HIP library is built."""
    if "srack_amd_workloads" in sys.modules:
        return sys.modules["srack_amd_workloads"]
    spec = importlib.util.spec_from_file_location("srack_amd_workloads", os.path.join(PKG_DIR, "workloads.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["srack_amd_workloads"] = mod
    spec.loader.exec_module(mod)
    return mod

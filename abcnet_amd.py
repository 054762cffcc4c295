"""Import shim: the package directory is named ``abc-net_amd`` (not a valid Python
identifier), so ``import abcnet_amd`` loads it from that directory under this
name and replaces this module with the real package in ``sys.modules``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "abc-net_amd")
_spec = importlib.util.spec_from_file_location(
    "abcnet_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["abcnet_amd"] = _mod
_spec.loader.exec_module(_mod)

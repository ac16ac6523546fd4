"""Import alias: the package lives in `gs-livm_amd/` (not a valid Python identifier).

`import gs_livm_amd` executes this file, which loads `gs-livm_amd/__init__.py` as the package
`gs_livm_amd` and replaces itself in sys.modules.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gs-livm_amd")
_spec = importlib.util.spec_from_file_location("gs_livm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gs_livm_amd"] = _mod
_spec.loader.exec_module(_mod)

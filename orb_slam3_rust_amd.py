"""Import shim: the package directory is named `orb-slam3-rust_amd/` (the name the build contract
fixes), which is not a valid Python identifier.  `import orb_slam3_rust_amd` loads that directory
as a regular package under this module's name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb-slam3-rust_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

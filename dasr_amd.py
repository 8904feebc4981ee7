"""Import alias for the package directory ``depth-aware-endoscopy-sr_amd/``.

The package directory carries the project's name, which contains hyphens and so
cannot be written in an ``import`` statement.  ``import dasr_amd`` executes this
file, which loads that directory as the package ``dasr_amd`` (sub-modules such
as ``dasr_amd.depthnet`` resolve inside it).
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "depth-aware-endoscopy-sr_amd")
_spec = importlib.util.spec_from_file_location(
    "dasr_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dasr_amd"] = _mod
_spec.loader.exec_module(_mod)

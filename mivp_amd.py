"""Import alias: the package directory is named after the reference repository
(``medical-image-segmentation-with-visual-prompts_amd``), which is not a valid
Python identifier.  ``import mivp_amd`` loads that directory as the package
``mivp_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "medical-image-segmentation-with-visual-prompts_amd")
_spec = importlib.util.spec_from_file_location(
    "mivp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mivp_amd"] = _mod
_spec.loader.exec_module(_mod)

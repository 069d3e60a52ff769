"""Import alias: the package directory is named `retrieval-augmented-diffusion-models_amd` (not a valid
Python identifier), so this module loads it under the name `rdm_amd`.

    import rdm_amd
    from rdm_amd.models.diffusion.ddim import DDIMSampler
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "retrieval-augmented-diffusion-models_amd")
_spec = importlib.util.spec_from_file_location("rdm_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rdm_amd"] = _mod
_spec.loader.exec_module(_mod)

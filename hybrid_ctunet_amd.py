"""Import shim: registers the directory ``hybrid-ctunet_amd/`` (not a valid Python identifier) as the package
``hybrid_ctunet_amd``.  After ``import hybrid_ctunet_amd`` the usual submodule imports work, e.g.
``from hybrid_ctunet_amd.networks.hybrid_CTUNet import CTUNet``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hybrid-ctunet_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

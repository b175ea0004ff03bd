"""Import shim: the package directory is `realsr-ncnn-vulkan_amd/` (reference name + `_amd`), which is
not a valid Python identifier.  `import realsr_ncnn_vulkan_amd` loads that directory as a package
(submodules such as `realsr_ncnn_vulkan_amd.synth` work)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "realsr-ncnn-vulkan_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

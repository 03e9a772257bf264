"""Import shim: the package directory is `cuda-qr_amd/` (hyphen, mirrors the reference repo name),
which Python cannot import by name.  `import cuda_qr_amd` loads it from its path."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cuda-qr_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

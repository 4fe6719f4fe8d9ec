"""Importable short alias for the package directory
``spatiotemporal-deepfake-detection-for-live-video-calls_amd/`` (whose name is not a valid Python
identifier).  ``import af_mi355x`` / ``from af_mi355x import engine`` / ``import af_mi355x.engine`` all
resolve to the very same module objects as the long name - no second copy of any submodule exists."""
import importlib
import importlib.abc
import importlib.machinery
import sys

_LONG = "spatiotemporal-deepfake-detection-for-live-video-calls_amd"
_ALIAS = __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_LONG + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[_ALIAS] = importlib.import_module(_LONG)

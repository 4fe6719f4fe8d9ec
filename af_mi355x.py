"""Importable short alias for the package directory
``spatiotemporal-deepfake-detection-for-live-video-calls_amd/`` (whose name is not a valid
Python identifier).  ``import af_mi355x`` / ``from af_mi355x import synth`` resolve to the very
same module objects — no second copy of any submodule is created."""
import importlib
import sys

_LONG = "spatiotemporal-deepfake-detection-for-live-video-calls_amd"
_pkg = importlib.import_module(_LONG)
for _name, _mod in list(sys.modules.items()):
    if _name == _LONG or _name.startswith(_LONG + "."):
        sys.modules[__name__ + _name[len(_LONG):]] = _mod

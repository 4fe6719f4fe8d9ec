"""Reference-side plugin shim.  Copy this ONE file to
``<reference>/altfreezing/model/classifier/i3d_mi355x.py`` and set ``classifier_type: i3d_mi355x`` in
``altfreezing/setting/i3d_ori.yaml`` (line 60).  ``PluginLoader.get_classifier("i3d_mi355x")`` then returns
this ``Classifier`` (utils/plugin_loader.py:27-52 imports ``model.classifier.<name>`` and takes its attribute
``Classifier``); demo.py / TEST2.py / feature.py / test/af_realtime.py need no change.

Requires this repository on ``sys.path`` (or ``AF_MI355X_ROOT`` pointing at it) with ``libafhip.so`` built.
"""
import os
import sys

_root = os.environ.get("AF_MI355X_ROOT")
if _root and _root not in sys.path:
    sys.path.insert(0, _root)

import af_mi355x  # noqa: E402,F401
from af_mi355x.classifier import Classifier as _Mi355xClassifier  # noqa: E402

try:                                   # inside the reference tree: honour its yaml (clip_size: 32, imsize: 224)
    from config import config as _cfg
    _CLIP, _IMSIZE = int(_cfg.clip_size), int(_cfg.imsize)
except Exception:                      # stand-alone use
    _CLIP, _IMSIZE = 32, 224


class Classifier(_Mi355xClassifier):
    def __init__(self):
        super().__init__(clip_size=_CLIP, imsize=_IMSIZE, precision=os.environ.get("AF_MI355X_PRECISION", "auto"))

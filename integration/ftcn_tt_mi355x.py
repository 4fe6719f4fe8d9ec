"""Reference-side plugin shim for the FTCN-TT plugin.  Copy this ONE file to
``<reference>/altfreezing/model/classifier/ftcn_tt_mi355x.py`` and set ``classifier_type: ftcn_tt_mi355x`` in
``altfreezing/setting/ftcn_tt.yaml`` (line 60, in place of ``i3d_temporal_var_fix_dropout_tt_cfg``).
``PluginLoader.get_classifier("ftcn_tt_mi355x")`` then returns this ``Classifier`` (utils/plugin_loader.py:27-52);
checkpoints saved by the reference plugin load unchanged (same state_dict keys, incl. the ``<bn>.0`` ones).

Requires this repository on ``sys.path`` (or ``AF_MI355X_ROOT`` pointing at it) with ``libafhip.so`` built.
"""
import os
import sys

_root = os.environ.get("AF_MI355X_ROOT")
if _root and _root not in sys.path:
    sys.path.insert(0, _root)

import af_mi355x  # noqa: E402,F401
from af_mi355x.classifier import FtcnTTClassifier as _Mi355xFtcnTT  # noqa: E402

try:                                   # inside the reference tree: honour its yaml (clip_size: 32, imsize: 224)
    from config import config as _cfg
    _CLIP, _IMSIZE = int(_cfg.clip_size), int(_cfg.imsize)
except Exception:                      # stand-alone use
    _CLIP, _IMSIZE = 32, 224


class Classifier(_Mi355xFtcnTT):
    def __init__(self):
        super().__init__(clip_size=_CLIP, imsize=_IMSIZE, precision=os.environ.get("AF_MI355X_PRECISION", "auto"))

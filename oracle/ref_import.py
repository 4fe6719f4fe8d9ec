"""TEST INFRASTRUCTURE — build-container only.

Imports the *unmodified* reference AltFreezing classifier from /root/reference
so that golden vectors can be generated (oracle/gen_golden.py) and the CPU
restatement (oracle/i3d_oracle.py) can be pinned against it.

Nothing in here may be imported by the product package, and nothing in here
runs on the GPU box (/root/reference does not exist there).  The three absent
pure-Python third-party packages the reference imports at module-import time
(fvcore, simplejson, termcolor) are replaced by minimal stand-ins in
``sys.modules``; none of them touches the arithmetic of the forward path
(SURVEY.md section 8c / Appendix A).
"""
import copy
import json
import os
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = "/root/reference"
ALTFREEZING_DIR = os.path.join(REFERENCE_ROOT, "altfreezing")


def reference_available() -> bool:
    return os.path.isdir(ALTFREEZING_DIR)


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


class _Registry(dict):
    """stand-in for fvcore.common.registry.Registry (slowfast/models/build.py:7-9)."""

    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


class _CfgNode(dict):
    """stand-in for fvcore.common.config.CfgNode (slowfast/config/defaults.py:6,23-27,818)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = type(self)(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = v


def _install_shims():
    if "fvcore" in sys.modules and getattr(sys.modules["fvcore"], "_af_shim", False):
        return
    for n in ("fvcore", "fvcore.common", "fvcore.nn"):
        _mod(n)._af_shim = True
    _mod("fvcore.common.registry").Registry = _Registry
    _mod("fvcore.common.config").CfgNode = _CfgNode
    _mod("fvcore.common.file_io").PathManager = type("PathManager", (), {"open": staticmethod(open)})

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    _mod("fvcore.nn.weight_init").c2_msra_fill = c2_msra_fill
    sj = _mod("simplejson")
    sj.dumps, sj.loads = json.dumps, json.loads
    _mod("termcolor").colored = lambda s, *a, **k: s


_CFG_DONE = False


def import_reference(setting="i3d_ori.yaml"):
    """Returns the reference ``config`` singleton and ``PluginLoader`` class."""
    global _CFG_DONE
    if not reference_available():
        raise RuntimeError("reference tree not present (this only works in the build container)")
    _install_shims()
    if ALTFREEZING_DIR not in sys.path:
        sys.path.insert(0, ALTFREEZING_DIR)
    from config import config as cfg  # reference altfreezing/config.py
    if not _CFG_DONE:
        cfg.init_with_yaml()
        cfg.update_with_yaml(setting)
        cfg.freeze()
        _CFG_DONE = True
    from utils.plugin_loader import PluginLoader  # reference utils/plugin_loader.py
    return cfg, PluginLoader


def build_reference_classifier():
    """``PluginLoader.get_classifier('i3d_ori')().eval()`` — ModelBase wrapping I3D8x8."""
    cfg, PluginLoader = import_reference()
    clf = PluginLoader.get_classifier(cfg.classifier_type)()
    return clf.eval()


def reference_modules():
    """The reference's layer modules (for per-layer known-answer fixtures)."""
    import_reference()
    from slowfast.models import head_helper, resnet_helper, stem_helper, video_model_builder
    return types.SimpleNamespace(
        stem_helper=stem_helper,
        resnet_helper=resnet_helper,
        head_helper=head_helper,
        video_model_builder=video_model_builder,
    )

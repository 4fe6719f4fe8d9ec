"""CPU, build container only: the reference-side shim integration/i3d_mi355x.py exercised through the reference's OWN
plugin loader (altfreezing/utils/plugin_loader.py:27-52) and ModelBase lifecycle (model/_base.py:39-104):
PluginLoader.get_classifier("i3d_mi355x")() -> .eval() -> .load(wrapped checkpoint) -> state_dict keys == the reference
layout.  Skipped where /root/reference does not exist (the GPU box).  Runs in a child process: importing the reference
puts its top-level packages (config, utils, model, ...) into sys.modules."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_json

SCRIPT = r'''
import json, os, sys, tempfile
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ref_import
cfg, PluginLoader = ref_import.import_reference()          # the reference's config singleton + loader, i3d_ori.yaml
import model.classifier as plugins                           # the reference's plugin package
plugins.__path__.append(os.path.join(ROOT, "integration"))  # = "copy i3d_mi355x.py next to i3d_ori.py"
os.environ["AF_MI355X_ROOT"] = ROOT
Cls = PluginLoader.get_classifier("i3d_mi355x")             # what `classifier_type: i3d_mi355x` resolves to
from model._base import ModelBase
ref_cls = PluginLoader.get_classifier(cfg.classifier_type)
clf = Cls()
clf.eval()
from af_mi355x import synth
sd = synth.synthetic_state_dict(seed=0)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "w.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, p)
    ok = clf.load(p)
    missing = clf.load(os.path.join(td, "nope.pth"))
got = clf.network.state_dict()
try:
    clf(torch.zeros(1, 3, 32, 224, 224))
    cpu_call = "ran"
except RuntimeError as e:
    cpu_call = "RuntimeError: " + str(e)[:60]
print(json.dumps({
    "class": Cls.__name__, "module": Cls.__module__, "clip": clf.network.clip_size,
    "load": list(ok), "load_missing": list(missing),
    "layout": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in got.items()],
    "loaded_equal": all(torch.equal(got[k], sd[k]) for k in sd),
    "has_surface": all(hasattr(clf, a) for a in ("network", "_warped_network", "load", "forward", "module_to_build")),
    "ref_surface": sorted(a for a in ("load", "forward", "module_to_build", "freeze") if hasattr(ref_cls, a)),
    "cpu_call": cpu_call,
}))
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/altfreezing"), reason="reference tree not present (GPU box)")
def test_shim_through_the_reference_plugin_loader(tmp_path):
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["class"] == "Classifier" and out["module"] == "model.classifier.i3d_mi355x"
    assert out["clip"] == 32                                        # read from the reference's yaml via its config singleton
    assert out["load"] == [True, -1] and out["load_missing"] == [False, -1]
    lay = load_json("layout.json")
    assert out["layout"] == lay["entries"] and len(out["layout"]) == 320
    assert out["loaded_equal"] and out["has_surface"]
    assert out["ref_surface"] == ["forward", "freeze", "load", "module_to_build"]
    assert out["cpu_call"].startswith("RuntimeError")              # no CPU fallback behind the plugin surface either

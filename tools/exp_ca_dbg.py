"""Run ON THE GPU BOX with AF_HIP_LIB=<package>/libafhip_stamps.so (tools/stamps_lib.sh af_conv_ca): the two conv_ca launches of s2
(B = 16 bf16) with parts of their memory traffic switched off (AF_CA_DBG bits; timing only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from af_mi355x import synth
from af_mi355x.classifier import Classifier
dev = torch.device("cuda", 0)
clf = Classifier(precision="bf16"); clf.network.load_state_dict(synth.synthetic_state_dict(seed=0)); clf = clf.to(dev).eval()
x = synth.normalize_like_callers(synth.synthetic_clips_u8(16, seed=2026, kind="uniform").to(dev))
with torch.inference_mode():
    clf(x)
    eng = clf.network._engines[("bf16", 16, (32, 224, 224))]
    for dbg in (0, 1, 2, 4, 8, 16, 3, 12, 15, 0):
        os.environ["AF_CA_DBG"] = str(dbg)
        best = {}
        for rep in range(5):
            ms = eng.run_timed()
            for n, m in zip(eng.op_names, ms):
                if "->" in n: best[n] = min(best.get(n, 1e9), m)
        print("dbg=%2d " % dbg + "  ".join("%s %.4f" % (n.replace("resnet.", "").replace("pathway0_", "").replace("branch2.", ""), m) for n, m in best.items()), flush=True)

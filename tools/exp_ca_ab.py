"""Run ON THE GPU BOX: the two conv_ca launches of s2 (B = 16 bf16) with the stage's c weights as an LDS image (AF_CA_CWL=1, default)
and as per-wave fragment loads (AF_CA_CWL=0), interleaved on one box."""
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
    y0 = None
    for cwl in ("0", "1", "0", "1"):
        os.environ["AF_CA_CWL"] = cwl
        y = clf(x)["final_output"].float().cpu()
        if y0 is None: y0 = y
        eng = clf.network._engines[("bf16", 16, (32, 224, 224))]
        best = {}
        for rep in range(8):
            ms = eng.run_timed()
            for n, m in zip(eng.op_names, ms):
                if "->" in n: best[n] = min(best.get(n, 1e9), m)
        print("cwl=%s " % cwl + "  ".join("%s %.4f" % (n.replace("resnet.", "").replace("pathway0_", "").replace("branch2.", ""), m) for n, m in best.items())
              + "  | logits equal to the first run: %s" % bool(torch.equal(y, y0)), flush=True)

"""Run ON THE GPU BOX: per-op device times of the B = 1 forward (bf16), sorted - where one clip's 0.83 ms goes."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from af_mi355x import synth, _lib
from af_mi355x.classifier import Classifier
dev = torch.device("cuda", 0)
clf = Classifier(precision="bf16"); clf.network.load_state_dict(synth.synthetic_state_dict(seed=0)); clf = clf.to(dev).eval()
x = synth.normalize_like_callers(synth.synthetic_clips_u8(1, seed=2026, kind="uniform").to(dev))
with torch.inference_mode():
    clf(x)
    eng = clf.network._engines[("bf16", 1, (32, 224, 224))]
    best = None
    for rep in range(10):
        ms = eng.run_timed()
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    print("sum of per-op minima %.4f ms over %d ops" % (sum(best), len(best)))
    for n, m in sorted(zip(eng.op_names, best), key=lambda t: -t[1])[:22]:
        print("%-60s %.4f" % (n[-60:], m))

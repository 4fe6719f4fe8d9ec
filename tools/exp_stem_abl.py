"""Run ON THE GPU BOX: the fused stem's time in the model (B = 16 bf16, min of 8) - with AF_HIP_LIB=<package>/libafhip_stemabl.so
(-DAF_STEM_NO_ROW1_LOADS: the second conv row of a pair re-uses the first row's fragments: timing only) it shows what half of the
stem's global loads cost."""
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
    best = None
    for rep in range(8):
        ms = eng.run_timed()
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    print(os.path.basename(os.environ.get("AF_HIP_LIB", "libafhip.so")), [(n, round(m, 4)) for n, m in zip(eng.op_names, best)][:3])

"""Run ON THE GPU BOX: device time of the B=16 forward from fp32 clips and from uint8 clips (fused prologue), per-op table of the u8 engine."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from af_mi355x import synth
from af_mi355x.classifier import Classifier
dev = torch.device("cuda", 0)
clf = Classifier(precision="bf16"); clf.network.load_state_dict(synth.synthetic_state_dict(seed=0)); clf = clf.to(dev).eval()
u8 = synth.synthetic_clips_u8(16, seed=2026, kind="uniform").to(dev)
x = synth.normalize_like_callers(u8)
def timed(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.inference_mode():
    print("fp32 input  : %.3f ms" % timed(lambda: clf(x)))
    print("uint8 input : %.3f ms" % timed(lambda: clf.network.forward_clips_u8(u8)))
    print("uint8 + pooled: %.3f ms" % timed(lambda: clf.network.forward_clips_u8(u8, return_pooled=True)))
    eng = clf.network._engines[("bf16", 16, (32, 224, 224))]
    ms = eng.run_timed()
    print("per-op (last bound input): first ops", [(n, round(m, 4)) for n, m in list(zip(eng.op_names, ms))[:3]], "sum %.3f" % sum(ms))

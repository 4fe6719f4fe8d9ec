"""Run ON THE GPU BOX: would running stem + s2 in clip chunks small enough for the 256 MiB Infinity Cache beat the whole-batch launches?
Times ops[0:k) (k = one past s2's last op) at B = 16 in one go against 16/b passes of a B = b engine over the b-clip slices."""
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
    for b in (16, 16):
        xs = [x[i:i + b] for i in range(0, 16, b)]
        clf(xs[0])
        eng = clf.network._engines[("bf16", b, (32, 224, 224))]
        k = max(i for i, n in enumerate(eng.op_names) if ".s2." in n) + 1
        def run():
            for xc in xs:
                eng._bind_f32(0, xc)
                eng.run_prefix(k)
        t = timed(run)
        def run_all():
            for xc in xs:
                eng._bind_f32(0, xc)
                eng.run_prefix(eng.n_ops)
        ta = timed(run_all)
        ms = eng.run_timed()
        print("B=%2d x %2d: ops[0:%d) %.3f ms per 16 clips; whole net %.3f; per-op (one chunk): %s" % (b, len(xs), k, t, ta, [(n.replace("resnet.", "").replace("pathway0_", ""), round(m, 4)) for n, m in list(zip(eng.op_names, ms))[:k]]), flush=True)

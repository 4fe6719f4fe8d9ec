"""Run ON THE GPU BOX: where a live-call window's enqueue -> score latency goes (aligner call, forward call, device time, sync),
paced like bench.py --model stream (GPU idle between windows) and back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from af_mi355x import aligner, synth
from af_mi355x.classifier import Classifier

dev = torch.device("cuda:0")
clf = Classifier(precision="bf16")
clf.network.load_state_dict(synth.synthetic_state_dict(seed=0))
clf = clf.to(dev).eval()
al = aligner.FasterCropAlignXRay(224, device=dev)
windows = [aligner.synthetic_clip(32, seed=2026 + i) for i in range(4)]

def one(k, paced):
    infos, crops = windows[k % 4]
    if paced:
        time.sleep(0.25)
    t0 = time.perf_counter()
    _, clip = al(infos, crops, device_output=True)
    t1 = time.perf_counter()
    s = clf.network.infer_scores(clip.unsqueeze(0), as_numpy=False)
    t2 = time.perf_counter()
    v = s.float().cpu().numpy()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3

with torch.inference_mode():
    for k in range(6): one(k, False)
    for paced in (True, False):
        r = np.array([one(k, paced) for k in range(16)])
        print("paced" if paced else "back-to-back", "median ms: aligner call %.3f | forward call %.3f | wait for score %.3f | total %.3f" % tuple(np.median(r, 0)))
    # device time of the pieces
    infos, crops = windows[0]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    _, clip = al(infos, crops, device_output=True); torch.cuda.synchronize()
    e[0].record(); _, clip = al(infos, crops, device_output=True); e[1].record()
    s = clf.network.infer_scores(clip.unsqueeze(0), as_numpy=False); e[2].record(); torch.cuda.synchronize()
    print("device ms: upload + warp %.3f | prologue + forward %.3f" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))

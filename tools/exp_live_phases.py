"""Run ON THE GPU BOX: where the live form's enqueue -> score time goes (host phases of close_window, idle vs busy GPU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from af_mi355x import aligner, synth
from af_mi355x.classifier import Classifier, LiveScorer

dev = torch.device("cuda", 0)
clf = Classifier(precision="bf16"); clf.network.load_state_dict(synth.synthetic_state_dict(seed=0)); clf = clf.to(dev).eval()
windows = [aligner.synthetic_clip(32, seed=2026 + i) for i in range(4)]
sal = aligner.StreamingCropAligner(224, capacity=64, device=dev)
scorer = LiveScorer(clf.network)

def capture(k, frames):
    infos, crops = windows[k % 4]
    for i in frames:
        sal.push(infos[i], crops[i])

def close(k, marks):
    t = time.perf_counter(); capture(k, (31,)); marks[0] += time.perf_counter() - t
    t = time.perf_counter(); sal.align_last(32, out=scorer.clip[0]); marks[1] += time.perf_counter() - t
    t = time.perf_counter(); s = scorer(); marks[2] += time.perf_counter() - t
    return s

with torch.inference_mode():
    for k in range(3):
        capture(k, range(31)); close(k, [0, 0, 0])
    for idle in (0.0, 0.03, 0.3, 1.0):
        marks, tot, n = [0.0, 0.0, 0.0], [], 12
        for k in range(n):
            capture(k, range(31)); torch.cuda.synchronize()
            time.sleep(idle)
            t0 = time.perf_counter(); close(k, marks); tot.append(time.perf_counter() - t0)
        print("idle %.2f s before the last frame: push %.3f ms | align_last (fit + warp launch) %.3f | scorer (graph + score to host) %.3f | total p50 %.3f ms"
              % (idle, 1e3 * marks[0] / n, 1e3 * marks[1] / n, 1e3 * marks[2] / n, 1e3 * float(np.median(tot))))
    # device time of the graph replay alone, GPU busy
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    scorer(); torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20): scorer.graph.replay()
    ev[1].record(); torch.cuda.synchronize()
    print("graph replay, back to back: %.3f ms each" % (ev[0].elapsed_time(ev[1]) / 20))

"""Run ON THE GPU BOX: where the host-inclusive aligner time goes (fit, staging copy by thread count, H2D, launch) and the
pipelined rate of FasterCropAlignXRay.__call__(device_output=True) back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from concurrent.futures import ThreadPoolExecutor
from af_mi355x import aligner

dev = torch.device("cuda:0")
clips = [aligner.synthetic_clip(32, seed=2026 + i) for i in range(4)]
al = aligner.FasterCropAlignXRay(224, device=dev)
infos, crops = clips[0]
print("cpus", os.cpu_count(), "copy threads", aligner._COPY_THREADS)
t = time.perf_counter()
for _ in range(200): al(infos)
print("landmark-only call: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
offs, total = al._layout(crops)
host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
hv = host.numpy()
pairs = [(hv[o:o + im.size], im) for im, o in zip(crops, offs)]
for nt in (1, 2, 4, 8, 16):
    pool = ThreadPoolExecutor(nt)
    for _ in range(3): list(pool.map(aligner._copy_group, [pairs[i::nt] for i in range(nt)]))
    t = time.perf_counter()
    for _ in range(30): list(pool.map(aligner._copy_group, [pairs[i::nt] for i in range(nt)]))
    dt = (time.perf_counter() - t) / 30
    print("stage copy %2d threads: %.3f ms (%.1f GB/s)" % (nt, dt * 1e3, total / dt / 1e9))
d = torch.empty(total, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): d.copy_(host, non_blocking=True)
e1.record(); torch.cuda.synchronize()
print("H2D %.1f MB: %.3f ms (%.1f GB/s)" % (total / 1e6, e0.elapsed_time(e1) / 20, total / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9))
for _ in range(5):
    for inf, cr in clips: al(inf, cr, device_output=True)
torch.cuda.synchronize()
t = time.perf_counter()
n = 0
for _ in range(10):
    for inf, cr in clips:
        out = al(inf, cr, device_output=True); n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("pipelined __call__(device_output=True): %.3f ms per clip = %.0f clips/s" % (dt / n * 1e3, n / dt))

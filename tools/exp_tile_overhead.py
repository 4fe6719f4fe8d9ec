"""Run ON THE GPU BOX: time of one full-chip round of generic-kernel tiles as a function of K (fixed cost per tile = intercept)."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from exp_conv111 import layer, timeit   # noqa
import numpy as np
for n_out, label in ((256, "N=256"), (128, "N=128"), (512, "N=512 (2 rounds of 256x256)")):
    xs, ys = [], []
    for cin in (576, 1152, 2304, 4608):
        run, name = layer(1, 64, 32, 32, cin, n_out, res=False)      # M = 65536 = 256 tiles of 256 rows
        t = min(timeit(run, 100), timeit(run, 100))
        xs.append(cin / 64); ys.append(t)
        print(label, "cin", cin, name, "%.1f us" % t, flush=True)
    b, a = np.polyfit(xs, ys, 1)
    print("   fit: %.2f us fixed + %.3f us per K-step (%.0f TFLOP/s marginal)" % (a, b, 2 * 65536 * n_out * 64 / b / 1e6))

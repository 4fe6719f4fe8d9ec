"""Run ON THE GPU BOX: s4's c conv (256 -> 1024 + residual, 14 x 14, B = 16) on conv111 with 32-channel wave columns / 4-slot ring
(shipped) against 64-channel columns / 2-slot ring (AF_C111_WC64=1: 128-byte row segments, one tile ahead)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_variants
from exp_variants import mk, layer, L
from exp_conv111 import timeit
for dt in ("bf16",):
    exp_variants.DT = dt
    d = mk(16, 16, 14, 14, 256, 1024)
    run = layer(d, None, True)
    for rep in range(3):
        for wc in ("0", "1"):
            os.environ["AF_C111_WC64"] = wc
            print(dt, "wc64=" + wc, "%.1f us" % timeit(run, 300), flush=True)

"""Run ON THE GPU BOX: launch time of the s2 `b` conv (64 -> 64, 1x3x3, 56 x 56, B = 16) in its forms (AF_C64_WAVES = 4 / 8 / default)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_variants
from exp_variants import mk, layer, L
from exp_conv111 import timeit
for dt in ("bf16", "f16"):
    exp_variants.DT = dt
    d = mk(16, 32, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    run = layer(d)
    for rep in range(2):
        for form in ("4", "8"):
            os.environ["AF_C64_WAVES"] = form
            print(dt, "form", form, "%.1f us" % timeit(run, 300), flush=True)

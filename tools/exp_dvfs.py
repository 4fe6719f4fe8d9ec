"""Run ON THE GPU BOX: is the sustained MFMA rate of the 256x256 tile clock / power bound?  The same tile (K = 4608) on 64, 128 and
256 CUs (one tile per CU), and the full chip back to back for ~40 ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from exp_conv111 import layer, timeit   # noqa
for t, label in ((16, "64 tiles"), (32, "128 tiles"), (64, "256 tiles")):
    run, name = layer(1, t, 32, 32, 4608, 256, res=False)
    for reps in (5, 50, 300):
        us = timeit(run, reps)
        print("%-10s %s K=4608: %7.1f us per launch over %3d launches  (%.0f TFLOP/s per active CU x 256)" %
              (label, name, us, reps, 2 * t * 1024 * 256 * 4608 / us / 1e6 * (64 / t)), flush=True)

"""Run ON THE GPU BOX: device time of the frame-resident 1x3x3 `b` convs (s3 / s4 shapes, B = 16) through the C ABI, with the
K loop's barrier stagger off / on (AF_G_STAGGER), 200 back-to-back launches each (a sustained-load figure: inside the model
these launches alternate with memory-bound ones and run a few percent faster)."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hip_helpers as hh
from exp_conv111 import timeit
from exp_variants import mk, layer, L
import exp_variants
B = 16
CASES = [("s4 b 256->256 14x14", mk(B, 16, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1))),
         ("s3 b 128->128 28x28", mk(B, 16, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)))]
for dt in ("bf16", "f16"):
    exp_variants.DT = dt
    for name, d in CASES:
        d.dtype = L.DTYPE_CODES[dt]
        run = layer(d)
        flop = 2.0 * d.n * d.to * d.ho * d.wo * d.cout * d.cin * 9
        line = "%-22s %s" % (name, dt)
        for stg in ("0", "1", "0", "1"):
            os.environ["AF_G_STAGGER"] = stg
            us = timeit(run, 200)
            line += " | stagger=%s %6.1f us %5.0f TF" % (stg, us, flop / us * 1e-6)
        print(line, flush=True)

"""Run ON THE GPU BOX with AF_HIP_LIB=<package>/libafhip_stamps.so (tools/stamps_lib.sh af_conv133g): phase times of the
frame-resident 1x3x3 kernel from in-kernel s_memtime stamps (prologue = first DMA issue -> operands landed, K loop, epilogue),
median over the work units of the last of 300 back-to-back launches, and the shader clock the chip held (s_memtime / s_memrealtime)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_variants
from exp_variants import mk, layer, L
from exp_conv111 import timeit
B = 16
CASES = [("s4 b 256->256 14x14", mk(B, 16, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)), 256),
         ("s3 b 128->128 28x28", mk(B, 16, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)), 512)]
exp_variants.DT = "bf16"
for name, d, units in CASES:
    d.dtype = L.DTYPE_CODES["bf16"]
    run = layer(d)
    buf = torch.zeros(units * 8 * 8, dtype=torch.int64, device="cuda")
    os.environ["AF_STAMP_PTR"] = hex(buf.data_ptr())
    for stg, dbg in (("0", 0), ("1", 0), ("1", 1), ("1", 2), ("1", 4), ("1", 20), ("1", 23), ("0", 23)):
        os.environ["AF_G_STAGGER"] = stg
        os.environ["AF_G_DBG"] = str(dbg)
        us = timeit(run, 300)
        torch.cuda.synchronize()
        s = buf.cpu().view(units, 8, 8).double()
        pro, loop, epi, tot = s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2], s[:, :, 3] - s[:, :, 0]
        clk = tot / (s[:, :, 7] - s[:, :, 6]) * 0.1        # GHz: s_memrealtime ticks at 100 MHz
        med = lambda t, w: t[:, w].median().item()
        span = (s[:, :, 3].max() - s[:, :, 0].min()).item()
        print("%s stagger=%s dbg=%-2d launch %.1f us | cycles (median over units) wave0: prologue %.0f loop %.0f epilogue %.0f total %.0f | wave4: %.0f %.0f %.0f %.0f | clock %.2f GHz | first start -> last end %.0f cycles"
              % (name, stg, dbg, us, med(pro, 0), med(loop, 0), med(epi, 0), med(tot, 0), med(pro, 4), med(loop, 4), med(epi, 4), med(tot, 4), clk.median().item(), span), flush=True)

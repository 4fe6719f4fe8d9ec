"""Run ON THE GPU BOX with AF_HIP_LIB=<package>/libafhip_stamps.so (tools/stamps_lib.sh af_conv): phase times of the generic kernel's
two-input (projection block) form - c conv + stride-2 shortcut in one accumulator - from in-kernel stamps, B = 16."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_variants
from exp_variants import mk, layer, L
from exp_conv111 import timeit
B = 16
CASES = [("#11 s3.res0.c+br1", mk(B, 16, 28, 28, 128, 512), mk(B, 16, 56, 56, 256, 512, s=(1, 2, 2)), 1792, 6),
         ("#23 s4.res0.c+br1", mk(B, 16, 14, 14, 256, 1024), mk(B, 16, 28, 28, 512, 1024, s=(1, 2, 2)), 896, 12),
         ("#41 s5.res0.c+br1", mk(B, 16, 7, 7, 512, 2048), mk(B, 16, 14, 14, 1024, 2048, s=(1, 2, 2)), 448, 24)]
exp_variants.DT = "bf16"
for name, d, d2, wgs, ksteps in CASES:
    d.dtype = L.DTYPE_CODES["bf16"]; d2.dtype = d.dtype
    run = layer(d, d2)
    buf = torch.zeros(wgs * 8 * 8, dtype=torch.int64, device="cuda")
    os.environ["AF_STAMP_PTR"] = hex(buf.data_ptr())
    for dbg in (0, 0):
        os.environ["AF_G_DBG"] = str(dbg)
        us = timeit(run, 300)
        torch.cuda.synchronize()
        s = buf.cpu().view(wgs, 8, 8).double()
        pro, loop, epi, tot = s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2], s[:, :, 3] - s[:, :, 0]
        setup = s[:, :, 0] - s[:, :, 4]
        clk = tot / (s[:, :, 7] - s[:, :, 6]) * 0.1
        med = lambda t, w: t[:, w].median().item()
        span = (s[:, :, 3].max() - s[:, :, 0].min()).item()
        print("%-22s dbg=%d launch %6.1f us | cycles wave0: set-up %6.0f prologue %6.0f loop %7.0f (%5.0f / K-step) epilogue %6.0f total %7.0f | wave4 loop %7.0f | clock %.2f GHz | all workgroups, first start -> last end %.0f"
              % (name, dbg, us, med(setup, 0), med(pro, 0), med(loop, 0), med(loop, 0) / ksteps, med(epi, 0), med(tot, 0), med(loop, 4), clk.median().item(), span), flush=True)

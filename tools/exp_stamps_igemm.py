"""Run ON THE GPU BOX with AF_HIP_LIB=<package>/libafhip_stamps.so (tools/stamps_lib.sh af_conv): phase times of the generic
implicit-GEMM kernel on its long-K layers (B = 16) from in-kernel stamps, with timing-only ablations of its K loop (AF_G_DBG:
1 no vmcnt wait, 2 no barrier, 4 no DMA)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_variants
from exp_variants import mk, layer, L
from exp_conv111 import timeit
B = 16
CASES = [("#24 s4 a 1x1x1 1024->256", mk(B, 16, 14, 14, 1024, 256), None, 224, 16),
         ("#10 s3.res0.b s2 128->128", mk(B, 16, 56, 56, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)), None, 392, 18),
         ("#43 s5 b 512->512 7x7", mk(B, 16, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)), None, 196, 72),
         ("#22 s4.res0.b s2", mk(B, 16, 28, 28, 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)), None, 224, 36)]
exp_variants.DT = "bf16"
for name, d, d2, wgs, ksteps in CASES:
    d.dtype = L.DTYPE_CODES["bf16"]
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
        print("%-36s dbg=%d launch %6.1f us | cycles wave0: set-up %5.0f prologue %6.0f loop %7.0f (%5.0f / K-step) epilogue %6.0f total %7.0f | wave4 loop %7.0f | clock %.2f GHz"
              % (name, dbg, us, med(setup, 0), med(pro, 0), med(loop, 0), med(loop, 0) / ksteps, med(epi, 0), med(tot, 0), med(loop, 4), clk.median().item()), flush=True)

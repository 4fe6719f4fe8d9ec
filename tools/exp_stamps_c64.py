"""Run ON THE GPU BOX with AF_HIP_LIB=<package>/libafhip_stamps.so (tools/stamps_lib.sh af_conv133): where a wave of the s2 `b`
kernel (conv133_c64x2) spends its cycles, by phase, summed over its 28 strips (median over workgroups), with timing-only ablations."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_variants
from exp_variants import mk, layer, L
from exp_conv111 import timeit
B = 16
d = mk(B, 32, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
exp_variants.DT = "bf16"
d.dtype = L.DTYPE_CODES["bf16"]
run = layer(d)
units = 256
buf = torch.zeros(units * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["AF_STAMP_PTR"] = hex(buf.data_ptr())
names = ["prologue", "vmcnt", "barrier1", "issue", "mfma", "barrier2", "epilogue", "tail"]
for form, dbg in (("8", 0), ("8", 0), ("8", 1), ("8", 2), ("8", 4), ("8", 7)):
    os.environ["AF_C64_DBG"] = str(dbg)
    os.environ["AF_C64_WAVES"] = form
    us = timeit(run, 200)
    torch.cuda.synchronize()
    s = buf.cpu().view(units, 8, 8).double()
    for w in (0, 4):
        print("form=" + form + " dbg=%d launch %.1f us wave %d: " % (dbg, us, w) + "  ".join("%s %.0f" % (n, s[:, w, i].median().item()) for i, n in enumerate(names))
              + "  | total %.0f cycles" % s[:, w, :].sum(dim=1).median().item(), flush=True)

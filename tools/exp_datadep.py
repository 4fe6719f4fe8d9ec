"""Run ON THE GPU BOX: is the sustained rate of an MFMA-bound tile data dependent (bit toggling -> power -> clocks)?
The 256x256 tile, K = 4608, full chip, with random / post-ReLU-like / constant / zero operands."""
import os, sys, ctypes as C, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hip_helpers as hh
from exp_conv111 import timeit
L = hh.lib()

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout.strip().split("\n")[-1].split(",")
        return "sclk %s power %s W" % (out[5], out[-1])
    except Exception as e:
        return "rocm-smi failed"

def build(kind):
    n, t, h, w, cin, cout = 1, 64, 32, 32, 4608, 256
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, cin, cout
    d.kt = d.kh = d.kw = 1; d.st = d.sh = d.sw = 1; d.pt = d.ph = d.pw = 0
    d.to, d.ho, d.wo = t, h, w
    d.relu, d.dtype, d.tpool = 1, L.DTYPE_CODES["bf16"], 0
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, t, h, w, cin, device="cuda", generator=g)
    wt = torch.randn(cout, cin, 1, 1, 1) * 0.05
    if kind == "relu": x = torch.relu(x)
    if kind == "const": x = torch.full_like(x, 0.75); wt = torch.full_like(wt, 0.03125)
    if kind == "zero": x = torch.zeros_like(x); wt = torch.zeros_like(wt)
    x = x.to(torch.bfloat16)
    wp = hh._pack_plain(wt, "bf16")
    o = torch.empty(n, t, h, w, cout, device="cuda", dtype=torch.bfloat16)
    sc = torch.ones(cout, device="cuda"); sf = torch.zeros(cout, device="cuda")
    def run():
        L.check(L.lib.af_conv3d_bn_act(C.byref(d), hh._p(x), hh._p(wp), hh._p(sc), hh._p(sf), None, hh._p(o), 0, None, 0, hh._stream()), "conv")
    run.keep = (x, wp, o, sc, sf)
    return run

for kind in ("randn", "relu", "const", "zero", "randn"):
    run = build(kind)
    us = timeit(run, 400)
    print("%-6s operands: %7.1f us per launch = %5.0f TFLOP/s   (%s right after)" % (kind, us, 2 * 65536 * 256 * 4608 / us / 1e6, smi()), flush=True)

"""Run ON THE GPU BOX: device time of single conv layers through the C ABI (the s4 / s3 / s2 `c` convs by default); `layer` and
`timeit` are also used by exp_tile_overhead.py / exp_variants.py.  (The ablation numbers in DESIGN 3.1d came from temporary
kernel switches that are not in the tree.)"""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_helpers as hh
L = hh.lib()

def layer(n, t, h, w, cin, cout, res=True, dtype="bf16", tpool=0):
    code = L.DTYPE_CODES[dtype]
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, cin, cout
    d.kt = d.kh = d.kw = 1; d.st = d.sh = d.sw = 1; d.pt = d.ph = d.pw = 0
    d.to, d.ho, d.wo = t, h, w
    d.relu, d.dtype, d.tpool = 1, code, tpool
    x = torch.randn(n, t, h, w, cin, device="cuda").to(hh.TORCH_DT[dtype])
    r = torch.randn(n, t, h, w, cout, device="cuda").to(hh.TORCH_DT[dtype]) if res else None
    o = torch.empty(n, t // 2 if tpool else t, h, w, cout, device="cuda", dtype=hh.TORCH_DT[dtype])
    wt = hh._pack_plain(torch.randn(cout, cin, 1, 1, 1) * 0.05, dtype)
    sc = torch.ones(cout, device="cuda"); sf = torch.zeros(cout, device="cuda")
    name = L.lib.af_conv_variant_name(L.lib.af_conv_variant(C.byref(d), None)).decode()
    def run():
        L.check(L.lib.af_conv3d_bn_act(C.byref(d), hh._p(x), hh._p(wt), hh._p(sc), hh._p(sf), hh._p(r), hh._p(o), 0, None, 0, hh._stream()), "conv")
    return run, name

def timeit(run, reps=200):
    for _ in range(20): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

if __name__ == "__main__":
    for shape, tp in [((16, 16, 14, 14, 256, 1024), 0), ((16, 16, 28, 28, 128, 512), 0), ((16, 32, 56, 56, 64, 256), 1)]:
        run, name = layer(*shape, tpool=tp)
        print(shape, "tpool", tp, name, "%7.1f us  %7.1f us" % (timeit(run), timeit(run)), flush=True)

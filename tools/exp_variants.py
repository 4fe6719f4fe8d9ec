"""Run ON THE GPU BOX: generic-kernel tile variant sweep per layer shape (temporary AF_FORCE_VAR override in pick_variant)."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hip_helpers as hh
from exp_conv111 import timeit
L = hh.lib()
DT = "bf16"

def mk(n, t, h, w, cin, cout, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0)):
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, cin, cout
    d.kt, d.kh, d.kw = k; d.st, d.sh, d.sw = s; d.pt, d.ph, d.pw = p
    d.to, d.ho, d.wo = [(a + 2 * pp - kk) // ss + 1 for a, pp, kk, ss in zip((t, h, w), p, k, s)]
    d.relu, d.dtype, d.tpool = 1, L.DTYPE_CODES[DT], 0
    return d

def layer(d, d2=None, res=False):
    x = torch.randn(d.n, d.t, d.h, d.w, d.cin, device="cuda").to(hh.TORCH_DT[DT])
    wt = hh._pack_plain(torch.randn(d.cout, d.cin, d.kt, d.kh, d.kw) * 0.05, DT)
    o = torch.empty(d.n, d.to, d.ho, d.wo, d.cout, device="cuda", dtype=hh.TORCH_DT[DT])
    r = torch.randn_like(o) if res else None
    sc = torch.ones(L.lib.af_padded_channels(d.cout), device="cuda"); sf = torch.zeros_like(sc)
    if d2 is None:
        def run():
            L.check(L.lib.af_conv3d_bn_act(C.byref(d), hh._p(x), hh._p(wt), hh._p(sc), hh._p(sf), hh._p(r), hh._p(o), 0, None, 0, hh._stream()), "conv")
    else:
        x2 = torch.randn(d2.n, d2.t, d2.h, d2.w, d2.cin, device="cuda").to(hh.TORCH_DT[DT])
        w2 = hh._pack_plain(torch.randn(d2.cout, d2.cin, 1, 1, 1) * 0.05, DT)
        def run():
            L.check(L.lib.af_conv3d_dual_bn_act(C.byref(d), hh._p(x), hh._p(wt), C.byref(d2), hh._p(x2), hh._p(w2), hh._p(sc), hh._p(sf), hh._p(o), 0, hh._stream()), "dual")
    run.keep = (x, wt, o, r, sc, sf)
    return run

def main():
    B = 16
    CASES = [
        ("#11 s3.res0.c+br1", mk(B, 16, 28, 28, 128, 512), mk(B, 16, 56, 56, 256, 512, s=(1, 2, 2)), False, (6, 0, 5, 2)),
        ("#23 s4.res0.c+br1", mk(B, 16, 14, 14, 256, 1024), mk(B, 16, 28, 28, 512, 1024, s=(1, 2, 2)), False, (6, 0, 5, 2)),
        ("#41 s5.res0.c+br1", mk(B, 16, 7, 7, 512, 2048), mk(B, 16, 14, 14, 1024, 2048, s=(1, 2, 2)), False, (6, 0, 5, 2)),
        ("#12 s3 a 512->128", mk(B, 16, 28, 28, 512, 128), None, False, (0, 2, 5, 7)),
        ("#24 s4 a 1024->256", mk(B, 16, 14, 14, 1024, 256), None, False, (6, 12, 0, 5, 2)),
        ("#39 s5.res0.a 1024->512", mk(B, 16, 14, 14, 1024, 512), None, False, (6, 12, 0, 5, 2)),
        ("#44 s5 c 512->2048 +res", mk(B, 16, 7, 7, 512, 2048), None, True, (2, 5, 0, 6, 12)),
        ("#45 s5 a 2048->512", mk(B, 16, 7, 7, 2048, 512), None, False, (0, 6, 12, 2, 5)),
        ("#10 s3.res0.b s2", mk(B, 16, 56, 56, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)), None, False, (7, 0, 2)),
        ("#22 s4.res0.b s2", mk(B, 16, 28, 28, 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)), None, False, (6, 12, 0, 2)),
        ("#40 s5.res0.b s2", mk(B, 16, 14, 14, 512, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1)), None, False, (0, 6, 12, 2)),
        ("#43 s5 b", mk(B, 16, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)), None, False, (0, 6, 12, 2)),
        ("#42 s5 a 3x1x1", mk(B, 16, 7, 7, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)), None, False, (0, 6, 12, 2)),
        ("#21 s4.res0.a 3x1x1", mk(B, 16, 28, 28, 512, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)), None, False, (6, 0, 2)),
        ("#27 s4 a 3x1x1", mk(B, 16, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)), None, False, (6, 0, 2)),
    ]
    names = {0: "128x256", 1: "64x256", 2: "128x128", 3: "64x128", 5: "128x128r2", 6: "256x256", 7: "128x512", 12: "256x224"}
    for name, d, d2, res, variants in CASES:
        os.environ.pop("AF_FORCE_VAR", None)
        run = layer(d, d2, res)
        cur = L.lib.af_conv_variant_name(L.lib.af_conv_variant(C.byref(d), C.byref(d2) if d2 else None)).decode()
        line = "%-26s default %-28s %6.1f us |" % (name, cur, timeit(run, 100))
        for v in variants:
            os.environ["AF_FORCE_VAR"] = str(v)
            try:
                line += " %s %6.1f" % (names[v], timeit(run, 100))
            except Exception as e:
                line += " %s ERR" % names[v]
        print(line, flush=True)

if __name__ == "__main__":
    main()

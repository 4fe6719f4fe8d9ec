"""Run ON THE GPU BOX: af_tstem_conv_bn_pool_relu_maxpool against the oracle on one small case, printing where they differ."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, torch.nn.functional as F
import hip_helpers as hh
import i3d_oracle as oracle
from af_mi355x import synth
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
with_nan = len(sys.argv) < 3 or sys.argv[2] != "nonan"
n, t, h, w, kt = 1, 6, 20, 28, 5
seed = 5151 + kt + w
lay = [("conv.weight", (64, 3, kt, 1, 1), "float32"), ("bn.0.weight", (64,), "float32"), ("bn.0.bias", (64,), "float32"),
       ("bn.0.running_mean", (64,), "float32"), ("bn.0.running_var", (64,), "float32")]
sd = synth.fill_layout(lay, seed)
sd["conv.weight"] = sd["conv.weight"] * 3.0
sd["bn.0.weight"][1::2] *= -1.0
sd["bn.0.weight"][4] = 0.0
x = synth.synthetic_tensor((n, 3, t, h, w), seed)
if with_nan:
    x[0, 1, t // 2, 5, 7] = float("nan")
x = x.to(hh.TORCH_DT[dtype]).float()
sd["conv.weight"] = sd["conv.weight"].to(hh.TORCH_DT[dtype]).float()
half = oracle._conv_bn_pool_act(x.double(), sd["conv.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn", (kt // 2, 0, 0), True, True)
want = F.max_pool3d(half, (1, 3, 3), (1, 2, 2), (0, 1, 1))
L = hh.lib(); code = L.DTYPE_CODES[dtype]
stem_in = hh.pack_input_f32(x.cuda(), dtype)
scale, shift = hh.fold_bn(sd, "bn.0")
packed = torch.empty(L.lib.af_packed_tstem_weight_bytes(code) // 2, dtype=hh.TORCH_DT[dtype], device="cuda")
L.check(L.lib.af_pack_tstem_weight(hh._p(sd["conv.weight"].float().cuda().contiguous()), 64, kt, code, hh._p(packed), hh._stream()), "pack")
d = L.ConvDesc()
d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, 3, 64
d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, 1, 1, 1, 1, 1, kt // 2, 0, 0
d.to, d.ho, d.wo, d.relu, d.dtype = t, h // 2, w // 2, 1, code
hq, wq = (h // 2 - 1) // 2 + 1, (w // 2 - 1) // 2 + 1
out = torch.full((n, t, hq, wq, 64), 7.0, dtype=hh.TORCH_DT[dtype], device="cuda")
L.check(L.lib.af_tstem_conv_bn_pool_relu_maxpool(C.byref(d), hh._p(stem_in), hh._p(packed), hh._p(scale), hh._p(shift), hh._p(out), hh._stream()), "tstem_pool3")
got = hh.to_ncdhw(out).double().cpu()
nw, ng = torch.isnan(want), torch.isnan(got)
print("want NaN %d, got NaN %d, mismatching %d" % (nw.sum(), ng.sum(), (nw != ng).sum()))
mm = (nw != ng).nonzero()
print(" first mismatches (n,c,t,qy,qx):", mm[:12].tolist())
if mm.numel():
    print(" channels of mismatches:", sorted(set(mm[:, 1].tolist()))[:40])
    print(" got-not-want:", int((ng & ~nw).sum()), " want-not-got:", int((nw & ~ng).sum()))
ok = ~(nw | ng)
e = (got - want)[ok].abs()
print("max err on the rest %.3e (max %.2f); seven-count %d" % (e.max(), want[ok].abs().max(), int((got == 7.0).sum())))
bad = ((got - want).abs() > 0.05 * want[ok].abs().max()) & ok
print(" bad values %d; first:" % bad.sum(), bad.nonzero()[:8].tolist())

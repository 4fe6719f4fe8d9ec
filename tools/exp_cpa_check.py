"""Run ON THE GPU BOX: conv_cpa against the fp64 oracle on one case, printing the trunk / a-output errors (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import hip_helpers as hh
import i3d_oracle as oracle
from af_mi355x import synth

dtype, x_sub = sys.argv[1], int(sys.argv[2])
n, t, h, w, ctrunk = 3, 32, 56, 56, 256
seed = 77
lay = [("c.weight", (ctrunk, 64, 1, 1, 1), "float32"), ("a.weight", (128, ctrunk, 3, 1, 1), "float32")]
for p_, ch in (("c_bn", ctrunk), ("a_bn", 128)):
    lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
sd = synth.fill_layout(lay, seed)
tdt = hh.TORCH_DT[dtype]
b = synth.synthetic_tensor((n, 64, t, h, w), seed).to(tdt).float()
res = synth.synthetic_tensor((n, ctrunk, t, h, w), seed + 1).to(tdt).float()
for k in ("c.weight", "a.weight"):
    sd[k] = sd[k].to(tdt).float()
sd64 = {k: v.double() for k, v in sd.items()}
x = F.relu(oracle.conv_bn_act(b.double(), sd64["c.weight"], sd64, "c_bn", (1, 1, 1), (0, 0, 0), False) + res.double())
xp = F.max_pool3d(x, (2, 1, 1), (2, 1, 1))
want_a = oracle.conv_bn_act(xp.to(tdt).double(), sd64["a.weight"], sd64, "a_bn", (1, 1, 1), (1, 0, 0), True)
want_x = xp[..., ::2, ::2] if x_sub == 2 else xp
for rep in range(3):
    out = hh.conv_cpa(hh.to_ndhwc(b, dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"), hh.to_ndhwc(res, dtype), sd["a.weight"],
                      hh.fold_bn(sd, "a_bn"), dtype, x_sub)
    gx, ga = hh.to_ncdhw(out[0]).double().cpu(), hh.to_ncdhw(out[1]).double().cpu()
    ex, ea = (gx - want_x).abs(), (ga - want_a).abs()
    print("rep %d trunk err %.3e (max %.2f) bad %d  a err %.3e (max %.2f) bad %d" % (
        rep, ex.max(), want_x.abs().max(), (ex > 0.05).sum(), ea.max(), want_a.abs().max(), (ea > 0.05 * want_a.abs().max()).sum()))
    if (ex > 0.05).any():
        idx = (ex > 0.05).nonzero()
        print("  first bad trunk (n,c,t,h,w):", idx[:6].tolist(), " t hist", torch.bincount(idx[:, 2], minlength=16).tolist(),
              " c//64 hist", torch.bincount(idx[:, 1] // 64, minlength=4).tolist())

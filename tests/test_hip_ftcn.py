"""GPU: the reference's second plugin, FTCN-TT, on the HIP kernels against the golden
logits / stage samples / head known-answer produced by the reference plugin itself (tests/golden/f6_ftcn*), plus
the plugin-specific kernels against the CPU oracle on small cases.
Tolerances: f32 2e-4 (north star 1e-3), f16 1e-3 (= the north star), bf16 2e-3 (measured 4.5e-4..5.1e-4: the fp32 head helps; 1e-2 before round 4) on an O(1) logit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_json, load_npz
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import i3d_oracle as oracle  # noqa: E402
import hip_helpers as hh  # noqa: E402
from af_mi355x import arch, synth  # noqa: E402
from af_mi355x.classifier import FtcnTT8x8, FtcnTTClassifier  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ftcn_weights():
    g = load_json("f6_ftcn.json")
    sd = synth.synthetic_state_dict(arch.ftcn_tt_spec(), seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"]
    return g, sd


def _clip(c):
    u8 = synth.synthetic_clips_u8(c["index"] + 1, seed=c["seed"], kind=c["kind"])[c["index"]:c["index"] + 1]
    assert synth.tensor_sha256(u8) == c["clip_sha256"]
    return synth.normalize_like_callers(u8).cuda()


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("f16", 1e-3), ("bf16", 2e-3)])
def test_ftcn_logits_match_reference(ftcn_weights, dtype, tol):
    g, sd = ftcn_weights
    net = FtcnTT8x8(precision=dtype)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    for c in g["clips"]:
        with torch.inference_mode():
            y = net(_clip(c))["final_output"]
        assert y.shape == (1, 1)
        err = abs(float(y[0, 0]) - c["logit_f32"])
        print("ftcn_tt %s %s: hip %.6f ref %.6f |d| %.2e" % (dtype, c["kind"], float(y[0, 0]), c["logit_f32"], err))
        assert err <= tol


def test_ftcn_stage_activations_and_tokens_f32(ftcn_weights):
    g, sd = ftcn_weights
    st = load_npz("f6_ftcn_stages.npz")
    net = FtcnTT8x8(precision="f32")
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x = _clip(g["clips"][0])                  # stays alive: run_prefix re-reads the bound input
    with torch.inference_mode():
        net(x)
    eng = net._engines[("f32", 1, (32, 224, 224))]
    names = eng.op_names
    # last op of each stage: the pooled stem, then the last c conv of s2 (its fused temporal pool is NOT part of the
    # reference's s2 output, so s2 is checked through s3), s3, s4
    last = {"s1": 2, "s3": max(i for i, n in enumerate(names) if n.startswith("resnet.s3.")),
            "s4": max(i for i, n in enumerate(names) if n.startswith("resnet.s4."))}
    for name, i in last.items():
        eng.run_prefix(i + 1)
        act = eng.activation(i).permute(0, 4, 1, 2, 3).contiguous().float().cpu()
        assert list(act.shape) == list(st[name + "_shape"]), (name, act.shape)
        got = act.flatten()[torch.from_numpy(st[name + "_idx"])].numpy()
        want = st[name + "_val"]
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), name
    eng.run_prefix(eng.n_ops)
    tok = eng.tokens_pooled.cpu()
    assert list(tok.shape) == list(st["tokens_shape"])
    got = tok.flatten()[torch.from_numpy(st["tokens_idx"])].numpy()
    assert np.abs(got - st["tokens_val"]).max() <= 1e-4 * max(1.0, float(np.abs(st["tokens_val"]).max()))


def test_ftcn_head_known_answer(ftcn_weights):
    """The transformer head alone: the reference's real token matrix in, its logit out (fp32 head in every mode)."""
    g, sd = ftcn_weights
    st = load_npz("f6_ftcn_stages.npz")
    net = FtcnTT8x8(precision="bf16")
    net.load_state_dict(sd)
    net = net.cuda().eval()
    with torch.inference_mode():
        net(_clip(g["clips"][0]))
    eng = net._engines[("bf16", 1, (32, 224, 224))]
    first = eng.op_names.index("tt_head.tokens")
    with torch.inference_mode():              # the engine's buffers were created under inference_mode
        eng.tokens_pooled.copy_(torch.from_numpy(st["head_tokens"]).cuda())
    from af_mi355x._lib import Op, check, lib
    ops = (Op * (eng.n_ops - first))(*[eng.ops[i] for i in range(first, eng.n_ops)])
    check(lib.af_run_ops(ops, eng.n_ops - first, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "af_run_ops")
    torch.cuda.synchronize()
    got = eng.logits.cpu().numpy()
    np.testing.assert_allclose(got, st["head_logit"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("dims,kt", [((1, 6, 10, 14), 5), ((2, 3, 8, 32), 5), ((1, 4, 6, 8), 3)])
def test_temporal_stem_vs_oracle(dtype, dims, kt):
    """Conv3d(3->64,[kt,1,1]) + BN + MaxPool3d((1,2,2)) + ReLU, incl. a ragged last tile (pooled width % 4 != 0), BN scales of
    both signs (the kernel pools BEFORE the affine map: max for a non-negative scale, min for a negative one) and a NaN
    pixel (every pooled output whose window or taps touch it is NaN, like ATen's conv + max_pool; the rest is untouched)."""
    n, t, h, w = dims
    seed = 4242 + kt + w
    lay = [("conv.weight", (64, 3, kt, 1, 1), "float32"), ("bn.0.weight", (64,), "float32"), ("bn.0.bias", (64,), "float32"),
           ("bn.0.running_mean", (64,), "float32"), ("bn.0.running_var", (64,), "float32")]
    sd = synth.fill_layout(lay, seed)
    sd["conv.weight"] = sd["conv.weight"] * 3.0                     # fan-out init is tiny for a 1x1 spatial kernel
    sd["bn.0.weight"][1::2] *= -1.0
    sd["bn.0.weight"][4] = 0.0
    x = synth.synthetic_tensor((n, 3, t, h, w), seed)
    x[0, 1, t // 2, 3, 5] = float("nan")
    if dtype != "f32":
        x = x.to(hh.TORCH_DT[dtype]).float()
        sd["conv.weight"] = sd["conv.weight"].to(hh.TORCH_DT[dtype]).float()
    want = oracle._conv_bn_pool_act(x.double(), sd["conv.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn",
                                    (kt // 2, 0, 0), True, True)
    L = hh.lib()
    code = L.DTYPE_CODES[dtype]
    stem_in = hh.pack_input_f32(x.cuda(), dtype)
    scale, shift = hh.fold_bn(sd, "bn.0")
    nbytes = L.lib.af_packed_tstem_weight_bytes(code)
    packed = torch.empty(nbytes // (4 if dtype == "f32" else 2), dtype=hh.TORCH_DT[dtype], device="cuda")
    wsrc = sd["conv.weight"].float().cuda().contiguous()
    L.check(L.lib.af_pack_tstem_weight(hh._p(wsrc), 64, kt, code, hh._p(packed), hh._stream()), "pack_tstem_weight")
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, 3, 64
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, 1, 1, 1, 1, 1, kt // 2, 0, 0
    d.to, d.ho, d.wo, d.relu, d.dtype = t, h // 2, w // 2, 1, code
    out = torch.full((n, t, h // 2, w // 2, 64), float("nan"), dtype=hh.TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_tstem_conv_bn_pool_relu(C.byref(d), hh._p(stem_in), hh._p(packed), hh._p(scale), hh._p(shift), hh._p(out),
                                             hh._stream()), "tstem")
    got = hh.to_ncdhw(out).double()
    tol = {"f32": 2e-6, "f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    nan = torch.isnan(want)
    assert nan.any() and not nan.all() and (torch.isnan(got) == nan).all()
    err = (got - want)[~nan].abs().max().item()
    assert err <= tol * (want[~nan].abs().max().item() + 1e-9), err


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dims,kt", [((1, 6, 20, 28), 5), ((2, 3, 16, 72), 5), ((1, 4, 10, 12), 3), ((1, 2, 14, 18), 1)])
def test_temporal_stem_with_fused_stem_pool_vs_oracle(dtype, dims, kt):
    """af_tstem_conv_bn_pool_relu_maxpool: the temporal stem with the stem's own MaxPool3d([1,3,3],[1,2,2],[0,1,1]) behind it in
    one launch, against the oracle's stem followed by F.max_pool3d: odd half-resolution sizes (the window's last row / column
    falls outside the map: clamped members), a ragged last tile, BN scales of both signs and a zero scale, a NaN pixel."""
    import torch.nn.functional as F
    n, t, h, w = dims
    seed = 5151 + kt + w
    lay = [("conv.weight", (64, 3, kt, 1, 1), "float32"), ("bn.0.weight", (64,), "float32"), ("bn.0.bias", (64,), "float32"),
           ("bn.0.running_mean", (64,), "float32"), ("bn.0.running_var", (64,), "float32")]
    sd = synth.fill_layout(lay, seed)
    sd["conv.weight"] = sd["conv.weight"] * 3.0
    sd["bn.0.weight"][1::2] *= -1.0
    sd["bn.0.weight"][4] = 0.0
    x = synth.synthetic_tensor((n, 3, t, h, w), seed)
    x[0, 1, t // 2, 5, 7] = float("nan")
    x = x.to(hh.TORCH_DT[dtype]).float()
    sd["conv.weight"] = sd["conv.weight"].to(hh.TORCH_DT[dtype]).float()
    half = oracle._conv_bn_pool_act(x.double(), sd["conv.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn",
                                    (kt // 2, 0, 0), True, True)
    want = F.max_pool3d(half, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    L = hh.lib()
    code = L.DTYPE_CODES[dtype]
    stem_in = hh.pack_input_f32(x.cuda(), dtype)
    scale, shift = hh.fold_bn(sd, "bn.0")
    nbytes = L.lib.af_packed_tstem_weight_bytes(code)
    packed = torch.empty(nbytes // 2, dtype=hh.TORCH_DT[dtype], device="cuda")
    wsrc = sd["conv.weight"].float().cuda().contiguous()
    L.check(L.lib.af_pack_tstem_weight(hh._p(wsrc), 64, kt, code, hh._p(packed), hh._stream()), "pack_tstem_weight")
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, 3, 64
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, 1, 1, 1, 1, 1, kt // 2, 0, 0
    d.to, d.ho, d.wo, d.relu, d.dtype = t, h // 2, w // 2, 1, code
    hq, wq = (h // 2 - 1) // 2 + 1, (w // 2 - 1) // 2 + 1
    assert tuple(want.shape[2:]) == (t, hq, wq)
    out = torch.full((n, t, hq, wq, 64), 7.0, dtype=hh.TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_tstem_conv_bn_pool_relu_maxpool(C.byref(d), hh._p(stem_in), hh._p(packed), hh._p(scale), hh._p(shift),
                                                     hh._p(out), hh._stream()), "tstem_pool3")
    got = hh.to_ncdhw(out).double()
    tol = {"f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    nan = torch.isnan(want)
    assert nan.any() and not nan.all() and (torch.isnan(got) == nan).all()
    err = (got - want)[~nan].abs().max().item()
    assert err <= tol * (want[~nan].abs().max().item() + 1e-9), err


def test_ftcn_plugin_surface_batch_and_hook(ftcn_weights, tmp_path):
    """ModelBase-style lifecycle of the FTCN-TT plugin; batch invariance; the last nn.Linear (mlp_head.1) is hookable
    like the reference's (feature.py:105-114)."""
    g, sd = ftcn_weights
    path = os.path.join(tmp_path, "ftcn.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, path)
    clf = FtcnTTClassifier(precision="f16").cuda().eval()
    assert clf.load(path) == (True, -1)
    x = torch.cat([_clip(g["clips"][0]), _clip(g["clips"][1])])
    with torch.inference_mode():
        yb = clf(x)["final_output"]
        y0 = clf(x[:1])["final_output"]
    assert yb.shape == (2, 1) and torch.allclose(yb[:1], y0, rtol=0, atol=2e-3)     # f16; batch sizes may split K differently
    assert abs(float(yb[1, 0]) - g["clips"][1]["logit_f32"]) <= 1e-3
    last = [m for m in clf.network.modules() if isinstance(m, torch.nn.Linear)][-1]
    seen = {}
    hdl = last.register_forward_hook(lambda m, i, o: seen.update(i=i[0].detach(), o=o.detach()))
    with torch.inference_mode():
        yh = clf(x)["final_output"]
    hdl.remove()
    assert seen["i"].shape == (2, 1024) and torch.allclose(yh, yb, atol=1e-5)

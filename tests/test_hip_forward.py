"""GPU: the whole HIP forward behind the drop-in Classifier against the golden logits / stage
activations produced by the imported reference (tests/golden/f1_logits.json, f2_stages.npz).

Stated tolerances on the O(0.3) logits of the synthetic checkpoint W(0) (north_star: "logits within 1e-3 of the CPU
reference"):
  f32  : asserted at 2e-4 (exact-fp32 MFMA path; measures 1.5e-6)
  f16  : asserted at 1e-3 - the north-star tolerance (measures 2.5e-4..3.2e-4); the reference's own GPU deployment
         precision (torch.amp.autocast fp16, test/af_realtime.py:70,84)
  bf16 : BF16_TOL = 3e-3 (round 4; 1e-2 before): the measured bound - 9e-4..1.6e-3 on the golden clips and at B = 16 - with a
         margin under 2x, so that a regression of the kernels shows.  bf16 (8 significant bits of weight) does NOT reliably meet
         1e-3: the reference itself under CPU bf16 autocast is 4.4e-3..1e-2 off.  Documented, not hidden; f16 is
         the 16-bit mode that meets the north-star tolerance.
On the "hot" checkpoint (|logit| 18..33, tests/golden/f1b_logits.json) the 16-bit bounds are relative to |logit|: bf16 7e-3
(measured 4.8e-3; 1.2e-2 before).  The shrunken 8-frame 64 x 64 network of the all-dtypes tests measures 3.7e-3 in bf16 (different
weights, three clips) and is asserted at 6e-3.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT, load_json, load_npz

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import i3d_oracle as oracle  # noqa: E402
from af_mi355x import synth  # noqa: E402
from af_mi355x.classifier import Classifier  # noqa: E402

pytestmark = pytest.mark.gpu
BF16_TOL = 3e-3
LOGIT_TOL = {"f32": 2e-4, "f16": 1e-3, "bf16": BF16_TOL}
REL_TOL = {"f32": 1e-5, "f16": 1.5e-3, "bf16": 7e-3}       # hot checkpoint: |d| <= REL_TOL * |logit|


@pytest.fixture(scope="module")
def ckpt_path(weights0):
    d = tempfile.mkdtemp()
    p = os.path.join(d, "w0.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in weights0.items()}}, p)   # wrapped + prefixed form
    return p


@pytest.fixture(scope="module")
def clf32(ckpt_path):
    clf = Classifier(precision="f32").to("cuda").eval()
    ok, epoch = clf.load(ckpt_path)
    assert (ok, epoch) == (True, -1)
    return clf


def _golden_clip(c):
    u8 = synth.synthetic_clips_u8(c["index"] + 1, seed=c["seed"], kind=c["kind"])[c["index"]:c["index"] + 1]
    assert synth.tensor_sha256(u8) == c["clip_sha256"]
    return u8


def test_f32_logits_match_reference(clf32, golden_f1):
    for c in golden_f1["clips"]:
        x = synth.normalize_like_callers(_golden_clip(c)).cuda()
        with torch.inference_mode():
            out = clf32(x)
        assert set(out) == {"final_output"}
        y = out["final_output"]
        assert y.shape == (1, 1) and y.dtype == torch.float32 and y.is_cuda
        err = abs(float(y[0, 0]) - c["logit_f32"])
        print("f32 %s[%d]: hip %.7f ref %.7f |d| %.2e" % (c["kind"], c["index"], float(y[0, 0]), c["logit_f32"], err))
        assert err <= LOGIT_TOL["f32"]


def test_f32_stage_activations_match_reference(clf32, golden_f1):
    f2 = load_npz("f2_stages.npz")
    c = golden_f1["clips"][0]
    x = synth.normalize_like_callers(_golden_clip(c)).cuda()
    net = clf32.network
    with torch.inference_mode():
        clf32(x)
    eng = net._engines[("f32", 1, (32, 224, 224))]
    names = eng.op_names
    last = lambda pfx: max(i for i, n in enumerate(names) if n.startswith(pfx))      # noqa: E731
    stage_ops = {"s1": 2, "s3": last("resnet.s3."), "s4": last("resnet.s4."), "s5": last("resnet.s5.")}
    if eng.ops[last("resnet.s2.")].conv.tpool:          # pathway0_pool rides in s2's last conv: s2 itself never hits HBM
        stage_ops["pool"] = last("resnet.s2.")
    else:
        stage_ops.update({"s2": last("resnet.s2."), "pool": last("resnet.s2.") + 1})
    for st, op_i in stage_ops.items():
        eng.run_prefix(op_i + 1)
        act = eng.activation(op_i).permute(0, 4, 1, 2, 3).contiguous().float().cpu()      # -> NCDHW
        assert list(act.shape) == list(f2[st + "_shape"]), st
        flat = act.flatten()
        want = f2[st + "_val"]
        got = flat[torch.from_numpy(f2[st + "_idx"])].numpy()
        scale = float(f2[st + "_stats"][2])
        assert np.abs(got - want).max() <= 1e-4 * max(scale, 1.0), (st, np.abs(got - want).max())
        np.testing.assert_allclose(float(flat.double().abs().mean()), f2[st + "_stats"][1], rtol=1e-5)
    eng.run_prefix(eng.n_ops)
    pooled = eng.pooled.flatten().cpu()
    np.testing.assert_allclose(pooled[torch.from_numpy(f2["avgpool_idx"])].numpy(), f2["avgpool_val"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_reduced_precision_logits(ckpt_path, golden_f1, dtype):
    clf = Classifier(precision=dtype).to("cuda").eval()
    assert clf.load(ckpt_path)[0]
    for c in golden_f1["clips"]:
        x = synth.normalize_like_callers(_golden_clip(c)).cuda()
        with torch.inference_mode():
            y = clf(x)["final_output"]
        err = abs(float(y[0, 0]) - c["logit_f32"])
        print("%s %s[%d]: hip %.6f ref %.6f |d| %.2e" % (dtype, c["kind"], c["index"], float(y[0, 0]), c["logit_f32"], err))
        assert err <= LOGIT_TOL[dtype]


def test_batch_invariance_and_u8_prologue(clf32, golden_f1):
    u8 = synth.synthetic_clips_u8(2, seed=2026, kind="uniform").cuda()
    x = synth.normalize_like_callers(u8)
    with torch.inference_mode():
        yb = clf32(x)["final_output"]
        y0 = clf32(x[0:1])["final_output"]
        y1 = clf32(x[1:2])["final_output"]
        yu = clf32.network.forward_clips_u8(u8)["final_output"]
        yc = clf32(x.contiguous())["final_output"]                 # NCDHW-contiguous input (feature.py:123)
    # clips are independent units; across batch SIZES the deep stages may split K differently (one clip: fp32 partial
    # sums over K ranges), so the logits agree to fp32 rounding, not bit for bit
    assert torch.allclose(yb, torch.cat([y0, y1]), rtol=0, atol=2e-5)
    assert torch.equal(yb, yu) and torch.equal(yb, yc)
    ref = golden_f1["batch2_uniform_logits_f32"]
    assert max(abs(float(yb[i, 0]) - ref[i]) for i in range(2)) <= LOGIT_TOL["f32"]
    probs = torch.sigmoid(yb).squeeze(1).cpu()
    assert torch.allclose(probs, oracle.scores(yb.cpu()))


def test_full_batch16_properties(clf32):
    """BASELINE config[1] size (B=16): permuting clips permutes logits; duplicated clips in one batch give identical bits."""
    base = synth.synthetic_clips_u8(4, seed=77, kind="smooth").cuda()
    idx = torch.tensor([0, 1, 2, 3, 3, 2, 1, 0, 0, 0, 1, 1, 2, 2, 3, 3], device="cuda")
    with torch.inference_mode():
        y4 = clf32.network.forward_clips_u8(base)["final_output"]
        y16 = clf32.network.forward_clips_u8(base[idx].contiguous())["final_output"]
    assert y16.shape == (16, 1)
    assert torch.allclose(y16, y4[idx], rtol=0, atol=2e-5)            # different batch sizes: fp32 rounding (split-K)
    assert torch.equal(y16[3], y16[4]) and torch.equal(y16[0], y16[7]) and torch.equal(y16[8], y16[9])
    assert torch.isfinite(y16).all()


def test_head_linear_hook_sees_pooled_feature(clf32, golden_f1):
    lin = [m for m in clf32.modules() if isinstance(m, torch.nn.Linear)][-1]
    seen = {}
    h = lin.register_forward_hook(lambda m, i, o: seen.update(inp=i[0].detach(), out=o.detach()))
    x = synth.normalize_like_callers(_golden_clip(golden_f1["clips"][0])).cuda()
    with torch.inference_mode():
        y = clf32(x)["final_output"]
    h.remove()
    assert tuple(seen["inp"].shape) == (1, 1, 1, 1, 2048)
    assert abs(float(y[0, 0]) - golden_f1["clips"][0]["logit_f32"]) <= LOGIT_TOL["f32"]
    with torch.inference_mode():
        y2 = clf32(x)["final_output"]
    assert abs(float(y2[0, 0]) - float(y[0, 0])) <= 1e-5


def test_autocast_selects_fp16_engine(ckpt_path, golden_f1):
    clf = Classifier().to("cuda").eval()                # precision="auto"
    clf.load(ckpt_path)
    x = synth.normalize_like_callers(_golden_clip(golden_f1["clips"][2])).cuda()
    with torch.inference_mode():
        with torch.amp.autocast("cuda"):
            y = clf(x)["final_output"]
    assert ("f16", 1, (32, 224, 224)) in clf.network._engines
    assert y.dtype == torch.float32
    assert abs(float(y[0, 0]) - golden_f1["clips"][2]["logit_f32"]) <= LOGIT_TOL["f16"]


def test_reload_repacks_weights(clf32, ckpt_path, golden_f1):
    x = synth.normalize_like_callers(_golden_clip(golden_f1["clips"][0])).cuda()
    sd1 = synth.synthetic_state_dict(seed=1)
    clf32.network.load_state_dict(sd1)
    with torch.inference_mode():
        y1 = float(clf32(x)["final_output"][0, 0])
    clf32.load(ckpt_path)
    with torch.inference_mode():
        y0 = float(clf32(x)["final_output"][0, 0])
    assert abs(y0 - golden_f1["clips"][0]["logit_f32"]) <= LOGIT_TOL["f32"]
    assert abs(y1 - y0) > 1e-3


def test_edge_inputs(clf32, weights0):
    """empty batch; odd batch; a crop larger than 224 (the reference then returns one logit per head position,
    head_helper.py:54,94); too few frames for the head pool -> error; fp16 input tensor."""
    with torch.inference_mode():
        y = clf32(torch.zeros((0, 3, 32, 224, 224), device="cuda"))["final_output"]
        assert y.shape == (0, 1)
        u8 = synth.synthetic_clips_u8(3, seed=5, kind="smooth")
        x3 = synth.normalize_like_callers(u8).cuda()
        y3 = clf32(x3)["final_output"]
        y1 = clf32(x3[2:3])["final_output"]
        assert y3.shape == (3, 1) and torch.allclose(y3[2:3].float(), y1.float(), rtol=0, atol=2e-3)   # batch sizes may split K differently
        yh = clf32(x3[:1].half())["final_output"]                      # fp16 tensor in: promoted, not rejected
        assert abs(float(yh[0, 0]) - float(y3[0, 0])) < 5e-3
        # 256x256 crop -> s5 is 8x8 -> AvgPool3d([16,7,7], stride 1) leaves 2x2 positions -> (B, 4) logits
        u8b = synth.synthetic_clips_u8(1, seed=6, kind="smooth", size=256)
        xb = synth.normalize_like_callers(u8b)
        got = clf32(xb.cuda())["final_output"].cpu()
        want = oracle.forward(weights0, xb, num_frames=32, crop=224)
        assert got.shape == want.shape == (1, 4)
        assert (got - want).abs().max().item() <= 2e-4
        with pytest.raises(ValueError, match="too small"):
            clf32(torch.zeros((1, 3, 16, 224, 224), device="cuda"))


def test_small_network_vs_oracle_all_dtypes():
    """Shrunken clip (8 frames, 64x64, head pool [4,2,2]) through all 53 convs, every dtype, vs the oracle."""
    clip_size, size = 8, 64
    from af_mi355x.arch import i3d_r50_spec
    sd = synth.synthetic_state_dict(i3d_r50_spec(clip_size, size), seed=5)
    u8 = synth.synthetic_clips_u8(3, seed=9, kind="smooth", num_frames=clip_size, size=size)
    x = synth.normalize_like_callers(u8)
    want = oracle.forward(sd, x, num_frames=clip_size, crop=size)
    for dtype, tol in (("f32", 1e-4), ("f16", 1e-3), ("bf16", 6e-3)):
        clf = Classifier(clip_size=clip_size, precision=dtype, crop_size=size)
        clf.network.load_state_dict(sd)
        clf = clf.to("cuda").eval()
        with torch.inference_mode():
            got = clf(x.cuda())["final_output"].cpu()
        err = (got - want).abs().max().item()
        print("small net", dtype, "max|d| %.3e" % err, "logits", want.flatten().tolist())
        assert err <= tol, (dtype, err)


@pytest.mark.parametrize("nan", [float("nan"), -float("nan")], ids=["nan", "nan_with_sign_bit"])
def test_non_finite_pixel_gives_nan_logit_like_reference(nan):
    """A NaN pixel must not turn into a plausible score: in the reference it survives Conv3d, eval BN, ReLU (clamp_min keeps
    NaN), MaxPool3d and AvgPool3d and the clip's logit is NaN; the other clips of the batch are untouched.  Same shrunken
    network as above (every conv kernel family is on the path); the oracle states the expectation.  The NaN with its sign bit
    set is what x86 code hands over for 0/0: the packed 16-bit ReLU would zero it, so the input packers clear the sign."""
    clip_size, size = 8, 64
    from af_mi355x.arch import i3d_r50_spec
    sd = synth.synthetic_state_dict(i3d_r50_spec(clip_size, size), seed=5)
    u8 = synth.synthetic_clips_u8(3, seed=9, kind="smooth", num_frames=clip_size, size=size)
    x = synth.normalize_like_callers(u8)
    x[1, 2, 3, 17, 40] = nan
    want = oracle.forward(sd, x, num_frames=clip_size, crop=size)
    assert torch.isnan(want[1]).all() and torch.isfinite(want[[0, 2]]).all()
    for dtype, tol in (("f32", 1e-4), ("f16", 1e-3), ("bf16", 6e-3)):
        clf = Classifier(clip_size=clip_size, precision=dtype, crop_size=size)
        clf.network.load_state_dict(sd)
        clf = clf.to("cuda").eval()
        with torch.inference_mode():
            got = clf(x.cuda())["final_output"].cpu()
        assert torch.isnan(got[1]).all(), (dtype, got)
        assert (got[[0, 2]] - want[[0, 2]]).abs().max().item() <= tol, (dtype, got, want)


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
def test_config1_batch16_at_its_own_dtype(weights0, dtype):
    """BASELINE config[1] as benchmarked: B=16 uniform clips (seed 2026 = bench.py's rank-0 batch), one forward, against the
    reference's own B=16 forward (tests/golden/f1b_logits.json).  The engine picks its kernels by batch (stream kernels need
    >= 4 tiles per CU, split-K below 128 workgroups, the 256x256 tile rules), so this is the plan the bench times."""
    g = load_json("f1b_logits.json")["batch16"]
    assert synth.state_dict_sha256(weights0) == g["weights_sha256"]
    u8 = synth.synthetic_clips_u8(16, seed=g["seed"], kind=g["kind"])
    assert synth.tensor_sha256(u8) == g["clips_sha256"]
    clf = Classifier(precision=dtype)
    clf.network.load_state_dict(weights0)
    clf = clf.cuda().eval()
    with torch.inference_mode():
        y = clf.network.forward_clips_u8(u8.cuda())["final_output"].cpu().flatten()
        y2 = clf(synth.normalize_like_callers(u8.cuda()))["final_output"].cpu().flatten()
    want = torch.tensor(g["logits_f32"])
    err = (y - want).abs().max().item()
    print("B=16 %s: max|d| %.3e (tolerance %.1e)" % (dtype, err, LOGIT_TOL[dtype]))
    assert torch.equal(y, y2)                              # uint8 prologue == callers' fp32 normalisation
    assert (dtype, 16, (32, 224, 224)) in clf.network._engines
    assert err <= LOGIT_TOL[dtype], (dtype, err)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_fused_stage_boundary_engine_matches_reference(weights0, dtype, monkeypatch):
    """AF_FUSE_CPA=1: s2's last c conv, pathway0_pool and s3's first a conv run as ONE launch (af_conv3d_cpa_bn_act) and s3's
    projection shortcut reads the packed even-position trunk with stride 1.  Off by default (measured slower), so the
    plan is exercised here: the B=16 golden batch against the reference's own logits, and the op list shows the fused launch."""
    from af_mi355x import _lib
    monkeypatch.setenv("AF_FUSE_CPA", "1")
    g = load_json("f1b_logits.json")["batch16"]
    u8 = synth.synthetic_clips_u8(16, seed=g["seed"], kind=g["kind"])
    clf = Classifier(precision=dtype)
    clf.network.load_state_dict(weights0)
    clf = clf.cuda().eval()
    with torch.inference_mode():
        y = clf.network.forward_clips_u8(u8.cuda())["final_output"].cpu().flatten()
    eng = clf.network._engines[(dtype, 16, (32, 224, 224))]
    kinds = [eng.ops[i].kind for i in range(eng.n_ops)]
    assert kinds.count(_lib.AF_OP_CONV_CPA) == 1
    cpa = eng.ops[kinds.index(_lib.AF_OP_CONV_CPA)]
    assert cpa.x_sub == 2 and cpa.conv2.cout == 128
    err = (y - torch.tensor(g["logits_f32"])).abs().max().item()
    print("B=16 %s with the fused s2 -> s3 boundary: max|d| %.3e" % (dtype, err))
    assert err <= LOGIT_TOL[dtype], (dtype, err)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_hot_checkpoint_logits(dtype):
    """Second weight seed with |logit| 18..33 that moves between clips (F1b "hot"): 8 clips in one batch."""
    g = load_json("f1b_logits.json")["hot"]
    sd = synth.synthetic_state_dict(seed=g["weights_seed"], recipe=g["recipe"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"]
    u8 = torch.cat([synth.synthetic_clips_u8(n, seed=seed, kind=kind) for kind, seed, n in g["clips"]])
    assert synth.tensor_sha256(u8) == g["clips_sha256"]
    clf = Classifier(precision=dtype)
    clf.network.load_state_dict(sd)
    clf = clf.cuda().eval()
    with torch.inference_mode():
        y = clf.network.forward_clips_u8(u8.cuda())["final_output"].cpu().flatten().double()
    want32, want64 = torch.tensor(g["logits_f32"]).double(), torch.tensor(g["logits_f64"]).double()
    rel = ((y - want32).abs() / want32.abs()).max().item()
    print("hot %s: max rel %.3e, abs %.3e; fp32-vs-fp64 of the reference itself %.3e"
          % (dtype, rel, (y - want32).abs().max().item(), (want32 - want64).abs().max().item()))
    assert rel <= REL_TOL[dtype], (dtype, rel)
    if dtype == "f32":                                     # the 1e-3 north-star tolerance holds in absolute terms too
        assert (y - want32).abs().max().item() <= 1e-3


def test_split_k_workspace_is_caller_owned_and_stream_ordered(weights0):
    """The B=1..3 live-call path splits K in the deep stages into a workspace the ENGINE owns (af_conv_workspace_bytes):
    a B=1 engine then a B=3 engine run back to back on one stream with no synchronisation in between - the library
    neither allocates nor synchronises - and both match the oracle.  A second stream gets engines of its own."""
    u8 = synth.synthetic_clips_u8(3, seed=31, kind="smooth")
    x = synth.normalize_like_callers(u8)
    want = oracle.forward(weights0, x).flatten()
    clf = Classifier(precision="f32")
    clf.network.load_state_dict(weights0)
    clf = clf.cuda().eval()
    xd = x.cuda()
    with torch.inference_mode():
        clf(xd[:1]); clf(xd)                                  # build both engines (weights packed, buffers allocated)
        torch.cuda.synchronize()
        e1 = clf.network._engines[("f32", 1, (32, 224, 224))]
        e3 = clf.network._engines[("f32", 3, (32, 224, 224))]
        assert e1.workspace is not None and e3.workspace is not None, "one clip in s5 must take the split-K path"
        assert e1.workspace.data_ptr() != e3.workspace.data_ptr()
        ya = clf(xd[:1])["final_output"]                      # no sync between the two forwards
        yb = clf(xd)["final_output"]
        yc = clf(xd[2:3])["final_output"]
        torch.cuda.synchronize()
    got = torch.cat([ya.flatten(), yb.flatten(), yc.flatten()]).cpu()
    exp = torch.cat([want[:1], want, want[2:3]])
    assert (got - exp).abs().max().item() <= LOGIT_TOL["f32"], (got, exp)


def test_engine_cache_is_bounded(weights0):
    clf = Classifier(clip_size=8, precision="f16", crop_size=64)
    from af_mi355x.arch import i3d_r50_spec
    clf.network.load_state_dict(synth.synthetic_state_dict(i3d_r50_spec(8, 64), seed=5))
    clf = clf.cuda().eval()
    clf.network.max_engines = 3
    with torch.inference_mode():
        ys = {b: clf(torch.zeros((b, 3, 8, 64, 64), device="cuda"))["final_output"].cpu() for b in (1, 2, 3, 4, 5, 1)}
    assert len(clf.network._engines) == 3
    assert list(clf.network._engines)[-1][1] == 1            # most recently used last
    assert torch.allclose(ys[1][0], ys[5][0], atol=2e-3)
    clf.network.invalidate_packed()
    assert not clf.network._engines and not clf.network._packed


def test_infer_scores_matches_callers_epilogue(clf32, weights0):
    """ClassifierSvc.infer_scores (test/af_realtime.py:75-96): uint8 (B,T,H,W,C) clips -> sigmoid(logit), the sigmoid computed by
    the head kernel; numpy in / numpy out like the reference, and the real-valued-pixel path through the callers' normalisation."""
    u8 = synth.synthetic_clips_u8(2, seed=41, kind="smooth")
    want = oracle.scores(oracle.forward(weights0, oracle.normalize(u8)))
    s = clf32.network.infer_scores(u8.numpy())
    assert isinstance(s, np.ndarray) and s.shape == (2,) and s.dtype == np.float32
    assert np.abs(s - want.numpy()).max() <= 1e-5
    s2 = clf32.network.infer_scores(u8.float() + 0.25, as_numpy=False)              # non-integral pixels: fp32 path
    want2 = oracle.scores(oracle.forward(weights0, synth.normalize_like_callers(u8.float() + 0.25)))
    assert s2.is_cuda and (s2.cpu() - want2).abs().max().item() <= 1e-5
    with torch.inference_mode():
        out = clf32.network.forward_clips_u8(u8.cuda(), return_scores=True, return_pooled=True)
    assert out["pooled"].shape == (2, 2048) and torch.allclose(out["scores"], torch.sigmoid(out["final_output"]).view(2), atol=1e-6)


def test_two_stream_mode_matches_single_stream(weights0):
    """Classifier(streams=2): a batch of >= 16 clips runs as two half-batches on two HIP streams (two engines); same logits as the
    single-stream engine up to the kernel choices that depend on the batch size, same golden tolerance; odd batches split 8 + 9."""
    g = load_json("f1b_logits.json")["batch16"]
    u8 = synth.synthetic_clips_u8(16, seed=g["seed"], kind=g["kind"]).cuda()
    want = torch.tensor(g["logits_f32"])
    clf2 = Classifier(precision="f16", streams=2)
    clf2.network.load_state_dict(weights0)
    clf2 = clf2.cuda().eval()
    with torch.inference_mode():
        out = clf2.network.forward_clips_u8(u8, return_scores=True, return_pooled=True)
        y = out["final_output"].cpu().flatten()
        y17 = clf2(synth.normalize_like_callers(torch.cat([u8, u8[:1]])))["final_output"].cpu().flatten()
        torch.cuda.synchronize()
    assert ("f16", 8, (32, 224, 224)) in clf2.network._engines and ("f16", 8, (32, 224, 224), 1) in clf2.network._engines
    assert (y - want).abs().max().item() <= LOGIT_TOL["f16"]
    assert out["scores"].shape == (16,) and out["pooled"].shape == (16, 2048)
    assert torch.allclose(out["scores"].cpu(), torch.sigmoid(y), atol=1e-6)
    assert y17.shape == (17,) and (y17[:16] - want).abs().max().item() <= LOGIT_TOL["f16"] and abs(float(y17[16] - want[0])) <= LOGIT_TOL["f16"]


@pytest.mark.parametrize("batch", [3, 5, 7, 12, 13, 24])
def test_kernel_selection_thresholds_across_batch_sizes(weights0, batch):
    """The engine picks kernels by batch (conv_ca from 3 clips, conv133g for s3 from 6 and for s4 from 12, split-K below, the
    persistent streams from 4 tiles per CU ...): the f16 engine at batch sizes around those thresholds against the exact-fp32
    engine on the same clips (itself pinned to the reference elsewhere in this file)."""
    u8 = synth.synthetic_clips_u8(batch, seed=100 + batch, kind="smooth").cuda()
    out = {}
    for dtype in ("f32", "f16"):
        clf = Classifier(precision=dtype)
        clf.network.load_state_dict(weights0)
        clf = clf.cuda().eval()
        with torch.inference_mode():
            out[dtype] = clf.network.forward_clips_u8(u8)["final_output"].cpu().flatten()
        names = clf.network._engines[(dtype, batch, (32, 224, 224))].op_names
        if dtype == "f16":
            assert any("->" in n for n in names) == (batch >= 3), names       # c(i) -> a(i+1) fused in s2
        del clf
        torch.cuda.empty_cache()
    err = (out["f16"] - out["f32"]).abs().max().item()
    print("B=%d f16 vs f32 engine: max|d| %.3e" % (batch, err))
    assert err <= LOGIT_TOL["f16"], (batch, err)


def test_f16_overflow_gives_non_finite_logit_like_fp16_autocast():
    """The f16 engine stores activations as fp16 like the reference under torch.amp.autocast (test/af_realtime.py:70,84): a
    conv output beyond 65 504 becomes inf there and the clip's logit ends up inf / NaN.  The engine must do the same - never
    a plausible number from a silently saturated activation - and must leave the other clips of the batch alone.  Set-up:
    the shrunken network with s3.res0's a_bn gain x8 and clip 1's pixels x1e4 (still fp16-representable): in fp32 the
    clip's s3.res0 `a` output passes 65 504 (asserted on the oracle), clips 0 and 2 stay O(1)."""
    clip_size, size = 8, 64
    from af_mi355x.arch import i3d_r50_spec
    sd = synth.synthetic_state_dict(i3d_r50_spec(clip_size, size), seed=5)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["resnet.s3.pathway0_res0.branch2.a_bn.weight"] *= 8.0
    u8 = synth.synthetic_clips_u8(3, seed=9, kind="smooth", num_frames=clip_size, size=size)
    x = synth.normalize_like_callers(u8)
    x[1] *= 1e4
    assert x.abs().max().item() < 65504.0                                  # the input itself is representable
    want, stages = oracle.forward(sd, x, num_frames=clip_size, crop=size, return_stages=True)
    a_out = oracle.conv_bn_act(stages["pool"], sd["resnet.s3.pathway0_res0.branch2.a.weight"], sd,
                               "resnet.s3.pathway0_res0.branch2.a_bn", (1, 1, 1), (1, 0, 0), relu=True)
    assert a_out[1].max().item() > 65504.0 and a_out[[0, 2]].max().item() < 1e3 and stages["s2"].abs().max().item() < 65504.0
    # the expectation under fp16 STORAGE of every conv output (what autocast's fp16 conv results are), stated by the oracle
    orig = oracle.conv_bn_act
    try:
        oracle.conv_bn_act = lambda *a, **k: orig(*a, **k).half().float()
        want16 = oracle.forward(sd, x, num_frames=clip_size, crop=size)
    finally:
        oracle.conv_bn_act = orig
    assert not torch.isfinite(want16[1]).any() and torch.isfinite(want16[[0, 2]]).all() and torch.isfinite(want).all()
    clf = Classifier(clip_size=clip_size, precision="f16", crop_size=size)
    clf.network.load_state_dict(sd)
    clf = clf.to("cuda").eval()
    with torch.inference_mode():
        got = clf(x.cuda())["final_output"].cpu()
        s = clf.network.forward(x.cuda(), return_scores=True)["scores"].cpu()
    print("f16 overflow: engine", got.flatten().tolist(), "fp16-storage oracle", want16.flatten().tolist(), "fp32 oracle", want.flatten().tolist())
    assert not torch.isfinite(got[1]).any(), got
    assert (got[[0, 2]] - want[[0, 2]]).abs().max().item() <= 1e-3, (got, want)
    assert not (s[1] > 0.0 and s[1] < 1.0) or not torch.isfinite(got[1]).any()    # the score is not a plausible probability either
    # the fp32 engine represents the same activations and stays finite, equal to the fp32 oracle (relative: |logit| is O(1e3))
    clf32 = Classifier(clip_size=clip_size, precision="f32", crop_size=size)
    clf32.network.load_state_dict(sd)
    clf32 = clf32.to("cuda").eval()
    with torch.inference_mode():
        g32 = clf32(x.cuda())["final_output"].cpu()
    assert torch.isfinite(g32).all() and ((g32 - want).abs() / want.abs().clamp_min(1.0)).max().item() <= 1e-4


def test_bench_two_rank_rehearsal_shards_the_real_forward():
    """SURVEY 8e on the REAL forward: `AF_BENCH_REHEARSAL=1 bench.py --gpus 2` starts two rank processes (gloo, both on this
    one GPU: RCCL needs a GPU per rank), each runs its own shard of clips through the HIP forward, the logits are all-gathered
    in clip order; the line must say n_gpus == 2 and the gathered logits must equal a single-process forward of the same
    clips.  (Two child processes + this one on the GPU; run once.)"""
    import json
    import subprocess
    B = 2
    env = dict(os.environ, AF_BENCH_REHEARSAL="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", str(B),
                        "--cpu-clips", "0", "--no-roofline", "--emit-logits"], env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["global_batch"] == 2 * B
    assert "roofline" not in line and "N=1" in line["note"]
    sd = synth.synthetic_state_dict(seed=0)
    clf = Classifier(precision="bf16")
    clf.network.load_state_dict(sd)
    clf = clf.cuda().eval()
    u8 = torch.cat([synth.synthetic_clips_u8(B, seed=2026 + r, kind="uniform") for r in range(2)])     # rank r's batch, in rank order
    with torch.inference_mode():
        ya = clf(synth.normalize_like_callers(u8[:B].cuda()))["final_output"].float().cpu().flatten()
        yb = clf(synth.normalize_like_callers(u8[B:].cuda()))["final_output"].float().cpu().flatten()
    want = torch.cat([ya, yb])
    got = torch.tensor(line["gathered_logits"])
    assert got.shape == want.shape and torch.equal(got, want), (got, want)


def test_live_scorer_graph_replay_equals_infer_scores():
    """LiveScorer: the B = 1 forward recorded once into a HIP graph and replayed per window gives infer_scores' value bit for bit,
    for a clip passed in and for one written into its static input (as the streaming aligner does), several windows in a row."""
    from af_mi355x.classifier import LiveScorer
    clf = Classifier(precision="f16")
    clf.network.load_state_dict(synth.synthetic_state_dict(seed=0))
    clf = clf.cuda().eval()
    scorer = LiveScorer(clf.network)
    with torch.inference_mode():
        for seed in (1, 2, 3):
            u8 = synth.synthetic_clips_u8(1, seed=seed, kind="smooth").cuda()
            want = clf.network.infer_scores(u8)
            got = scorer(u8[0])
            assert got.shape == want.shape and float(got[0]) == float(want[0]), (got, want)
            scorer.clip.copy_(u8)
            assert float(scorer()[0]) == float(want[0])

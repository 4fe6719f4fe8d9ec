"""CPU: the oracle restatement (oracle/i3d_oracle.py) against the golden vectors that
oracle/gen_golden.py produced from the imported reference."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_json, load_npz

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import i3d_oracle as oracle  # noqa: E402
from af_mi355x import arch, synth  # noqa: E402


def test_layout_matches_reference():
    lay = load_json("layout.json")
    mine = [[k, list(s), d] for k, s, d in arch.state_dict_layout(arch.i3d_r50_spec())]
    assert lay["num_keys"] == 320 and lay["num_params"] == 27225921
    assert mine == lay["entries"]


def test_macs_match_survey():
    total, rows = arch.conv_macs_per_clip(arch.i3d_r50_spec())
    assert total == 113627365376 and len(rows) == 53


def _kat_state(case):
    name = case["name"]
    return synth.fill_layout([tuple(e) for e in case["layout"]], case["seed"],
                             final_bn=[name + "." + b for b in case["final_bn"]],
                             linear=[name + "." + l for l in case["linear"]])


def _kat_input(case):
    x = synth.synthetic_tensor(case["in_shape"], case["seed"], case["in_scale"])
    assert synth.tensor_sha256(x) == case["in_sha256"]
    return x


def run_oracle_kat(case):
    sd, x, name = _kat_state(case), _kat_input(case), case["name"]
    kind = case["kind"]
    with torch.no_grad():
        if kind == "stem":
            return oracle.stem(x, sd, name)
        if kind == "block":
            return oracle.res_block(x, sd, name, case["stride"])
        if kind == "maxpool":
            return torch.nn.functional.max_pool3d(x, case["kernel"], case["stride"], case["pad"])
        if kind == "head":
            return oracle.head(x, sd, tuple(case["pool"]), name)
        if kind == "fuse":
            k, a = case["kernel"], case["alpha"]
            return oracle.conv_bn_act(x, sd[name + ".conv_f2s.weight"], sd, name + ".bn", (a, 1, 1), (k // 2, 0, 0), True)
    raise KeyError(kind)


def test_kats_bit_exact(golden_f3):
    cases, arrays = golden_f3
    assert len(cases) >= 13
    for case in cases:
        got = run_oracle_kat(case).numpy()
        want = arrays[case["name"] + "_out"]
        assert got.shape == want.shape, case["name"]
        if case["kind"] == "head":      # torch.cat in the reference changes the GEMV's memory layout: last-bit noise
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-6, err_msg=case["name"])
        else:
            assert np.array_equal(got, want), "%s: max|d|=%g" % (case["name"], np.abs(got - want).max())


def test_full_size_logits_and_stages(golden_f1, weights0):
    f2 = load_npz("f2_stages.npz")
    for ci, c in enumerate(golden_f1["clips"]):
        u8 = synth.synthetic_clips_u8(c["index"] + 1, seed=c["seed"], kind=c["kind"])[c["index"]:c["index"] + 1]
        assert synth.tensor_sha256(u8) == c["clip_sha256"]
        x = oracle.normalize(u8)
        assert torch.equal(x, synth.normalize_like_callers(u8))
        logits, stages = oracle.forward(weights0, x, return_stages=True)
        assert logits.shape == (1, 1)
        assert abs(float(logits[0, 0]) - c["logit_f32"]) <= 1e-5
        assert abs(float(logits[0, 0]) - c["logit_f64"]) <= 1e-4
        if ci == 0:
            for n, t in stages.items():
                assert list(t.shape) == list(f2[n + "_shape"]), n
                flat = t.flatten()
                np.testing.assert_allclose(flat[torch.from_numpy(f2[n + "_idx"])].numpy(), f2[n + "_val"],
                                           rtol=1e-4, atol=1e-5, err_msg=n)
                np.testing.assert_allclose(float(flat.double().abs().mean()), f2[n + "_stats"][1], rtol=1e-5)


def test_slowfast_logits_and_stages():
    """oracle.slowfast_forward against the reference's SlowFast-R50 (tests/golden/f5_slowfast*)."""
    g = load_json("f5_slowfast.json")
    st = load_npz("f5_slowfast_stages.npz")
    spec = arch.slowfast_r50_spec()
    assert g["num_keys"] == 662 == len(arch.state_dict_layout(spec)) and g["num_params"] == 33560521
    sd = synth.synthetic_state_dict(spec, seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"]
    for ci, c in enumerate(g["clips"]):
        u8 = synth.synthetic_clips_u8(c["index"] + 1, seed=c["seed"], kind=c["kind"])[c["index"]:c["index"] + 1]
        assert synth.tensor_sha256(u8) == c["clip_sha256"]
        x = oracle.normalize(u8)
        logits, stages = oracle.slowfast_forward(sd, x[:, :, ::g["alpha"]], x, alpha=g["alpha"], return_stages=True)
        assert logits.shape == (1, 1)
        assert abs(float(logits[0, 0]) - c["logit_f32"]) <= 1e-5
        if ci == 0:
            for name, key in (("s1_fuse", "s1"), ("s2_fuse", "s2"), ("s3_fuse", "s3"), ("s4_fuse", "s4"), ("s5", "s5")):
                for tag, t in zip(("slow", "fast"), stages[key]):
                    assert list(t.shape) == list(st["%s_%s_shape" % (name, tag)]), (name, tag)
                    got = t.flatten()[torch.from_numpy(st["%s_%s_idx" % (name, tag)])].numpy()
                    np.testing.assert_allclose(got, st["%s_%s_val" % (name, tag)], rtol=1e-4, atol=1e-5)


def test_ftcn_tt_logits_stages_and_head():
    """oracle.ftcn_forward against the reference's FTCN-TT plugin (tests/golden/f6_ftcn*)."""
    g = load_json("f6_ftcn.json")
    st = load_npz("f6_ftcn_stages.npz")
    spec = arch.ftcn_tt_spec()
    assert g["num_keys"] == 275 == len(arch.state_dict_layout(spec)) and g["num_params"] == 14765889
    sd = synth.synthetic_state_dict(spec, seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"]
    # the transformer head on its own: the reference's real token matrix -> its logit
    got = oracle.time_transformer(torch.from_numpy(st["head_tokens"]), sd)
    np.testing.assert_allclose(got.numpy(), st["head_logit"], rtol=0, atol=2e-6)
    c = g["clips"][0]
    u8 = synth.synthetic_clips_u8(c["index"] + 1, seed=c["seed"], kind=c["kind"])[c["index"]:c["index"] + 1]
    assert synth.tensor_sha256(u8) == c["clip_sha256"]
    logits, stages = oracle.ftcn_forward(sd, oracle.normalize(u8), return_stages=True)
    assert logits.shape == (1, 1) and abs(float(logits[0, 0]) - c["logit_f32"]) <= 1e-5
    for name in ("s1", "s2", "s3", "s4", "tokens"):
        t = stages[name]
        assert list(t.shape) == list(st["%s_shape" % name]), name
        got = t.flatten()[torch.from_numpy(st["%s_idx" % name])].numpy()
        np.testing.assert_allclose(got, st["%s_val" % name], rtol=1e-4, atol=1e-5)


def test_dualrun_oracle_matches_reference():
    """oracle/dualrun_oracle.py against the reference's DualEncoderAU_LMK (tests/golden/f7_dualrun.*): ragged lengths, no
    lengths, and a clip without a single valid frame."""
    import dualrun_oracle
    from af_mi355x import dualrun
    g = load_json("f7_dualrun.json")
    st = load_npz("f7_dualrun.npz")
    sp = dualrun.DualSpec()
    sd = dualrun.dual_synthetic_state_dict(sp, seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"] and len(sd) == g["num_keys"] == 136
    assert sum(v.numel() for v in sd.values()) == g["num_params"] == 5789989
    for tag, batch, frames in (("b6_t8", 6, 8), ("b3_t8_full", 3, 8), ("b2_t5", 2, 5)):
        A, L, _ = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=g["inputs_seed"])
        ln = st[tag + "_lengths"]
        lengths = None if ln[0] < 0 else torch.from_numpy(ln)
        logits, z = dualrun_oracle.dual_forward(sd, A, L, lengths, heads=sp.heads, tau=sp.pool_tau)
        np.testing.assert_allclose(logits.numpy(), st[tag + "_logits_f32"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(z.numpy(), st[tag + "_z_f32"], rtol=2e-6, atol=5e-6)
    msd = {k[len("moe_w_"):]: torch.from_numpy(st[k]) for k in st.files if k.startswith("moe_w_")}
    zf, gate = dualrun_oracle.gated_moe(msd, torch.from_numpy(st["moe_z_rgb"]), torch.from_numpy(st["moe_z_dual"]))
    np.testing.assert_allclose(zf.numpy(), st["moe_z"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gate.numpy(), st["moe_gate"], rtol=1e-6, atol=1e-7)


def test_checkpoint_unwrap_rules():
    base = {"resnet.a": torch.ones(1)}
    assert list(oracle.strip_checkpoint({"state_dict": {"module.resnet.a": 1}})) == ["resnet.a"]
    assert list(oracle.strip_checkpoint({"module.network.resnet.a": 1})) == ["network.resnet.a"]
    assert list(oracle.strip_checkpoint(base)) == ["resnet.a"]


def test_scores_postprocessing():
    l = torch.tensor([[0.0], [2.0]])
    assert torch.allclose(oracle.scores(l), torch.sigmoid(l[:, 0]))
    l2 = torch.tensor([[0.0, 1.0]])
    assert torch.allclose(oracle.scores(l2), torch.softmax(l2, 1)[:, 1])


ALIGNER_CASES = (("t32_224", 224), ("t1_256", 256), ("t8_mirrored", 224))


def _aligner_infos(g, tag):
    return [(None, l5.copy(), l68.copy(), box.copy()) for l5, l68, box in zip(g[tag + "_ldm5"], g[tag + "_ldm68"], g[tag + "_boxes"])]


def test_aligner_similarity_fit_matches_reference():
    """F8: the clip aligner's similarity fit and landmark transform against the reference's numpy code (incl. the
    reflective branch of findSimilarity and its in-place reflection of the targets); cv2.warpAffine is not covered by
    any reference output - the warp restatement is parity-unpinned (oracle/aligner_oracle.py header)."""
    import aligner_oracle as ao
    g = load_npz("f8_aligner.npz")
    np.testing.assert_allclose(ao.STD_POINTS_256, g["std_points_256"], rtol=0, atol=0)
    for tag, size in ALIGNER_CASES:
        infos = _aligner_infos(g, tag)
        t5, t68 = ao.crop_align(infos, None, size=size, return_ldm5=True)
        np.testing.assert_allclose(t5, g[tag + "_t5"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(t68, g[tag + "_t68"], rtol=1e-10, atol=1e-9)
        boxes = g[tag + "_boxes"]
        diff = boxes[:, :2] - boxes[:, :2].min(0)[None]
        tfm, trans = ao.estimate_batch_transform(g[tag + "_ldm5"] + diff[:, None, :], ao.STD_POINTS_256 * size / 256.0)
        np.testing.assert_allclose(tfm, g[tag + "_tfm"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(trans, g[tag + "_trans"], rtol=1e-10, atol=1e-10)
    assert g["t8_mirrored_tfm"][0, 0] < 0          # the mirrored case really took the reflective solution


def test_aligner_warp_restatement_properties():
    """size-independent properties of the fixed-point bilinear warp (unpinned against cv2): identity and integer
    translations reproduce the source exactly with a zero border; a half-pixel shift is the rounded mean of neighbours;
    the result of a general transform stays within the hull of the 4 taps."""
    import aligner_oracle as ao
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=(40, 52, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    out = ao.warp_affine_u8(src, ident, 64)
    assert (out[:40, :52] == src).all() and (out[40:] == 0).all() and (out[:, 52:] == 0).all()
    out = ao.warp_affine_u8(src, np.array([[1.0, 0, 7], [0, 1.0, 3]]), 64)
    assert (out[3:43, 7:59] == src).all() and (out[:3] == 0).all() and (out[:, :7] == 0).all()
    out = ao.warp_affine_u8(src, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), 64)          # dst(x) samples src(x - 0.5)
    want = (src[:, :-1].astype(np.int64) + src[:, 1:].astype(np.int64) + 1) >> 1
    assert (out[:40, 1:52] == want).all()
    m = np.array([[0.83, -0.21, 5.3], [0.21, 0.83, -2.7]])
    out = ao.warp_affine_u8(np.full((30, 30, 3), 200, dtype=np.uint8), m, 48)
    assert out.max() == 200 and set(np.unique(out)) - {0, 200} != set()                # edges blend towards the border 0


def test_hot_checkpoint_logits_match_reference():
    """F1b: the "hot" checkpoint W(3, hot) - logits O(10..40) that move between clips - on one uniform and one smooth clip."""
    f1b = load_json("f1b_logits.json")["hot"]
    sd = synth.synthetic_state_dict(seed=f1b["weights_seed"], recipe=f1b["recipe"])
    assert synth.state_dict_sha256(sd) == f1b["weights_sha256"]
    u8 = hot_clips(f1b)
    assert synth.tensor_sha256(u8) == f1b["clips_sha256"]
    pick = [0, 5]
    with torch.no_grad():
        y = oracle.forward(sd, oracle.normalize(u8[pick])).flatten()
    want = torch.tensor([f1b["logits_f32"][i] for i in pick])
    assert (y - want).abs().max().item() <= 2e-5 * 40, (y, want)


def hot_clips(f1b):
    return torch.cat([synth.synthetic_clips_u8(n, seed=seed, kind=kind) for kind, seed, n in f1b["clips"]])


def test_dualrun_rgb_oracle_matches_reference():
    """F9: the tri-modal DualEncoderRGB restatement (oracle/dualrun_oracle.dual_rgb_forward) against the reference class."""
    import dualrun_oracle
    from af_mi355x import dualrun
    g = load_json("f9_dualrgb.json")
    st = load_npz("f9_dualrgb.npz")
    sp = dualrun.DualSpec(36, 132, 256, 4, 4, 768, 0.7, 128)
    sd = dualrun.dual_rgb_synthetic_state_dict(sp, 2048, seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"]
    assert [k for k, _ in dualrun.dual_rgb_state_dict_layout(sp, 2048)] == list(sd) and len(sd) == g["num_keys"] == 129
    for tag, batch, frames, tv in (("b6_t8", 6, 8, 8), ("b3_t8_nomask", 3, 8, 8), ("b4_t8_v1", 4, 8, 1)):
        A, L, _ = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=g["inputs_seed"])
        V = torch.rand((batch, tv, 2048), generator=torch.Generator().manual_seed(g["inputs_seed"] - 1 + 77)) * 2.0
        ln = st[tag + "_lengths"]
        pad = None if ln[0] < 0 else dualrun_oracle.lengths_to_mask(torch.from_numpy(ln), frames)
        y, _ = dualrun_oracle.dual_rgb_forward(sd, A, L, V, pad, heads=4, tau=0.7)
        np.testing.assert_allclose(y.numpy(), st[tag + "_logits_f32"], rtol=0, atol=2e-5)
        y64, _ = dualrun_oracle.dual_rgb_forward(sd, A, L, V, pad, heads=4, tau=0.7, dtype=torch.float64)
        np.testing.assert_allclose(y64.numpy(), st[tag + "_logits_f64"], rtol=0, atol=1e-7)

"""GPU: the dualrun AU / landmark dual encoder - one HIP launch per modality + a head
kernel - against the logits / clip vectors of the reference's ``DualEncoderAU_LMK`` (tests/golden/f7_dualrun.*) and the
CPU oracle on further shapes.  fp32 throughout: tolerance 2e-5 on O(1) logits (north star 1e-3)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_json, load_npz
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dualrun_oracle  # noqa: E402
from af_mi355x import dualrun, synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dual():
    g = load_json("f7_dualrun.json")
    sp = dualrun.DualSpec()
    sd = dualrun.dual_synthetic_state_dict(sp, seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"]
    net = dualrun.DualEncoderAU_LMK(au_dim=sp.au_dim, lmk_dim=sp.lmk_dim, d_model=sp.d_model, depth=sp.depth, heads=sp.heads,
                                    mlp_ratio=float(sp.ff) / sp.d_model, pool_tau=sp.pool_tau)
    net.load_state_dict(sd)
    return g, sp, sd, net.cuda().eval()


@pytest.mark.parametrize("tag,batch,frames", [("b6_t8", 6, 8), ("b3_t8_full", 3, 8), ("b2_t5", 2, 5)])
def test_dual_encoder_matches_reference(dual, tag, batch, frames):
    g, sp, sd, net = dual
    st = load_npz("f7_dualrun.npz")
    A, L, _ = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=g["inputs_seed"])
    ln = st[tag + "_lengths"]
    lengths = None if ln[0] < 0 else torch.from_numpy(ln).cuda()
    with torch.inference_mode():
        out = net(A.cuda(), L.cuda(), lengths, return_z=True)
    assert out["dom_logits"] is None and out["bin_logits"].shape == (batch,)
    np.testing.assert_allclose(out["bin_logits"].cpu().numpy(), st[tag + "_logits_f32"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["z"].cpu().numpy(), st[tag + "_z_f32"], rtol=2e-5, atol=1e-4)


@pytest.mark.parametrize("batch,frames", [(1, 16), (5, 12), (16, 8), (3, 1)])
def test_dual_encoder_vs_oracle_other_shapes(dual, batch, frames):
    g, sp, sd, net = dual
    A, L, lengths = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=31 + frames)
    want, wz = dualrun_oracle.dual_forward(sd, A, L, lengths, heads=sp.heads, tau=sp.pool_tau)
    with torch.inference_mode():
        out = net(A.cuda(), L.cuda(), lengths.cuda(), return_z=True)
    np.testing.assert_allclose(out["bin_logits"].cpu().numpy(), want.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["z"].cpu().numpy(), wz.numpy(), rtol=2e-5, atol=1e-4)          # |z| up to ~10; summation order differs


def test_dual_encoder_contract_errors(dual):
    g, sp, sd, net = dual
    A, L, lengths = dualrun.synthetic_dual_inputs(2, sp, frames=8, seed=1)
    with pytest.raises(RuntimeError):
        net(A, L)                                              # CPU tensors: no fallback
    with pytest.raises(ValueError):
        net(A.cuda(), L[:, :7].cuda())
    with pytest.raises(ValueError):
        net(torch.zeros(1, 17, sp.au_dim).cuda(), torch.zeros(1, 17, sp.lmk_dim).cuda())
    with pytest.raises(NotImplementedError):
        net(A.cuda(), L.cuda(), need_aux=True)
    with torch.inference_mode():
        e = net(A[:0].cuda(), L[:0].cuda())
    assert e["bin_logits"].shape == (0,)


def test_gated_moe_matches_reference():
    """GatedMoE (dualrun/rgb/engine_rgb.py:369-384) against the reference class's outputs for seeded parameters."""
    st = load_npz("f7_dualrun.npz")
    moe = dualrun.GatedMoE()
    moe.load_state_dict({k[len("moe_w_"):]: torch.from_numpy(st[k]) for k in st.files if k.startswith("moe_w_")})
    moe = moe.cuda().eval()
    with torch.inference_mode():
        z, g = moe(torch.from_numpy(st["moe_z_rgb"]).cuda(), torch.from_numpy(st["moe_z_dual"]).cuda())
    np.testing.assert_allclose(z.cpu().numpy(), st["moe_z"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(g.cpu().numpy(), st["moe_gate"], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        moe(torch.zeros(2, 1), torch.zeros(2, 1))


# ---- tri-modal DualEncoderRGB (reference dualrun/model/dual_rgb.py:47-122), golden F9 ---------------------------------
RGB_CASES = [("b6_t8", 6, 8, 8), ("b3_t8_nomask", 3, 8, 8), ("b4_t8_v1", 4, 8, 1)]


def rgb_case(g, sp, tag, batch, frames, tv, vis=2048):
    """inputs of a golden F9 case, regenerated from the seeds recorded in f9_dualrgb.json (gen_golden.gen_dualrun_rgb)"""
    st = load_npz("f9_dualrgb.npz")
    A, L, _ = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=g["inputs_seed"])
    V = torch.rand((batch, tv, vis), generator=torch.Generator().manual_seed(g["inputs_seed"] - 1 + 77)) * 2.0
    ln = st[tag + "_lengths"]
    lengths = None if ln[0] < 0 else torch.from_numpy(ln)
    return A, L, V, lengths, st[tag + "_logits_f32"]


@pytest.fixture(scope="module")
def dual_rgb():
    g = load_json("f9_dualrgb.json")
    sp = dualrun.DualSpec(36, 132, 256, 4, 4, 768, 0.7, 128)
    sd = dualrun.dual_rgb_synthetic_state_dict(sp, 2048, seed=g["weights_seed"])
    assert synth.state_dict_sha256(sd) == g["weights_sha256"] and len(sd) == g["num_keys"]
    net = dualrun.DualEncoderRGB(au_dim=36, lmk_dim=132, vis_dim=2048, d_model=256, depth=4, heads=4, ff_dim=3.0)
    net.load_state_dict(sd)
    return g, sp, sd, net.cuda().eval()


@pytest.mark.parametrize("tag,batch,frames,tv", RGB_CASES)
def test_dual_rgb_matches_reference(dual_rgb, tag, batch, frames, tv):
    g, sp, sd, net = dual_rgb
    A, L, V, lengths, want = rgb_case(g, sp, tag, batch, frames, tv)
    mask = None if lengths is None else net.lengths_to_mask(lengths, frames, "cuda")
    with torch.inference_mode():
        y, s = net(A.cuda(), L.cuda(), V.cuda(), key_padding_mask=mask, return_scores=True)
        y2 = net(A.cuda(), L.cuda(), V.cuda(), key_padding_mask=mask)
    assert y.shape == (batch,) and torch.equal(y, y2)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=0, atol=3e-5)
    np.testing.assert_allclose(s.cpu().numpy(), torch.sigmoid(torch.from_numpy(want)).numpy(), rtol=0, atol=1e-5)


def test_dual_rgb_vs_oracle_and_contract(dual_rgb):
    g, sp, sd, net = dual_rgb
    A, L, lengths = dualrun.synthetic_dual_inputs(16, sp, frames=8, seed=5)        # config[3] batch
    V = torch.rand((16, 8, 2048), generator=torch.Generator().manual_seed(9)) * 2.0
    pad = dualrun_oracle.lengths_to_mask(lengths, 8)
    want, _ = dualrun_oracle.dual_rgb_forward(sd, A, L, V, pad, heads=4, tau=0.7)
    with torch.inference_mode():
        y = net(A.cuda(), L.cuda(), V.cuda(), key_padding_mask=pad.cuda())
    np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=0, atol=3e-5)
    with pytest.raises(ValueError, match="suffix"):
        bad = pad.clone(); bad[0, 0] = True; bad[0, 1] = False
        net(A.cuda(), L.cuda(), V.cuda(), key_padding_mask=bad.cuda())
    with pytest.raises(RuntimeError):
        net(A, L, V)                                                               # CPU tensors: no fallback
    with pytest.raises(NotImplementedError):
        net(A.cuda(), L.cuda(), V.cuda(), return_seq=True)
    with pytest.raises(ValueError, match="mlp_ratio"):
        dualrun.DualEncoderRGB(36, 132, 2048)                                      # upstream's default ff_dim=768: 196 608-wide layers


def test_mask_check_is_not_skipped_for_a_new_tensor_at_a_recycled_address(dual_rgb):
    """The padding-mask validity check is remembered per mask TENSOR (identity + version), not per (address, shape): a fresh invalid
    mask that the allocator places where the last valid one lived - what a caller who rebuilds the mask every step produces - must
    still raise (outside inference mode, where tensors carry version counters and the cache is live)."""
    g, sp, sd, net = dual_rgb
    A, L, lengths = dualrun.synthetic_dual_inputs(4, sp, frames=8, seed=11)
    V = torch.rand((4, 8, 2048), generator=torch.Generator().manual_seed(3))
    A, L, V = A.cuda(), L.cuda(), V.cuda()
    with torch.no_grad():
        good = dualrun_oracle.lengths_to_mask(lengths, 8).cuda()
        y0 = net(A, L, V, key_padding_mask=good)
        y1 = net(A, L, V, key_padding_mask=good)                    # same live tensor, unmodified: the cached verdict applies
        assert torch.equal(y0, y1)
        shape = tuple(good.shape)
        del good
        bad = torch.zeros(shape, dtype=torch.bool, device="cuda")   # same shape, usually the freed block of `good`, one in-place write
        bad[:, 0] = True                                            # padding first = not a suffix mask
        with pytest.raises(ValueError, match="suffix"):
            net(A, L, V, key_padding_mask=bad)
        # ... and a mask edited in place after it passed is checked again
        ok = dualrun_oracle.lengths_to_mask(lengths, 8).cuda()
        net(A, L, V, key_padding_mask=ok)
        ok[0, 0] = True; ok[0, 1] = False
        with pytest.raises(ValueError, match="suffix"):
            net(A, L, V, key_padding_mask=ok)


def test_two_stream_model_as_one_unit(dual_rgb):
    """BASELINE config[3] shape: AltFreezing (shrunken clip, fp32) -> pooled 2048-vector -> DualEncoderRGB -> GatedMoE with the
    AltFreezing logit, everything on the device; each stage against its oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import i3d_oracle
    from af_mi355x.arch import i3d_r50_spec
    from af_mi355x.classifier import Classifier
    g, sp, sd, net = dual_rgb
    clip_size, size, B = 8, 64, 4
    sdc = synth.synthetic_state_dict(i3d_r50_spec(clip_size, size), seed=5)
    clf = Classifier(clip_size=clip_size, precision="f32", crop_size=size)
    clf.network.load_state_dict(sdc)
    clf = clf.cuda().eval()
    u8 = synth.synthetic_clips_u8(B, seed=9, kind="smooth", num_frames=clip_size, size=size)
    A, L, lengths = dualrun.synthetic_dual_inputs(B, sp, frames=8, seed=6)
    net.rgb_backbone[0], net.rgb_from_features = clf, False
    try:
        mask = net.lengths_to_mask(lengths, 8, "cuda")
        with torch.inference_mode():
            rgb = clf.network.forward_clips_u8(u8.cuda(), return_scores=True, return_pooled=True)
            z_dual = net(A.cuda(), L.cuda(), u8.cuda(), key_padding_mask=mask)
            # return_rgb: the same call also hands back the backbone's own outputs (one forward for both streams)
            z_dual2, rgb2 = net(A.cuda(), L.cuda(), u8.cuda(), key_padding_mask=mask, return_rgb=True)
            assert torch.equal(z_dual2, z_dual) and torch.equal(rgb2["final_output"], rgb["final_output"])
            assert torch.equal(rgb2["pooled"], rgb["pooled"])
            moe = dualrun.GatedMoE().cuda().eval()
            z, gate = moe(rgb["final_output"], z_dual.view(B, 1))
    finally:
        net.rgb_backbone[0], net.rgb_from_features = None, True
    x = synth.normalize_like_callers(u8)
    want_logit, stages = i3d_oracle.forward(sdc, x, num_frames=clip_size, crop=size, return_stages=True)
    feat = stages["avgpool"].reshape(B, 1, -1)
    assert (rgb["pooled"].cpu() - feat.view(B, -1)).abs().max().item() <= 1e-4
    assert (rgb["scores"].cpu() - i3d_oracle.scores(want_logit)).abs().max().item() <= 1e-5
    want_dual, _ = dualrun_oracle.dual_rgb_forward(sd, A, L, feat, dualrun_oracle.lengths_to_mask(lengths, 8), heads=4, tau=0.7)
    np.testing.assert_allclose(z_dual.cpu().numpy(), want_dual.numpy(), rtol=0, atol=1e-4)
    msd = {k: v.detach().cpu() for k, v in moe.state_dict().items()}
    wz, wg = dualrun_oracle.gated_moe(msd, want_logit, want_dual.view(B, 1))
    np.testing.assert_allclose(z.cpu().numpy(), wz.numpy(), rtol=0, atol=2e-4)


@pytest.fixture(scope="module")
def config3_oracle(dual_rgb):
    """BASELINE config[3] at its own size: bench.py's rank-0 batch (16 uniform uint8 clips, seed 2026, W(0)); the CPU oracles
    (AltFreezing fp32 forward -> pooled feature -> DualEncoderRGB -> GatedMoE) on the first 4 clips."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import i3d_oracle
    g, sp, sd, net = dual_rgb
    B, n = 16, 4
    sdc = synth.synthetic_state_dict(seed=0)
    u8 = synth.synthetic_clips_u8(B, seed=2026, kind="uniform")
    A, L, lengths = dualrun.synthetic_dual_inputs(B, sp, frames=8, seed=2026)
    want_logit, stages = i3d_oracle.forward(sdc, synth.normalize_like_callers(u8[:n]), return_stages=True)
    feat = stages["avgpool"].reshape(n, 1, -1)
    want_dual, _ = dualrun_oracle.dual_rgb_forward(sd, A[:n], L[:n], feat, dualrun_oracle.lengths_to_mask(lengths[:n], 8), heads=4, tau=0.7)
    return sdc, u8, A, L, lengths, want_logit, feat.view(n, -1), want_dual


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_config3_full_size_two_stream(dual_rgb, config3_oracle, dtype):
    """The composite bench.py --model dualrun_rgb times, as one model at its own size: 16 uint8 clips of 32x224x224 ->
    forward_clips_u8(return_pooled=True) on the 16-bit trunk -> DualEncoderRGB (V = the pooled feature) -> GatedMoE with the
    AltFreezing logit (reference dualrun/model/dual_rgb.py:47-122, rgb/engine_rgb.py:369-404); every stage against its oracle
    on 4 of the clips: AltFreezing logit <= 1e-3 (f16) / the bf16 bound, pooled feature, dual logit, fused logit."""
    from af_mi355x.classifier import Classifier
    g, sp, sd, net = dual_rgb
    sdc, u8, A, L, lengths, want_logit, want_feat, want_dual = config3_oracle
    B, n = 16, 4
    clf = Classifier(precision=dtype)
    clf.network.load_state_dict(sdc)
    clf = clf.cuda().eval()
    mask = net.lengths_to_mask(lengths, 8, "cuda")
    moe = dualrun.GatedMoE().cuda().eval()
    with torch.inference_mode():
        rgb = clf.network.forward_clips_u8(u8.cuda(), return_pooled=True)
        z_dual = net(A.cuda(), L.cuda(), rgb["pooled"].view(B, 1, -1), key_padding_mask=mask)
        z, gate = moe(rgb["final_output"], z_dual.view(B, 1))
    assert (dtype, B, (32, 224, 224)) in clf.network._engines                      # the B=16 plan, not a shrunken one
    logit_tol = {"f16": 1e-3, "bf16": 1e-2}[dtype]                                 # north-star tolerance / tests' BF16_TOL
    feat_tol = {"f16": 2e-3, "bf16": 2e-2}[dtype]                                  # pooled features are O(1): relative to their scale
    e_logit = (rgb["final_output"][:n].cpu() - want_logit).abs().max().item()
    e_feat = (rgb["pooled"][:n].cpu() - want_feat).abs().max().item() / want_feat.abs().max().item()
    e_dual = (z_dual[:n].cpu() - want_dual).abs().max().item()
    msd = {k: v.detach().cpu() for k, v in moe.state_dict().items()}
    wz, _ = dualrun_oracle.gated_moe(msd, want_logit, want_dual.view(n, 1))
    e_fused = (z[:n].cpu() - wz).abs().max().item()
    print("config[3] %s: logit %.2e pooled(rel) %.2e dual %.2e fused %.2e" % (dtype, e_logit, e_feat, e_dual, e_fused))
    assert torch.isfinite(z).all() and z.shape == (B, 1)
    assert e_logit <= logit_tol and e_feat <= feat_tol, (e_logit, e_feat)
    # the dual logit sees the trunk's rounding only through rgb_proj of the pooled feature; the fused logit mixes both
    assert e_dual <= logit_tol and e_fused <= logit_tol, (e_dual, e_fused)

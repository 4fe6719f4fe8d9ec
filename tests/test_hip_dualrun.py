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

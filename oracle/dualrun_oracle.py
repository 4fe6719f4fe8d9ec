"""TEST INFRASTRUCTURE - CPU restatement of the reference's dualrun AU / landmark dual encoder
(dualrun/model/dual_encoder.py), written against ``torch.nn.functional`` only.  Pinned by tests/golden/f7_dualrun.*
(generated from the imported reference by oracle/gen_golden.py --dualrun).  Only tests/, smoke() and bench.py's
cpu_baseline leg may use it; the product path (dualrun.py -> csrc/af_dual.hip) never does."""
import math

import torch
import torch.nn.functional as F


def sinusoid(frames, d_model, dtype):
    # PositionalEncoding (dual_encoder.py:16-27)
    pe = torch.zeros(frames, d_model)
    pos = torch.arange(0, frames, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def lengths_to_mask(lengths, frames):
    """True = padding (dual_encoder.py:137-156); a clip with no valid frame keeps frame 0 (:162-166)."""
    if lengths is None:
        return None
    pad = torch.arange(frames).expand(lengths.numel(), frames) >= lengths.view(-1, 1)
    allp = pad.all(dim=1)
    pad[allp, 0] = False
    return pad


def branch_encoder(x, sd, p, heads, tau, pad=None):
    """BranchEncoder.forward (dual_encoder.py:73-107) -> clip vector (B, D)."""
    w = lambda k: sd[p + "." + k].to(x.dtype)
    B, T, _ = x.shape
    D = w("proj.weight").shape[0]
    h = F.layer_norm(F.linear(x, w("proj.weight"), w("proj.bias")), (D,), w("ln_in.weight"), w("ln_in.bias"), 1e-5)
    delta = torch.cat([torch.zeros_like(h[:, :1]), h[:, 1:] - h[:, :-1]], dim=1)
    hc = h.transpose(1, 2)
    highp = (hc - F.avg_pool1d(hc, kernel_size=5, stride=1, padding=2)).transpose(1, 2)
    h = h + 0.5 * delta + 0.5 * highp
    hc = h.transpose(1, 2)
    pyr = sum(F.conv1d(hc, w("temporal.%d.weight" % i), w("temporal.%d.bias" % i), padding=d, dilation=d, groups=D)
              for i, d in enumerate((1, 2, 4)))
    hc = F.gelu(F.conv1d(pyr + hc, w("pointwise.weight"), w("pointwise.bias")))
    h = hc.transpose(1, 2) + sinusoid(T, D, x.dtype)
    dh = D // heads
    layer = 0
    while (p + ".encoder.layers.%d.norm1.weight" % layer) in sd:            # nn.TransformerEncoderLayer, norm_first, gelu
        q_ = "encoder.layers.%d." % layer
        y = F.layer_norm(h, (D,), w(q_ + "norm1.weight"), w(q_ + "norm1.bias"), 1e-5)
        qkv = F.linear(y, w(q_ + "self_attn.in_proj_weight"), w(q_ + "self_attn.in_proj_bias"))
        q, k, v = [t.reshape(B, T, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
        s = q @ k.transpose(-1, -2) / math.sqrt(dh)
        if pad is not None:
            s = s.masked_fill(pad[:, None, None, :], float("-inf"))
        o = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
        h = h + F.linear(o, w(q_ + "self_attn.out_proj.weight"), w(q_ + "self_attn.out_proj.bias"))
        y = F.layer_norm(h, (D,), w(q_ + "norm2.weight"), w(q_ + "norm2.bias"), 1e-5)
        h = h + F.linear(F.gelu(F.linear(y, w(q_ + "linear1.weight"), w(q_ + "linear1.bias"))),
                         w(q_ + "linear2.weight"), w(q_ + "linear2.bias"))
        layer += 1
    scores = (h @ w("pool.v")) / max(float(tau), 1e-3)                       # AttentionPooling (:30-47)
    if pad is not None:
        scores = scores.masked_fill(pad, torch.finfo(scores.dtype).min)
    wts = torch.softmax(scores, dim=1)
    return (wts.unsqueeze(-1) * h).sum(dim=1)


def dual_forward(sd, A, L, lengths=None, heads=4, tau=1.0, dtype=torch.float32):
    """DualEncoderAU_LMK.forward (dual_encoder.py:158-198), inference outputs: (bin_logits (B,), z (B, 2D))."""
    A, L = A.to(dtype), L.to(dtype)
    pad = lengths_to_mask(lengths, A.shape[1])
    with torch.no_grad():
        z = torch.cat([branch_encoder(A, sd, "au_enc", heads, tau, pad), branch_encoder(L, sd, "lmk_enc", heads, tau, pad)], dim=-1)
        w = lambda k: sd[k].to(dtype)
        y = F.layer_norm(z, (z.shape[-1],), w("head.0.weight"), w("head.0.bias"), 1e-5)
        y = F.gelu(F.linear(y, w("head.1.weight"), w("head.1.bias")))
        logits = F.linear(y, w("head.4.weight"), w("head.4.bias")).squeeze(-1)
    return logits, z


def gated_moe(sd, z_rgb, z_dual):
    """GatedMoE.forward (dualrun/rgb/engine_rgb.py:376-384) -> (fused logit, gate)."""
    x = torch.cat([z_rgb, z_dual, torch.abs(z_rgb - z_dual)], dim=1)
    g = torch.sigmoid(F.linear(F.relu(F.linear(x, sd["gate.0.weight"], sd["gate.0.bias"])), sd["gate.2.weight"], sd["gate.2.bias"]))
    p = g * torch.sigmoid(z_rgb / sd["t_rgb"].clamp_min(1.0)) + (1 - g) * torch.sigmoid(z_dual / sd["t_dual"].clamp_min(0.1))
    return torch.log((p + 1e-6) / (1 - p + 1e-6)), g


def dual_rgb_forward(sd, A, L, V, pad=None, heads=4, tau=0.7, dtype=torch.float32):
    """DualEncoderRGB.forward (dualrun/model/dual_rgb.py:91-122) with AltFreezingRGBEncoder.from_features (:27-44):
    (bin_logits (B,), z (B, 3D)).  ``pad``: (B,T) bool key-padding mask (True = padding) or None; ``tau`` = BranchEncoder's
    default pool_tau 0.7 (the constructor does not pass one, :59-60)."""
    A, L, V = A.to(dtype), L.to(dtype), V.to(dtype)
    with torch.no_grad():
        za = branch_encoder(A, sd, "au_enc", heads, tau, pad)
        zl = branch_encoder(L, sd, "lmk_enc", heads, tau, pad)
        if pad is None:
            zv_clip = V.mean(dim=1)
        else:
            valid = (~pad).to(dtype)
            w = (valid / valid.clamp_min(1e-6).sum(dim=1, keepdim=True)).unsqueeze(-1)
            zv_clip = (V * w).sum(dim=1)
        zv = F.linear(zv_clip, sd["rgb_proj.weight"].to(dtype))
        z = torch.cat([za, zl, zv], dim=-1)
        w_ = lambda k: sd[k].to(dtype)
        y = F.layer_norm(z, (z.shape[-1],), w_("head.0.weight"), w_("head.0.bias"), 1e-5)
        y = F.gelu(F.linear(y, w_("head.1.weight"), w_("head.1.bias")))
        logits = F.linear(y, w_("head.4.weight"), w_("head.4.bias")).squeeze(-1)
    return logits, z

"""dualrun AU/landmark dual encoder on the MI355X: drop-in for the reference's
``DualEncoderAU_LMK`` (dualrun/model/dual_encoder.py:110-198; built by dualrun/cli/run.py:175-187 and cli/best.py:367-378
from ``checkpoints/*/args.json``: d_model 256, 4 layers, 4 heads, ff_dim 768, T = 8 frames of 36 action units and 132
landmark coordinates per clip).

Each modality runs through a ``BranchEncoder`` (:53-107): Linear + LayerNorm, a first-difference / moving-average
high-pass mix, a dilated depthwise Conv1d pyramid + pointwise Conv1d + GELU, sinusoidal positions, ``depth`` pre-norm
``nn.TransformerEncoderLayer``s (GELU) and a soft attention pooling; the two clip vectors are concatenated and scored by
LayerNorm -> Linear -> GELU -> Linear.  Both branches are ONE HIP kernel launch (one workgroup per (clip, modality),
activations never leave LDS, weights streamed from L2 in a pre-transposed flat buffer), the head a second tiny kernel:
csrc/af_dual.hip.  The module below only owns the parameters (same ``state_dict`` keys as the reference, 136 of them,
including the auxiliary heads that inference never evaluates) and the launch plumbing - no CPU / eager fallback.
"""
import ctypes as C
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import nn

from .synth import _gen


@dataclass(frozen=True)
class DualSpec:
    au_dim: int = 36
    lmk_dim: int = 132
    d_model: int = 256
    depth: int = 4
    heads: int = 4
    ff: int = 768                  # dim_feedforward = int(d_model * mlp_ratio), mlp_ratio = ff_dim / d_model (run.py:175)
    pool_tau: float = 1.0
    proj_dim: int = 128

    def branches(self) -> List[Tuple[str, int]]:
        return [("au_enc", self.au_dim), ("lmk_enc", self.lmk_dim)]


def dual_state_dict_layout(spec: DualSpec):
    """[(key, shape)] in the reference's ``DualEncoderAU_LMK.state_dict()`` order (use_dat False)."""
    D, F = spec.d_model, spec.ff
    out = []
    for name, din in spec.branches():
        out += [(name + ".proj.weight", (D, din)), (name + ".proj.bias", (D,)),
                (name + ".ln_in.weight", (D,)), (name + ".ln_in.bias", (D,))]
        for i in range(3):
            out += [(name + ".temporal.%d.weight" % i, (D, 1, 3)), (name + ".temporal.%d.bias" % i, (D,))]
        out += [(name + ".pointwise.weight", (D, D, 1)), (name + ".pointwise.bias", (D,))]
        for l in range(spec.depth):
            p = name + ".encoder.layers.%d." % l
            out += [(p + "self_attn.in_proj_weight", (3 * D, D)), (p + "self_attn.in_proj_bias", (3 * D,)),
                    (p + "self_attn.out_proj.weight", (D, D)), (p + "self_attn.out_proj.bias", (D,)),
                    (p + "linear1.weight", (F, D)), (p + "linear1.bias", (F,)),
                    (p + "linear2.weight", (D, F)), (p + "linear2.bias", (D,)),
                    (p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                    (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,))]
        out.append((name + ".pool.v", (D,)))
    out += [("head.0.weight", (2 * D,)), ("head.0.bias", (2 * D,)), ("head.1.weight", (2 * D, 2 * D)),
            ("head.1.bias", (2 * D,)), ("head.4.weight", (1, 2 * D)), ("head.4.bias", (1,)),
            ("au_from_lmk.0.weight", (D,)), ("au_from_lmk.0.bias", (D,)),
            ("au_from_lmk.1.weight", (spec.au_dim, D)), ("au_from_lmk.1.bias", (spec.au_dim,)),
            ("proj_au.weight", (spec.proj_dim, D)), ("proj_au.bias", (spec.proj_dim,)),
            ("proj_lmk.weight", (spec.proj_dim, D)), ("proj_lmk.bias", (spec.proj_dim,))]
    return out


def dual_synthetic_state_dict(spec: DualSpec = None, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded recipe weights in the reference layout (the trained ``best.pt`` files are not shipped: SURVEY 8c):
    matrices N(0, 1/fan_in), LayerNorm gamma 1 + 0.1 N(0,1), every bias 0.05 N(0,1), pooling vector N(0,1)."""
    spec = spec or DualSpec()
    sd = OrderedDict()
    for idx, (key, shape) in enumerate(dual_state_dict_layout(spec)):
        g = _gen(seed + 7000, idx)
        leaf = key.rsplit(".", 1)[-1]
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "v":
            t = torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[key] = t.contiguous()
    return sd


def synthetic_dual_inputs(batch: int, spec: DualSpec = None, frames: int = 8, seed: int = 0):
    """z-scored-looking AU / landmark tracks (the extractors - LibreFace, MediaPipe - are absent offline) and a ragged
    ``lengths`` vector (valid frames per clip; the rest is padding)."""
    spec = spec or DualSpec()
    g = _gen(seed + 9000, 0)
    A = torch.randn((batch, frames, spec.au_dim), generator=g)
    L = torch.randn((batch, frames, spec.lmk_dim), generator=g)
    lengths = torch.full((batch,), frames, dtype=torch.int32)
    for b in range(batch):
        if b % 3 == 2:
            lengths[b] = max(1, frames - 1 - (b % frames) // 2)
    return A, L, lengths


def sinusoid_table(frames: int, d_model: int) -> torch.Tensor:
    """PositionalEncoding.pe[:frames] (dual_encoder.py:16-23)."""
    pe = torch.zeros(frames, d_model)
    pos = torch.arange(0, frames, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class _Pool(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.v = nn.Parameter(torch.randn(d))


class _Branch(nn.Module):
    """Parameter container with the reference BranchEncoder's names (dual_encoder.py:54-71)."""

    def __init__(self, din, spec: DualSpec):
        super().__init__()
        D = spec.d_model
        self.proj = nn.Linear(din, D)
        self.ln_in = nn.LayerNorm(D)
        self.temporal = nn.ModuleList([nn.Conv1d(D, D, 3, padding=d, groups=D, dilation=d) for d in (1, 2, 4)])
        self.pointwise = nn.Conv1d(D, D, 1)
        layer = nn.TransformerEncoderLayer(d_model=D, nhead=spec.heads, dim_feedforward=spec.ff, dropout=0.0,
                                           batch_first=True, activation="gelu", norm_first=True)
        self.encoder = nn.TransformerEncoder(layer, num_layers=spec.depth, enable_nested_tensor=False)
        self.pool = _Pool(D)


class DualEncoderAU_LMK(nn.Module):
    """Same constructor arguments, ``state_dict`` and inference outputs as the reference class; ``forward`` returns
    ``{"bin_logits": (B,), "dom_logits": None[, "z": (B, 2*d_model)]}``.  Training-only outputs (``need_aux``,
    ``return_seq``, the DAT head) are not part of the inference path and raise."""

    MAX_FRAMES = 16

    def __init__(self, au_dim=36, lmk_dim=132, d_model=256, depth=4, heads=4, mlp_ratio=2.0, dropout=0.1, proj_dim=128,
                 use_dat=False, domain_classes=0, pool_tau: float = 1.0):
        super().__init__()
        if use_dat:
            raise NotImplementedError("the domain-adversarial head is training-only")
        if d_model != 256 or d_model % heads:
            raise ValueError("the HIP branch kernel is built for d_model = 256 (one channel per thread)")
        self.spec = DualSpec(au_dim, lmk_dim, d_model, depth, heads, int(d_model * mlp_ratio), float(pool_tau), proj_dim)
        self.au_enc = _Branch(au_dim, self.spec)
        self.lmk_enc = _Branch(lmk_dim, self.spec)
        self.head = nn.Sequential(nn.LayerNorm(2 * d_model), nn.Linear(2 * d_model, 2 * d_model), nn.GELU(), nn.Dropout(0.2),
                                  nn.Linear(2 * d_model, 1))
        self.au_from_lmk = nn.Sequential(nn.LayerNorm(d_model), nn.Linear(d_model, au_dim))
        self.proj_au = nn.Linear(d_model, proj_dim)
        self.proj_lmk = nn.Linear(d_model, proj_dim)
        self.use_dat, self.domain_head = False, None
        self._packed = None                     # (signature, {branch: flat buffer}, head buffer)
        self._pe = {}

    # -- weights -> the kernels' flat fp32 images ---------------------------------------------------------
    def _signature(self):
        first = next(self.parameters())
        return (str(first.device), first.data_ptr(), sum(t._version for t in self.parameters()))

    def _pack(self, device):
        from . import _lib
        from ._lib import check, lib
        sig = self._signature()
        if self._packed is not None and self._packed[0] == sig:
            return self._packed[1], self._packed[2]
        sp = self.spec
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in self.state_dict().items()}
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

        def flat(parts):
            n = sum(rows * cols if tr else t.numel() for t, tr, rows, cols in parts)
            buf = torch.empty(n, dtype=torch.float32, device=device)
            off = 0
            for t, tr, rows, cols in parts:
                if tr:      # (rows, cols) row-major -> (cols, rows): the kernels read W^T so that threads walk output columns
                    check(lib.af_transpose_f32(C.c_void_p(t.data_ptr()), rows, cols, C.c_void_p(buf.data_ptr() + 4 * off), st),
                          "af_transpose_f32")
                    off += rows * cols
                else:
                    buf[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            assert off == n
            return buf

        D, F = sp.d_model, sp.ff
        branches = {}
        for name, din in sp.branches():
            g = lambda k: sd[name + "." + k]
            parts = [(g("proj.weight"), True, D, din), (g("proj.bias"), False, 0, 0),
                     (g("ln_in.weight"), False, 0, 0), (g("ln_in.bias"), False, 0, 0)]
            for i in range(3):
                parts += [(g("temporal.%d.weight" % i), False, 0, 0), (g("temporal.%d.bias" % i), False, 0, 0)]
            parts += [(g("pointwise.weight").reshape(D, D), True, D, D), (g("pointwise.bias"), False, 0, 0)]
            for l in range(sp.depth):
                p = "encoder.layers.%d." % l
                parts += [(g(p + "norm1.weight"), False, 0, 0), (g(p + "norm1.bias"), False, 0, 0),
                          (g(p + "self_attn.in_proj_weight"), True, 3 * D, D), (g(p + "self_attn.in_proj_bias"), False, 0, 0),
                          (g(p + "self_attn.out_proj.weight"), True, D, D), (g(p + "self_attn.out_proj.bias"), False, 0, 0),
                          (g(p + "norm2.weight"), False, 0, 0), (g(p + "norm2.bias"), False, 0, 0),
                          (g(p + "linear1.weight"), True, F, D), (g(p + "linear1.bias"), False, 0, 0),
                          (g(p + "linear2.weight"), True, D, F), (g(p + "linear2.bias"), False, 0, 0)]
            parts.append((g("pool.v"), False, 0, 0))
            buf = flat(parts)
            assert buf.numel() == lib.af_dual_branch_weight_floats(din, D, sp.depth, F), "flat layout out of sync with the kernel"
            branches[name] = buf
        head = flat([(sd["head.0.weight"], False, 0, 0), (sd["head.0.bias"], False, 0, 0),
                     (sd["head.1.weight"], True, 2 * D, 2 * D), (sd["head.1.bias"], False, 0, 0),
                     (sd["head.4.weight"].reshape(-1), False, 0, 0), (sd["head.4.bias"], False, 0, 0)])
        torch.cuda.current_stream(device).synchronize()
        self._packed = (sig, branches, head)
        return branches, head

    def forward(self, A, L, lengths=None, need_aux=False, return_z=False, return_seq=False, dat_lambda: float = 0.0):
        from . import _lib
        from ._lib import check, lib
        if need_aux or return_seq or dat_lambda > 0:
            raise NotImplementedError("need_aux / return_seq / DAT are training-time outputs")
        if not (A.is_cuda and L.is_cuda):
            raise RuntimeError("the MI355X dual encoder only runs on HIP device tensors (no CPU fallback)")
        if self.training:
            raise RuntimeError("inference only: call .eval() first")
        B, T, _ = A.shape
        sp = self.spec
        if L.shape[:2] != (B, T) or A.shape[2] != sp.au_dim or L.shape[2] != sp.lmk_dim:
            raise ValueError("expected A (B,T,%d) and L (B,T,%d)" % (sp.au_dim, sp.lmk_dim))
        if T < 1 or T > self.MAX_FRAMES:
            raise ValueError("1..%d frames per clip (got %d)" % (self.MAX_FRAMES, T))
        dev = A.device
        if B == 0:
            out = {"bin_logits": A.new_zeros((0,), dtype=torch.float32), "dom_logits": None}
            if return_z:
                out["z"] = A.new_zeros((0, 2 * sp.d_model), dtype=torch.float32)
            return out
        with torch.cuda.device(dev):
            branches, head = self._pack(dev)
            if T not in self._pe or self._pe[T].device != dev:
                self._pe[T] = sinusoid_table(T, sp.d_model).to(dev)
            if lengths is not None:
                if lengths.dim() != 1 or lengths.numel() != B:
                    raise ValueError("lengths must be (B,) valid-frame counts")
                lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            z = torch.empty((B, 2 * sp.d_model), dtype=torch.float32, device=dev)
            logits = torch.empty((B,), dtype=torch.float32, device=dev)
            xs = [x.to(torch.float32).contiguous() for x in (A, L)]
            nb = len(xs)
            xp = (C.c_void_p * nb)(*[x.data_ptr() for x in xs])
            wp = (C.c_void_p * nb)(*[branches[name].data_ptr() for name, _ in sp.branches()])
            dins = (C.c_int * nb)(*[din for _, din in sp.branches()])
            check(lib.af_dual_branch_encoders(nb, xp, wp, dins, None if lengths is None else C.c_void_p(lengths.data_ptr()),
                                              C.c_void_p(self._pe[T].data_ptr()), B, T, sp.d_model, sp.depth, sp.heads, sp.ff,
                                              C.c_float(sp.pool_tau), C.c_void_p(z.data_ptr()), 2 * sp.d_model, st),
                  "af_dual_branch_encoders")
            check(lib.af_dual_head(C.c_void_p(z.data_ptr()), C.c_void_p(head.data_ptr()), B, 2 * sp.d_model,
                                   C.c_void_p(logits.data_ptr()), st), "af_dual_head")
        out = {"bin_logits": logits, "dom_logits": None}
        if return_z:
            out["z"] = z
        return out


class GatedMoE(nn.Module):
    """Drop-in for the reference's ``GatedMoE`` (dualrun/rgb/engine_rgb.py:369-384): fuses the RGB (AltFreezing) logit and
    the dual-encoder logit of a clip.  ``forward(z_rgb (B,1), z_dual (B,1)) -> (z (B,1), g (B,1))`` on HIP tensors."""

    def __init__(self, hidden: int = 8):
        super().__init__()
        self.t_rgb = nn.Parameter(torch.tensor(1.0))
        self.t_dual = nn.Parameter(torch.tensor(1.0))
        self.gate = nn.Sequential(nn.Linear(3, hidden), nn.ReLU(), nn.Linear(hidden, 1))
        self.hidden = hidden

    def forward(self, z_rgb: torch.Tensor, z_dual: torch.Tensor):
        from ._lib import check, lib
        if not (z_rgb.is_cuda and z_dual.is_cuda):
            raise RuntimeError("GatedMoE only runs on HIP device tensors (no CPU fallback)")
        if z_rgb.shape != z_dual.shape or z_rgb.dim() != 2 or z_rgb.shape[1] != 1:
            raise ValueError("z_rgb and z_dual must both be (B,1)")
        dev = z_rgb.device
        with torch.cuda.device(dev):
            w = torch.cat([self.t_rgb.reshape(1), self.t_dual.reshape(1), self.gate[0].weight.reshape(-1),
                           self.gate[0].bias, self.gate[2].weight.reshape(-1), self.gate[2].bias]).detach().float().contiguous()
            a, b = z_rgb.detach().float().contiguous(), z_dual.detach().float().contiguous()
            z, g = torch.empty_like(a), torch.empty_like(a)
            check(lib.af_gated_moe(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(w.data_ptr()), self.hidden,
                                   a.numel(), C.c_void_p(z.data_ptr()), C.c_void_p(g.data_ptr()),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "af_gated_moe")
        return z, g

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


def _flat(parts, device):
    """Concatenates (tensor, transpose?, rows, cols) parts into one flat fp32 device buffer; a transposed part (rows, cols)
    row-major becomes (cols, rows): the kernels read W^T so that consecutive threads walk consecutive output columns."""
    from ._lib import check, lib
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    n = sum(rows * cols if tr else t.numel() for t, tr, rows, cols in parts)
    buf = torch.empty(n, dtype=torch.float32, device=device)
    off = 0
    for t, tr, rows, cols in parts:
        if tr:
            check(lib.af_transpose_f32(C.c_void_p(t.data_ptr()), rows, cols, C.c_void_p(buf.data_ptr() + 4 * off), st),
                  "af_transpose_f32")
            off += rows * cols
        else:
            buf[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
    assert off == n
    return buf


def _pack_branch_images(sd, sp: DualSpec, device):
    """{branch name: flat fp32 weight image of af_dual_branch_encoders} from fp32 device tensors keyed like the reference."""
    from ._lib import lib
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
        buf = _flat(parts, device)
        assert buf.numel() == lib.af_dual_branch_weight_floats(din, D, sp.depth, F), "flat layout out of sync with the kernel"
        branches[name] = buf
    return branches


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
        sig = self._signature()
        if self._packed is not None and self._packed[0] == sig:
            return self._packed[1], self._packed[2]
        sp = self.spec
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in self.state_dict().items()}
        D = sp.d_model
        branches = _pack_branch_images(sd, sp, device)
        head = _flat([(sd["head.0.weight"], False, 0, 0), (sd["head.0.bias"], False, 0, 0),
                      (sd["head.1.weight"], True, 2 * D, 2 * D), (sd["head.1.bias"], False, 0, 0),
                      (sd["head.4.weight"].reshape(-1), False, 0, 0), (sd["head.4.bias"], False, 0, 0)], device)
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


def dual_rgb_state_dict_layout(spec: DualSpec, vis_dim: int):
    """[(key, shape)] in the reference's ``DualEncoderRGB.state_dict()`` order (dualrun/model/dual_rgb.py:53-84)."""
    D = spec.d_model
    branch_keys = [(k, s) for k, s in dual_state_dict_layout(spec) if k.startswith(("au_enc.", "lmk_enc."))]
    return branch_keys + [("rgb_proj.weight", (D, vis_dim)),
                          ("head.0.weight", (3 * D,)), ("head.0.bias", (3 * D,)), ("head.1.weight", (2 * D, 3 * D)),
                          ("head.1.bias", (2 * D,)), ("head.4.weight", (1, 2 * D)), ("head.4.bias", (1,))]


def dual_rgb_synthetic_state_dict(spec: DualSpec, vis_dim: int, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """The W(seed) recipe of ``dual_synthetic_state_dict`` on the tri-modal layout."""
    sd = OrderedDict()
    for idx, (key, shape) in enumerate(dual_rgb_state_dict_layout(spec, vis_dim)):
        g = _gen(seed + 7100, idx)
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


class _Identity(nn.Module):
    """``AltFreezingRGBEncoder`` with ``from_features=True`` holds no parameters (dual_rgb.py:16-24)."""


class DualEncoderRGB(nn.Module):
    """Drop-in for the reference's tri-modal ``DualEncoderRGB`` (dualrun/model/dual_rgb.py:47-122): the AU / landmark branch
    encoders + a frozen RGB stream = the masked mean of AltFreezing features ``V`` (``AltFreezingRGBEncoder`` with
    ``from_features=True``, :27-44: the backbone's pooled 2048-vector per clip window, feature.py:105-114) projected by
    ``rgb_proj`` (Linear without bias), concatenated and scored by LayerNorm(3d) -> Linear(3d, 2d) -> GELU -> Linear(2d, 1).
    Same constructor arguments and ``state_dict`` (112 keys) as the reference; ``forward(A, L, V, key_padding_mask)``
    returns the (B,) logits like the reference (``return_weights`` / ``return_seq`` are training-time outputs and raise).

    Two things are mirrored AS WRITTEN upstream: the branch encoders get ``ff_dim`` in ``BranchEncoder``'s ``mlp_ratio``
    slot (dual_rgb.py:59-60), so ``dim_feedforward = int(d_model * ff_dim)`` - pass ``ff_dim=3.0`` for the 768-wide layers
    of the shipped ``args.json`` (the default 768 would mean 196 608-wide layers; the HIP kernel holds <= 768); and the
    branches pool with ``BranchEncoder``'s default ``pool_tau = 0.7``.  ``key_padding_mask`` (True = padding) must be a
    prefix mask (what ``lengths_to_mask`` produces, :87-89): the kernels take the count of valid frames.
    With ``rgb_backbone`` = an ``I3D8x8`` / ``Classifier`` of this package and ``rgb_from_features=False``, ``V`` is a uint8
    clip batch (B,T,H,W,3) and the pooled feature comes from the AltFreezing engine (one window per sample)."""

    MAX_FRAMES = 16

    def __init__(self, au_dim: int, lmk_dim: int, vis_dim: int, d_model: int = 256, depth: int = 4, heads: int = 4,
                 ff_dim: float = 768, dropout: float = 0.1, rgb_backbone: Optional[nn.Module] = None,
                 rgb_from_features: bool = True):
        super().__init__()
        ff = int(d_model * ff_dim)
        if ff > 768:
            raise ValueError("dim_feedforward = int(d_model * ff_dim) = %d: upstream passes ff_dim into BranchEncoder's mlp_ratio "
                             "slot (dual_rgb.py:59-60); the HIP branch kernel holds <= 768 - pass ff_dim = 3.0 for the shipped "
                             "checkpoints' 768-wide layers" % ff)
        if d_model != 256 or d_model % heads:
            raise ValueError("the HIP branch kernel is built for d_model = 256 (one channel per thread)")
        self.spec = DualSpec(au_dim, lmk_dim, d_model, depth, heads, ff, 0.7, 128)     # BranchEncoder's default pool_tau
        self.au_enc = _Branch(au_dim, self.spec)
        self.lmk_enc = _Branch(lmk_dim, self.spec)
        self.vis_dim = int(vis_dim)
        self.rgb = _Identity()                                        # AltFreezingRGBEncoder: no parameters of its own here
        self.rgb_backbone = [rgb_backbone]                            # not a submodule: frozen and owned by the caller
        self.rgb_from_features = bool(rgb_from_features)
        self.rgb_proj = nn.Linear(vis_dim, d_model, bias=False)
        self.head = nn.Sequential(nn.LayerNorm(3 * d_model), nn.Linear(3 * d_model, 2 * d_model), nn.GELU(), nn.Dropout(dropout),
                                  nn.Linear(2 * d_model, 1))
        self.use_dat, self.domain_head, self.quality_head = False, None, None
        self._packed = None
        self._pe = {}
        self._mask_checked = None              # (tensor, version) of the last key_padding_mask that passed validation (a strong reference)
        self._side = None                      # side stream of the two-stream form (rgb_from_features=False)

    def _signature(self):
        first = next(self.parameters())
        return (str(first.device), first.data_ptr(), sum(t._version for t in self.parameters()))

    @staticmethod
    def lengths_to_mask(t_valid: torch.Tensor, T: int, device):
        """True = padding; a clip without a valid frame keeps frame 0 (dual_encoder.py:137-156, dual_rgb.py:87-89)."""
        pad = torch.arange(T, device=device).expand(t_valid.numel(), T) >= t_valid.to(device).view(-1, 1)
        allp = pad.all(dim=1)
        pad[allp, 0] = False
        return pad

    def _pack(self, device):
        """(branch images, flat head image for af_mlp_head, rgb_proj^T [vis][d]); cached until the parameters change"""
        from ._lib import check, lib
        sig = self._signature()
        if self._packed is not None and self._packed[0] == sig:
            return self._packed[1:]
        sp, D = self.spec, self.spec.d_model
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in self.state_dict().items()}
        branches = _pack_branch_images(sd, sp, device)
        head = _flat([(sd["head.0.weight"], False, 0, 0), (sd["head.0.bias"], False, 0, 0),
                      (sd["head.1.weight"], True, 2 * D, 3 * D), (sd["head.1.bias"], False, 0, 0),
                      (sd["head.4.weight"].reshape(-1), False, 0, 0), (sd["head.4.bias"], False, 0, 0)], device)
        wpt = _flat([(sd["rgb_proj.weight"], True, D, self.vis_dim)], device)
        torch.cuda.current_stream(device).synchronize()
        self._packed = (sig, branches, head, wpt)
        return branches, head, wpt

    def forward(self, A, L, V=None, key_padding_mask=None, return_weights: bool = False, return_seq: bool = False,
                return_scores: bool = False, return_rgb: bool = False):
        from ._lib import check, lib
        if return_weights or return_seq:
            raise NotImplementedError("return_weights / return_seq are training-time outputs")
        if V is None:
            raise ValueError("DualEncoderRGB needs V: (B,T,vis_dim) features, or uint8 clips with rgb_from_features=False")
        if not (A.is_cuda and L.is_cuda and V.is_cuda):
            raise RuntimeError("the MI355X dual encoder only runs on HIP device tensors (no CPU fallback)")
        if self.training:
            raise RuntimeError("inference only: call .eval() first")
        B, T, _ = A.shape
        sp = self.spec
        dev = A.device
        if L.shape[:2] != (B, T) or A.shape[2] != sp.au_dim or L.shape[2] != sp.lmk_dim:
            raise ValueError("expected A (B,T,%d) and L (B,T,%d)" % (sp.au_dim, sp.lmk_dim))
        if T < 1 or T > self.MAX_FRAMES:
            raise ValueError("1..%d frames per clip (got %d)" % (self.MAX_FRAMES, T))
        lengths = None
        if key_padding_mask is not None:
            if key_padding_mask.shape != (B, T) or key_padding_mask.dtype != torch.bool:
                raise ValueError("key_padding_mask must be a (B,T) bool tensor (True = padding)")
            valid = ~key_padding_mask.to(dev)
            lengths = valid.sum(dim=1).to(torch.int32)
            # the two checks read a flag back from the device, i.e. wait for everything enqueued before (a whole AltFreezing forward
            # in the two-stream model): the mask TENSOR that passed them is not checked again until it is modified.  The module keeps
            # a strong reference to it and compares identity + version counter: an address or a shape identifies no contents (a new
            # mask built per step lands at the freed address of the last one with the same version), an object that is still alive
            # and unmodified does.  (An inference-mode tensor has no version counter: always checked.)
            seen = self._mask_checked
            if (key_padding_mask.is_inference() or seen is None or seen[0] is not key_padding_mask
                    or seen[1] != key_padding_mask._version):
                prefix = torch.arange(T, device=dev).expand(B, T) < lengths.view(-1, 1)
                if not bool((prefix == valid).all()):
                    raise ValueError("key_padding_mask must mark a suffix of every clip as padding (lengths_to_mask form)")
                if bool((lengths == 0).any()):
                    raise ValueError("a clip without any valid frame: upstream's softmax over an all-masked row is NaN; "
                                     "build the mask with lengths_to_mask, which keeps frame 0")
                self._mask_checked = None if key_padding_mask.is_inference() else (key_padding_mask, key_padding_mask._version)
            lengths = lengths.contiguous()
        clips = None
        if not self.rgb_from_features:
            bb = self.rgb_backbone[0]
            if bb is None or V.dtype != torch.uint8 or V.dim() != 5 or V.shape[0] != B:
                raise ValueError("rgb_from_features=False needs rgb_backbone (an af_mi355x I3D8x8 / Classifier) and uint8 clips (B,T,H,W,3)")
            clips, V = V, None
        elif V.dim() != 3 or V.shape[0] != B or V.shape[2] != self.vis_dim or V.shape[1] not in (1, T):
            raise ValueError("V must be (B, T or 1, %d) features" % self.vis_dim)
        if return_rgb and clips is None:
            raise ValueError("return_rgb needs rgb_from_features=False (the module then runs the RGB backbone itself)")
        if B == 0:
            return A.new_zeros((0,), dtype=torch.float32)
        rgb = None
        with torch.cuda.device(dev):
            branches, head, wpt = self._pack(dev)
            if T not in self._pe or self._pe[T].device != dev:
                self._pe[T] = sinusoid_table(T, sp.d_model).to(dev)
            cur = torch.cuda.current_stream(dev)
            D = sp.d_model
            z = torch.empty((B, 3 * D), dtype=torch.float32, device=dev)
            logits = torch.empty((B,), dtype=torch.float32, device=dev)
            scores = torch.empty((B,), dtype=torch.float32, device=dev) if return_scores else None
            xs = [x.to(torch.float32).contiguous() for x in (A, L)]
            xp = (C.c_void_p * 2)(*[x.data_ptr() for x in xs])
            wp = (C.c_void_p * 2)(*[branches[name].data_ptr() for name, _ in sp.branches()])
            dins = (C.c_int * 2)(*[din for _, din in sp.branches()])
            lp = None if lengths is None else C.c_void_p(lengths.data_ptr())

            def branch_encoders(stream):
                check(lib.af_dual_branch_encoders(2, xp, wp, dins, lp, C.c_void_p(self._pe[T].data_ptr()), B, T, D, sp.depth, sp.heads,
                                                  sp.ff, C.c_float(sp.pool_tau), C.c_void_p(z.data_ptr()), 3 * D,
                                                  C.c_void_p(stream.cuda_stream)), "af_dual_branch_encoders")

            if clips is None:
                branch_encoders(cur)
            else:
                # the AU / landmark encoders do not depend on the RGB stream: their 2 x B workgroups (0.5 ms on 2 x B of the 256 CUs)
                # run on a side stream beside the backbone's forward and meet it in front of the projection of the pooled feature
                if self._side is None or self._side.device != dev:
                    self._side = torch.cuda.Stream(dev)
                side = self._side
                side.wait_stream(cur)
                for t_ in xs + [z] + ([lengths] if lengths is not None else []):
                    t_.record_stream(side)
                branch_encoders(side)
                net = getattr(self.rgb_backbone[0], "network", self.rgb_backbone[0])
                rgb = net.forward_clips_u8(clips, return_pooled=True)
                V = rgb["pooled"].view(B, 1, -1)                                      # one window per sample
                cur.wait_stream(side)
            st = C.c_void_p(cur.cuda_stream)
            Vf = V.to(torch.float32).contiguous()
            check(lib.af_masked_mean_proj(C.c_void_p(Vf.data_ptr()), B, Vf.shape[1], self.vis_dim, lp, T, C.c_void_p(wpt.data_ptr()), D,
                                          C.c_void_p(z.data_ptr() + 4 * 2 * D), 3 * D, st), "af_masked_mean_proj")
            check(lib.af_mlp_head(C.c_void_p(z.data_ptr()), C.c_void_p(head.data_ptr()), B, 3 * D, 2 * D, C.c_void_p(logits.data_ptr()),
                                  None if scores is None else C.c_void_p(scores.data_ptr()), st), "af_mlp_head")
        out = (logits, scores) if return_scores else logits
        return (out, rgb) if return_rgb else out


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

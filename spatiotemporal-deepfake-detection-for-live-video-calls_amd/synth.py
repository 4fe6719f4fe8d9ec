"""Seeded synthetic checkpoints and clips.

The trained ``altfreezing/checkpoints/model.pth`` is not shipped with the reference
 and there is no network, so parity and throughput are
measured on *recipe* weights W(seed) laid out exactly like the reference's
``network.state_dict()`` (320 tensors).  The recipe is deterministic on the CPU
generator of a given torch build (same image here and on the GPU box), is
independent of generation order (one generator per key) and deliberately does
NOT zero the last BN gamma of each block (the reference's stock init does,
``ZERO_INIT_FINAL_BN``; that would turn every residual branch into a no-op and
make the parity tests vacuous).
"""
import hashlib
import math
from collections import OrderedDict

import numpy as np
import torch

from .arch import IMAGENET_MEAN, IMAGENET_STD, NetSpec, i3d_r50_spec, state_dict_layout


def _gen(seed: int, idx: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + int(idx) * 7919 + 12345) & 0x7FFFFFFFFFFF)
    return g


def fill_layout(layout, seed: int, final_bn=(), linear=(), recipe: str = "mild") -> "OrderedDict[str, torch.Tensor]":
    """Applies the W(seed) recipe to any [(key, shape, dtype_name)] layout.
    ``final_bn``: BN prefixes that get the small-gamma rule; ``linear``: nn.Linear prefixes.
    ``recipe``: "mild" (every block's last BN gamma ~ U(0, 0.4), head N(0, 0.05): logits O(0.3), nearly input
    independent) or "hot" (last BN gamma ~ U(0.3, 0.9), head N(0, 0.1): the residual branches dominate, logits are
    O(10..40) and move by O(1..10) between clips - a localized kernel bug shows in the logit)."""
    if recipe not in ("mild", "hot"):
        raise ValueError(recipe)
    hot = recipe == "hot"
    final_bn, linear = set(final_bn), set(linear)
    sd = OrderedDict()
    for idx, (key, shape, dtype) in enumerate(layout):
        g = _gen(seed, idx)
        shape = tuple(shape)
        prefix, leaf = key.rsplit(".", 1)
        if dtype == "int64":
            t = torch.zeros(shape, dtype=torch.int64)
        elif len(shape) == 5:                                  # conv weight, c2-MSRA style
            fan_out = shape[0] * shape[2] * shape[3] * shape[4]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif prefix in linear:
            # transformer-head linears: the reference's own trunc_normal std (time_transformer.py:262-266)
            t = torch.randn(shape, generator=g) * (0.02 if ".transformer." in key else 0.1 if hot else 0.05)
        elif leaf in ("pos_embedding", "cls_token"):           # transformer head parameters: randn like the reference
            t = torch.randn(shape, generator=g)
        elif leaf == "weight":                                 # BN gamma
            if prefix in final_bn:
                t = torch.rand(shape, generator=g) * 0.6 + 0.3 if hot else torch.rand(shape, generator=g) * 0.4
            else:
                t = torch.rand(shape, generator=g) + 0.5
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def synthetic_state_dict(spec: NetSpec = None, seed: int = 0, recipe: str = "mild") -> "OrderedDict[str, torch.Tensor]":
    """W(seed): fp32 CPU tensors keyed/ordered like the reference state_dict."""
    spec = spec or i3d_r50_spec()
    linear = spec.linear_prefixes() if hasattr(spec, "linear_prefixes") else [spec.head]
    return fill_layout(state_dict_layout(spec), seed,
                       final_bn=[cv.bn_key for cv in spec.convs() if cv.final_bn], linear=linear, recipe=recipe)


def synthetic_tensor(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    """Seeded N(0, scale) activation tensor for per-layer known-answer tests."""
    return torch.randn(tuple(shape), generator=_gen(seed, 424242)) * scale


def state_dict_sha256(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def tensor_sha256(t: torch.Tensor) -> str:
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def synthetic_clips_u8(batch: int, seed: int = 2026, kind: str = "uniform",
                       num_frames: int = 32, size: int = 224) -> torch.Tensor:
    """Caller-layout face-crop clips: uint8 (B, T, H, W, 3), RGB, 0..255
    (the layout handed to ``ClassifierSvc.infer_scores``, reference test/af_realtime.py:75-77).

    kind="uniform": i.i.d. U{0..255} pixels (the BASELINE.md synthetic input).
    kind="smooth" : low-frequency colour field + mild noise (image-like spatial structure,
                    so that a spatially mis-indexed kernel cannot hide behind i.i.d. data).
    """
    clips = []
    for b in range(batch):
        g = _gen(seed, 100000 + b)
        if kind == "uniform":
            c = torch.randint(0, 256, (num_frames, size, size, 3), generator=g, dtype=torch.int32)
        elif kind == "smooth":
            low = torch.rand((1, 3, max(num_frames // 4, 2), 7, 7), generator=g) * 255.0
            up = torch.nn.functional.interpolate(low, size=(num_frames, size, size),
                                                 mode="trilinear", align_corners=True)
            noise = torch.randn((num_frames, size, size, 3), generator=g) * 6.0
            c = (up[0].permute(1, 2, 3, 0) + noise).round().clamp_(0, 255).to(torch.int32)
        else:
            raise ValueError(kind)
        clips.append(c.to(torch.uint8))
    return torch.stack(clips, 0)


def normalize_like_callers(clips_bthwc: torch.Tensor) -> torch.Tensor:
    """(B,T,H,W,C) 0..255 -> logical (B,C,T,H,W) fp32, physically channels-last,
    exactly as the callers do it: as_tensor(float32).permute(0,4,1,2,3).sub(mean).div(std)
    (reference test/af_realtime.py:77-83)."""
    x = torch.as_tensor(clips_bthwc, dtype=torch.float32).permute(0, 4, 1, 2, 3)
    mean, std = pixel_mean_std_f32(x.device)
    return x.sub(mean.view(1, 3, 1, 1, 1)).div(std.view(1, 3, 1, 1, 1))


def pixel_mean_std_f32(device="cpu"):
    """fp32 (mean, std) on the 0..255 scale, rounded the way the callers round them:
    ``torch.tensor([0.485, 0.456, 0.406]) * 255`` evaluated in float32."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=device) * 255
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=device) * 255
    return mean, std

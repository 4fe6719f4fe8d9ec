"""Drop-in for the reference classifier plugin ``model.classifier.i3d_ori``.

Surface mirrored (same names, argument meaning, return values and error behaviour):

* ``Classifier()`` -> ``.to(device)`` / ``.cuda()`` -> ``.eval()`` -> ``.load(ckpt)`` -> ``__call__(x)``
  (reference altfreezing/model/_base.py:17-104, callers demo.py:403-404, test/af_realtime.py:68-69)
* ``network`` / ``_warped_network`` attributes, ``network.state_dict()`` = the 320-key checkpoint layout
  (``resnet.s1.pathway0_stem.conv.weight`` ... ``resnet.head.projection.bias``)
* ``forward(images, noise=None, has_mask=None, freeze_backbone=False, return_feature_maps=False)``
  -> ``{"final_output": (B,1) fp32 logits}``  (altfreezing/model/classifier/i3d_ori.py:92-104)
* the last ``nn.Linear`` in ``.modules()`` is the head projection and sees the pooled
  ``(B,1,1,1,2048)`` feature whenever somebody hooks it (altfreezing/feature.py:105-114)

The parameters live in an ordinary ``nn.Module`` tree (so ``state_dict``/``load_state_dict``/``.to``
behave exactly like the reference's), but no torch op computes the forward: it is executed by the
HIP engine (engine.py -> libafhip.so).  There is no CPU or eager fallback - calling the model on a
non-HIP tensor raises.
"""
import logging
import math
import os
import traceback
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from .arch import ConvSpec, NetSpec, ftcn_tt_spec, i3d_r50_spec, slowfast_r50_spec

logger = logging.getLogger("af_mi355x")

PRECISIONS = ("auto", "f32", "bf16", "f16")


class _Node(nn.Module):
    """Parameter container; only there to reproduce the reference's module/parameter names."""

    def forward(self, *a, **k):            # pragma: no cover
        raise RuntimeError("skeleton module: the forward is executed by the HIP engine, not by torch.nn")


def _conv_module(cv: ConvSpec) -> nn.Conv3d:
    return nn.Conv3d(cv.cin, cv.cout, cv.kernel, stride=cv.stride, padding=cv.pad, bias=False)


def _bn_module(cv: ConvSpec) -> nn.BatchNorm3d:
    bn = nn.BatchNorm3d(cv.cout, eps=1e-5, momentum=0.1)
    if cv.final_bn:
        bn.transform_final_bn = True
    return bn


def _attach(root: nn.Module, dotted: str, module: nn.Module):
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        if not hasattr(cur, p):
            cur.add_module(p, _Node())
        cur = getattr(cur, p)
    cur.add_module(parts[-1], module)


def _init_like_reference(root: nn.Module, fc_std: float = 0.01, zero_init_final_bn: bool = True):
    """Fresh-model init as the reference does it (slowfast/utils/weight_init_helper.py:10-43): conv
    c2-MSRA, BN gamma 1 (0 on each block's last BN), beta 0, fc N(0, 0.01).  Only matters until load()."""
    for m in root.modules():
        if isinstance(m, nn.Conv3d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm3d):
            final = getattr(m, "transform_final_bn", False) and zero_init_final_bn
            m.weight.data.fill_(0.0 if final else 1.0)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=fc_std)
            if m.bias is not None:
                m.bias.data.zero_()


class _HipNetwork(nn.Module):
    """Parameter skeleton (reference names / shapes, so state_dict / load_state_dict / .to behave as upstream)
    + the HIP engine cache.  Subclasses define ``forward``."""

    def __init__(self, spec, precision: str) -> None:
        super().__init__()
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self.spec, self.precision = spec, precision
        self.resnet = _Node()
        for cv in self.spec.convs():
            _attach(self, cv.conv, _conv_module(cv))
            _attach(self, cv.bn_key, _bn_module(cv))
        self._build_head()
        _init_like_reference(self)
        self._packed: Dict[str, Tuple[tuple, object]] = {}      # dtype -> (signature, PackedWeights)
        # (dtype, batch, dims) -> Engine, least recently used first.  An engine owns a full activation set (about 3 GB
        # at B=16 in bf16), so a live-call service whose face count changes per step must not pin one per batch size
        # forever: at most ``max_engines`` stay alive (288 GB of HBM make 8 a comfortable default).
        self._engines: "OrderedDict[tuple, object]" = OrderedDict()
        self.max_engines = 8

    def _build_head(self):
        _attach(self, "resnet.head.dropout", nn.Dropout(0.5))
        _attach(self, self.spec.head, nn.Linear(self.spec.head_in, self.spec.num_classes, bias=True))

    def _head_linear(self) -> nn.Linear:
        return self.resnet.head.projection

    # -- weight / engine caches ---------------------------------------------------------------------
    def invalidate_packed(self):
        """Drops the packed device weights and every engine; the next forward repacks from the current parameters.
        Needed after edits the signature cannot see: in-place writes through ``.data`` (``p.data.copy_(...)``) do
        not bump a tensor's version counter.  ``load()`` / ``load_state_dict()`` / ``.to()`` are detected."""
        self._packed.clear()
        self._engines.clear()

    def _signature(self):
        first = next(self.parameters())
        ver = 0
        for t in list(self.parameters()) + list(self.buffers()):
            ver += t._version
        return (str(first.device), first.data_ptr(), ver)

    def _select_dtype(self) -> str:
        if self.precision != "auto":
            return self.precision
        if torch.is_autocast_enabled():              # callers' amp=True (af_realtime.py:70,84)
            return "bf16" if torch.get_autocast_dtype("cuda") == torch.bfloat16 else "f16"
        return "f32"

    def _engine(self, dtype: str, batch: int, dims, device, slot: int = 0):
        from .engine import Engine, PackedWeights            # imports libafhip.so: fails loudly if absent
        sig = self._signature()
        cached = self._packed.get(dtype)
        if cached is None or cached[0] != sig:
            state = {k: v for k, v in self.state_dict().items()}
            self._packed[dtype] = (sig, PackedWeights(self.spec, state, dtype, device))
            for k in [k for k in self._engines if k[0] == dtype]:
                del self._engines[k]
        key = (dtype, batch, tuple(dims)) if slot == 0 else (dtype, batch, tuple(dims), slot)
        if key in self._engines:
            self._engines.move_to_end(key)
        else:
            while len(self._engines) >= max(1, self.max_engines):
                self._engines.popitem(last=False)            # least recently used; its buffers go back to the allocator
            self._engines[key] = Engine(self.spec, self._packed[dtype][1], batch, device, dims)
        return self._engines[key]

    def _check_input(self, images):
        if not isinstance(images, torch.Tensor) or images.dim() != 5 or images.size(1) != 3:
            raise ValueError("images must be a (B,3,T,H,W) tensor")
        if not images.is_cuda:
            raise RuntimeError("the MI355X classifier only runs on a HIP device tensor (no CPU fallback); "
                               "got a tensor on %s" % images.device)
        if self.training:
            raise RuntimeError("inference only: call .eval() first (BatchNorm running statistics are folded)")
        if next(self.parameters()).device != images.device:
            raise RuntimeError("model parameters are on %s but the input is on %s"
                               % (next(self.parameters()).device, images.device))
        return images if images.dtype == torch.float32 else images.float()

    def _finish(self, eng, logits, pooled, B):
        proj = self._head_linear()
        if proj._forward_hooks or proj._forward_pre_hooks:
            # somebody (feature.py:105-114) listens on the head Linear: feed it the pooled feature in the
            # reference's (N,T',H',W',C) layout so the hook sees the same input/output as upstream
            feat = pooled.view((B,) + tuple(eng.head_dims) + (pooled.shape[-1],))
            if not hasattr(self.resnet.head, "projection"):          # FTCN-TT: mlp_head's Linear sees (B, dim)
                feat = pooled.view(B, pooled.shape[-1])
            return proj(feat.clone()).reshape(B, -1)
        return logits.clone().view(B, -1)


class I3D8x8(_HipNetwork):
    """The plugin's network module (``module_to_build`` of the reference Classifier)."""

    def __init__(self, clip_size: int = 32, imsize: int = 224, precision: str = "auto", crop_size: int = 224,
                 streams: Optional[int] = None) -> None:
        # the head pool is sized from DATA.CROP_SIZE=224 (defaults.py:277), not from imsize;
        # crop_size is only changed by tests that run a shrunken network
        super().__init__(i3d_r50_spec(num_frames=clip_size, crop=crop_size), precision)
        self.clip_size, self.imsize = clip_size, imsize
        # streams = 2: a batch of >= split_min_batch clips runs as two half-batches on two HIP streams (two engines).  The forward
        # alternates between MFMA-bound and HBM-bound launches; the launches of two independent half-batches can fill some of each
        # other's idle time.  Measured with the round-2 kernels: B=16 +2-3 %, B=32 +5 %; with the round-3 / round-4 kernels (whole
        # rounds of persistent workgroups per launch) it is a LOSS at B=16 (driver's round-3 line: 3 047 against 3 121 clips/s) -
        # DESIGN.md 7.  Off by default (1): opt in with the argument or AF_MI355X_STREAMS=2, and measure.
        self.streams = int(streams if streams is not None else os.environ.get("AF_MI355X_STREAMS", "1"))
        self.split_min_batch = 16
        self._side = {}

    def _run(self, B, dims, dev, runner):
        """runner(engine, lo, hi) -> (logits, pooled) for clips [lo, hi).  Returns (an engine of the run, logits, pooled, scores)."""
        dtype = self._select_dtype()
        ns = self.streams if (self.streams > 1 and B >= self.split_min_batch) else 1
        if ns == 1:
            eng = self._engine(dtype, B, dims, dev)
            logits, pooled = runner(eng, 0, B)
            return eng, logits, pooled, eng.scores
        cur = torch.cuda.current_stream(dev)
        if (str(dev), ns) not in self._side:
            self._side[(str(dev), ns)] = [torch.cuda.Stream(dev) for _ in range(ns)]
        side = self._side[(str(dev), ns)]
        parts = []
        for slot in range(ns):
            lo, hi = slot * B // ns, (slot + 1) * B // ns
            side[slot].wait_stream(cur)                       # the caller's input is ready
            with torch.cuda.stream(side[slot]):
                eng = self._engine(dtype, hi - lo, dims, dev, slot=slot)
                logits, pooled = runner(eng, lo, hi)
                parts.append((eng, logits, pooled, eng.scores))
        for s in side:
            cur.wait_stream(s)
        cat = lambda i: None if parts[0][i] is None else torch.cat([p[i] for p in parts])
        return parts[0][0], cat(1), cat(2), cat(3)

    def forward(self, images, noise=None, has_mask=None, freeze_backbone=False, return_feature_maps=False, return_scores=False):
        assert not freeze_backbone
        x = self._check_input(images)
        B, _, T, H, W = x.shape
        if B == 0:                                   # an empty batch is an empty answer (no launch)
            return {"final_output": x.new_zeros((0, self.spec.num_classes))}
        with torch.cuda.device(x.device):
            eng, logits, pooled, scores = self._run(B, (T, H, W), x.device, lambda e, lo, hi: e.run_f32(x[lo:hi]))
            out = {"final_output": self._finish(eng, logits, pooled, B)}
            if return_scores:                        # not a reference argument: the callers' sigmoid, from the head kernel
                out["scores"] = self._scores_of(eng, scores, B)
        return out

    @staticmethod
    def _scores_of(eng, scores, B):
        if scores is None or eng.head_positions != 1:
            raise ValueError("scores are defined for a 1- or 2-class head on a crop with one head position")
        return scores.clone().view(B)

    def forward_clips_u8(self, clips_bthwc: torch.Tensor, mean=None, std=None, return_scores=False, return_pooled=False):
        """Fused caller prologue: uint8 (B,T,H,W,3) 0..255 RGB clips straight from the aligner; replaces
        as_tensor/permute/sub/div of ``ClassifierSvc.infer_scores`` (test/af_realtime.py:77-83).
        ``return_scores`` adds ``"scores"``: (B,) sigmoid(logit) - or softmax[:,1] for a 2-class head - computed by the
        head kernel itself (the callers' epilogue, af_realtime.py:88-95); ``return_pooled`` adds ``"pooled"``: the
        (B, 2048) average-pooled feature that the head's Linear consumes (what feature.py:105-114 extracts with a hook
        and dualrun's RGB stream consumes, dualrun/model/dual_rgb.py:27-44)."""
        from .synth import pixel_mean_std_f32
        if mean is None or std is None:
            m, s = pixel_mean_std_f32()
            mean, std = m.tolist(), s.tolist()
        dev = clips_bthwc.device
        B, T, H, W, _ = clips_bthwc.shape
        with torch.cuda.device(dev):
            clips = clips_bthwc.contiguous()
            eng, logits, pooled, scores = self._run(B, (T, H, W), dev, lambda e, lo, hi: e.run_u8(clips[lo:hi], mean, std))
            out = {"final_output": self._finish(eng, logits, pooled, B)}
            if return_scores:
                out["scores"] = self._scores_of(eng, scores, B)
            if return_pooled:
                out["pooled"] = pooled.clone().view(B, -1)
        return out

    def infer_scores(self, aligned_batch_bthwc, as_numpy: bool = True):
        """``ClassifierSvc.infer_scores`` (test/af_realtime.py:75-96; = TEST2.py:151-204) as ONE op list: (B,T,H,W,C) RGB
        0..255 clips (uint8 / float, numpy or tensor, any device) -> (B,) fake probabilities.  Normalisation, forward and
        sigmoid all run on the GPU; only the B floats come back."""
        x = torch.as_tensor(aligned_batch_bthwc)
        dev = next(self.parameters()).device
        if x.dtype == torch.uint8:                   # what the aligner hands over: normalisation fused into the input pack
            s = self.forward_clips_u8(x.to(dev), return_scores=True)["scores"]
        else:                                        # real-valued pixels: normalise like the callers (af_realtime.py:77-83)
            from .synth import normalize_like_callers
            s = self.forward(normalize_like_callers(x.to(dev)), return_scores=True)["scores"]
        return s.float().cpu().numpy() if as_numpy else s


class LiveScorer:
    """One tracked face of a live call: ``score = scorer(aligned_clip_u8)`` = ``ClassifierSvc.infer_scores`` for a single window
    (test/af_realtime.py:75-96), with the ~50 launches of the B = 1 forward recorded ONCE into a HIP graph and replayed per window:
    after a second of idle between windows the host side runs cold and issuing those launches one by one (0.5 - 0.9 ms) was on the
    critical path of enqueue -> score; a replay is one call.  ``scorer.clip`` is the static (1, T, H, W, 3) uint8 input - the
    aligner can write the window straight into it (``StreamingCropAligner.align_last(n, out=scorer.clip[0])``) - and the returned
    scores are those of ``infer_scores`` bit for bit (same kernels, same order)."""

    def __init__(self, network: "I3D8x8", clip_size: int = 32, crop: int = 224):
        self.network = network
        dev = next(network.parameters()).device
        self.clip = torch.zeros((1, clip_size, crop, crop, 3), dtype=torch.uint8, device=dev)
        with torch.inference_mode(), torch.cuda.device(dev):
            for _ in range(2):                                          # engine construction, weight packing, kernel attributes
                network.forward_clips_u8(self.clip, return_scores=True)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._scores = network.forward_clips_u8(self.clip, return_scores=True)["scores"]
        self._host = torch.empty(self._scores.shape, dtype=torch.float32, pin_memory=True)

    def __call__(self, aligned_clip_u8: Optional[torch.Tensor] = None, wait_for=None):
        """aligned_clip_u8: (T,H,W,3) or (1,T,H,W,3) uint8 device tensor, or None when the window was written into ``self.clip``.
        The replay runs on the CURRENT stream of the clip's device: whatever wrote ``self.clip`` (e.g.
        ``StreamingCropAligner.align_last(out=scorer.clip[0])``) must have been enqueued on that stream, or be handed over as
        ``wait_for`` (a ``torch.cuda.Event`` or ``torch.cuda.Stream`` the replay then waits on)."""
        dev = self.clip.device
        with torch.cuda.device(dev):                  # (a multi-GPU process: the caller's current device may be another one)
            cur = torch.cuda.current_stream(dev)
            if isinstance(wait_for, torch.cuda.Event):
                cur.wait_event(wait_for)
            elif isinstance(wait_for, torch.cuda.Stream):
                cur.wait_stream(wait_for)
            elif wait_for is not None:
                raise TypeError("wait_for: a torch.cuda.Event or torch.cuda.Stream")
            if aligned_clip_u8 is not None:
                self.clip.copy_(aligned_clip_u8.reshape(self.clip.shape), non_blocking=True)
            self.graph.replay()
            self._host.copy_(self._scores.float(), non_blocking=True)
            cur.synchronize()
        return self._host.numpy().copy()


class SlowFast8x8(_HipNetwork):
    """Two-pathway SlowFast-R50 (reference slowfast/models/video_model_builder.py:146-387) on the same kernels:
    the Fast->Slow laterals (FuseFastToSlow, :86-143) are strided temporal convs that write their channels straight
    behind the Slow pathway's in the same NDHWC rows (no concat copy).  The reference ships no plugin that builds
    it; the parameter names are those of its ``SlowFast`` module under a ``resnet.`` prefix like ``I3D8x8``.

    ``forward(inputs)``: ``inputs = [slow, fast]`` exactly like ``SlowFast.forward`` (slow = T/alpha frames, fast = T
    frames, both (B,3,t,H,W)); a single (B,3,T,H,W) clip is also accepted and read with a frame stride of alpha for
    the Slow pathway."""

    def __init__(self, clip_size: int = 32, precision: str = "auto", crop_size: int = 224, alpha: int = 8) -> None:
        super().__init__(slowfast_r50_spec(num_frames=clip_size, crop=crop_size, alpha=alpha), precision)
        self.clip_size = clip_size

    def forward(self, inputs, bboxes=None):
        xs = [inputs] if isinstance(inputs, torch.Tensor) else list(inputs)
        if len(xs) not in (1, 2):
            raise ValueError("SlowFast takes [slow, fast] (or one clip)")
        xs = [self._check_input(x) for x in xs]
        fast = xs[-1]
        B, _, T, H, W = fast.shape
        with torch.cuda.device(fast.device):
            eng = self._engine(self._select_dtype(), B, (T, H, W), fast.device)
            logits, pooled = eng.run_f32(*xs)
            pred = self._finish(eng, logits, pooled, B)
        return {"final_output": pred}


class _TokenParams(_Node):
    """pos_embedding / cls_token of the reference's TimeTransformer (time_transformer.py:243-244)."""

    def __init__(self, tokens: int, dim: int):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.randn(1, tokens + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))


class FtcnTT8x8(_HipNetwork):
    """Network module of the reference's second plugin, FTCN-TT (``classifier_type: i3d_temporal_var_fix_dropout_tt_cfg``
    with ``setting/ftcn_tt.yaml``; altfreezing/model/classifier/i3d_temporal_var_fix_dropout_tt_cfg.py:290-359):
    I3D-R50 trunk with every spatial kernel shrunk to 1x1 (strides replaced by 2x2 max-pools after the BN), s5
    dropped, and a one-layer TimeTransformer over 16 per-frame tokens + class token as the head.  Same parameter
    names as the reference module (the BNs followed by a pool live under ``<bn>.0``), same forward contract."""

    def __init__(self, clip_size: int = 32, imsize: int = 224, precision: str = "auto", crop_size: int = 224) -> None:
        super().__init__(ftcn_tt_spec(num_frames=clip_size, crop=crop_size), precision)
        self.clip_size, self.imsize = clip_size, imsize

    def _build_head(self):
        sp = self.spec
        h, inner = sp.head, sp.heads * sp.dim_head
        _attach(self, h, _TokenParams(sp.tokens, sp.dim))
        l0, l1 = h + ".transformer.layers.0.0.fn", h + ".transformer.layers.0.1.fn"
        _attach(self, l0 + ".norm", nn.LayerNorm(sp.dim))
        _attach(self, l0 + ".fn.to_qkv", nn.Linear(sp.dim, 3 * inner, bias=False))
        _attach(self, l0 + ".fn.to_out.0", nn.Linear(inner, sp.dim))
        _attach(self, l1 + ".norm", nn.LayerNorm(sp.dim))
        _attach(self, l1 + ".fn.net.0", nn.Linear(sp.dim, sp.mlp_dim))
        _attach(self, l1 + ".fn.net.3", nn.Linear(sp.mlp_dim, sp.dim))
        _attach(self, h + ".mlp_head.0", nn.LayerNorm(sp.dim))
        _attach(self, h + ".mlp_head.1", nn.Linear(sp.dim, sp.num_classes))

    def _head_linear(self) -> nn.Linear:
        return getattr(self.resnet.head.time_T.mlp_head, "1")

    def forward(self, images, noise=None, has_mask=None, freeze_backbone=False, return_feature_maps=False):
        assert not freeze_backbone
        x = self._check_input(images)
        B, _, T, H, W = x.shape
        if B == 0:
            return {"final_output": x.new_zeros((0, self.spec.num_classes))}
        with torch.cuda.device(x.device):
            eng = self._engine(self._select_dtype(), B, (T, H, W), x.device)
            logits, pooled = eng.run_f32(x)
            pred = self._finish(eng, logits, pooled, B)
        return {"final_output": pred}


def _unwrap_checkpoint(saved):
    if isinstance(saved, dict):
        for k in ("state_dict", "classifier_state_dict", "model_state_dict"):
            if k in saved:
                return saved[k]
    return saved


def _strip_prefix(k: str) -> str:
    for p in ("module.", "network.", "_warped_network."):
        if k.startswith(p):
            return k[len(p):]
    return k


class Classifier(nn.Module):
    """ModelBase + Classifier of the reference, for this one plugin."""

    name = "i3d_ori"

    def __init__(self, clip_size: int = 32, imsize: int = 224, precision: str = "auto",
                 model_dir: Optional[str] = None, crop_size: int = 224, streams: Optional[int] = None):
        super().__init__()
        self._build_kwargs = dict(clip_size=clip_size, imsize=imsize, precision=precision, crop_size=crop_size)
        if streams is not None:
            self._build_kwargs["streams"] = streams
        self.model_dir = model_dir
        self.network = self.build_network()
        self._warped_network = self.network

    @property
    def module_to_build(self):
        return I3D8x8

    def build_network(self) -> nn.Module:
        return self.module_to_build(**self._build_kwargs)

    def forward(self, *input, **kwargs):
        return self._warped_network(*input, **kwargs)

    def parameters(self, recurse=True):
        return self.network.parameters(recurse)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False

    def find_last(self, epoch=-1, model_dir=None):
        model_dir = model_dir or self.model_dir
        if not model_dir or not os.path.exists(model_dir):
            return None, -1
        found = {}
        for f in os.listdir(model_dir):
            if f.startswith(self.name) and f.endswith(".pth"):
                try:
                    found[int(f.split(".")[0].split("_")[-1])] = os.path.join(model_dir, f)
                except ValueError:
                    continue
        if not found:
            return None, -1
        if epoch == -1:
            e = max(found)
            return found[e], e
        if epoch not in found:
            raise RuntimeError("no checkpoint for epoch {} in {}".format(epoch, model_dir))
        return found[epoch], epoch

    def load(self, fullpath=None, epoch=-1, pretrained=None):
        """Same contract as ModelBase.load (altfreezing/model/_base.py:39-104): returns (ok, epoch);
        a missing/corrupt file (OSError / ValueError) is reported as (False, -1) WITHOUT raising."""
        if fullpath is None:
            fullpath, loaded_epoch = self.find_last(epoch)
        else:
            loaded_epoch = epoch
        if fullpath is None:
            if pretrained is None:
                logger.info("No existing %s model found", self.name)
                return False, -1
            fullpath, loaded_epoch = pretrained, -1
        try:
            saved = torch.load(fullpath, map_location="cpu", weights_only=True)
            sd = _unwrap_checkpoint(saved)
            sd = OrderedDict((_strip_prefix(k), v) for k, v in sd.items())
            param_dict = self.network.state_dict()
            loadable = {k: v for k, v in sd.items() if k in param_dict and param_dict[k].shape == v.shape}
            redundant = sorted(k for k in sd if k not in param_dict)
            unmatch = sorted(k for k in sd if k in param_dict and param_dict[k].shape != sd[k].shape)
            unfound = sorted(set(param_dict) - set(loadable) - set(unmatch))
            if redundant:
                logger.warning("%d keys are in the checkpoint but not in model %s", len(redundant), self.name)
            if unfound:
                logger.warning("%d keys are in model %s but not in the checkpoint", len(unfound), self.name)
            if unmatch:
                logger.warning("%d keys have unmatching shapes between checkpoint and model %s", len(unmatch), self.name)
            param_dict.update(loadable)
            self.network.load_state_dict(param_dict, strict=False)
            logger.info("load weights from %s", fullpath)
        except (ValueError, OSError) as err:
            logger.warning("Failed loading existing training data for %s (%s): the model keeps its current weights",
                           self.name, err)
            return False, -1
        except Exception:
            logger.error(traceback.format_exc())
            raise
        return True, loaded_epoch


class FtcnTTClassifier(Classifier):
    """``classifier_type: i3d_temporal_var_fix_dropout_tt_cfg`` (setting/ftcn_tt.yaml:60): same ModelBase surface,
    FTCN-TT network."""

    name = "i3d_temporal_var_fix_dropout_tt_cfg"

    @property
    def module_to_build(self):
        return FtcnTT8x8

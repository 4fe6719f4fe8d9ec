"""Declarative description of the AltFreezing ``i3d_ori`` network (I3D-ResNet-50).

This is the single source of truth that both the host-side module skeleton
(``classifier.py``) and the HIP execution plan (``engine.py``: the flat ``af_op`` list run by
``af_run_ops``, csrc/af_api.hip) are generated from.  It restates *what* the reference builds, not how:

* plugin yaml ``MODEL.ARCH: i3d``, ``RESNET.DEPTH: 50``  (reference
  altfreezing/model/classifier/i3d_ori.py:4-62)
* stage depths (3,4,6,3), temporal-kernel basis of ``i3d``
  (altfreezing/slowfast/models/video_model_builder.py:18,36-42,76)
* per-block temporal kernel schedule (slowfast/models/resnet_helper.py:530-534)
* bottleneck = Tx1x1 -> 1x3x3 (carries the spatial stride, STRIDE_1X1=False) -> 1x1x1
  (resnet_helper.py:255-309), projection shortcut only in block 0 (:411-436)
* stem 5x7x7/s(1,2,2) + BN + ReLU + maxpool 1x3x3/s(1,2,2)/p(0,1,1) (stem_helper.py:156-178)
* temporal max-pool (2,1,1) after s2 (video_model_builder.py:474-480)
* head: AvgPool3d([T/2, 7, 7]) -> Linear(2048, 1), no activation (head_helper.py:50-95)
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

BN_EPS = 1e-5  # stem_helper.py:20, resnet_helper.py:211 (nn.BatchNorm3d default as used)

# ImageNet statistics the callers normalise with, on 0..255 pixel values
# (reference test/af_realtime.py:72-73, altfreezing/demo.py:84-87).
IMAGENET_MEAN = (0.485, 0.456, 0.406)   # multiplied by 255 in float32 by the callers
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass(frozen=True)
class ConvSpec:
    """One Conv3d(bias=False) + BatchNorm3d(eval) [+ ReLU] unit."""
    conv: str                      # state_dict prefix of the conv   ("....branch2.a")
    bn: str                        # state_dict prefix of its BN     ("....branch2.a_bn")
    cin: int
    cout: int
    kernel: Tuple[int, int, int]   # (kT, kH, kW)
    stride: Tuple[int, int, int]
    pad: Tuple[int, int, int]
    relu: bool                     # ReLU directly after the BN
    final_bn: bool = False         # `transform_final_bn` marker (resnet_helper.py:309)
    pool_after_bn: Optional["PoolSpec"] = None   # FTCN: MaxPool3d between the BN and the ReLU; the BN then lives in an
                                                 # nn.Sequential, i.e. under `<bn>.0` in the state_dict

    @property
    def bn_key(self):
        return self.bn + ".0" if self.pool_after_bn is not None else self.bn

    @property
    def weight_shape(self):
        return (self.cout, self.cin) + tuple(self.kernel)

    def out_dims(self, t, h, w):
        o = []
        for d, k, s, p in zip((t, h, w), self.kernel, self.stride, self.pad):
            o.append((d + 2 * p - k) // s + 1)
        return tuple(o)

    def macs(self, t, h, w):
        to, ho, wo = self.out_dims(t, h, w)
        return to * ho * wo * self.cout * self.cin * self.kernel[0] * self.kernel[1] * self.kernel[2]


@dataclass(frozen=True)
class BlockSpec:
    """ResBlock: relu( shortcut(x) + c_bn(c(relu(b_bn(b(relu(a_bn(a(x)))))))) )."""
    name: str                      # "resnet.s2.pathway0_res0"
    branch1: Optional[ConvSpec]    # projection shortcut (block 0 of a stage) or None (identity)
    a: ConvSpec
    b: ConvSpec
    c: ConvSpec


@dataclass(frozen=True)
class StageSpec:
    name: str
    blocks: Tuple[BlockSpec, ...]


@dataclass(frozen=True)
class PoolSpec:
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    pad: Tuple[int, int, int]


@dataclass(frozen=True)
class NetSpec:
    num_frames: int
    crop: int
    stem: ConvSpec
    stem_pool: PoolSpec
    stages: Tuple[StageSpec, ...]
    pool_after_s2: PoolSpec
    head_pool: Tuple[int, int, int]
    head_in: int
    num_classes: int
    head: str = "resnet.head.projection"

    def convs(self) -> List[ConvSpec]:
        out = [self.stem]
        for st in self.stages:
            for blk in st.blocks:
                if blk.branch1 is not None:
                    out.append(blk.branch1)
                out += [blk.a, blk.b, blk.c]
        return out


_STAGE_DEPTH_R50 = (3, 4, 6, 3)
_I3D_TEMPORAL_BASIS = ((5,), (3,), (3, 1), (3, 1), (1, 3))   # stem, s2, s3, s4, s5
_WIDTH = 64


def _block_temporal_kernels(basis, num_blocks, num_block_temp_kernel):
    ks = (list(basis) * num_blocks)[:num_block_temp_kernel]
    return ks + [1] * (num_blocks - num_block_temp_kernel)


def i3d_r50_spec(num_frames: int = 32, crop: int = 224) -> NetSpec:
    stem = ConvSpec(
        conv="resnet.s1.pathway0_stem.conv", bn="resnet.s1.pathway0_stem.bn",
        cin=3, cout=_WIDTH, kernel=(_I3D_TEMPORAL_BASIS[0][0], 7, 7),
        stride=(1, 2, 2), pad=(_I3D_TEMPORAL_BASIS[0][0] // 2, 3, 3), relu=True)
    stages = []
    dim_in = _WIDTH
    for si, depth in enumerate(_STAGE_DEPTH_R50):
        sname = "resnet.s%d" % (si + 2)
        dim_inner = _WIDTH * (2 ** si)
        dim_out = dim_inner * 4
        stage_stride = 1 if si == 0 else 2
        tks = _block_temporal_kernels(_I3D_TEMPORAL_BASIS[si + 1], depth, depth)
        blocks = []
        for bi in range(depth):
            bname = "%s.pathway0_res%d" % (sname, bi)
            cin = dim_in if bi == 0 else dim_out
            s = stage_stride if bi == 0 else 1
            tk = tks[bi]
            branch1 = None
            if cin != dim_out or s != 1:
                branch1 = ConvSpec(bname + ".branch1", bname + ".branch1_bn", cin, dim_out,
                                   (1, 1, 1), (1, s, s), (0, 0, 0), relu=False)
            a = ConvSpec(bname + ".branch2.a", bname + ".branch2.a_bn", cin, dim_inner,
                         (tk, 1, 1), (1, 1, 1), (tk // 2, 0, 0), relu=True)
            b = ConvSpec(bname + ".branch2.b", bname + ".branch2.b_bn", dim_inner, dim_inner,
                         (1, 3, 3), (1, s, s), (0, 1, 1), relu=True)
            c = ConvSpec(bname + ".branch2.c", bname + ".branch2.c_bn", dim_inner, dim_out,
                         (1, 1, 1), (1, 1, 1), (0, 0, 0), relu=False, final_bn=True)
            blocks.append(BlockSpec(bname, branch1, a, b, c))
        stages.append(StageSpec(sname, tuple(blocks)))
        dim_in = dim_out
    return NetSpec(
        num_frames=num_frames, crop=crop, stem=stem,
        stem_pool=PoolSpec((1, 3, 3), (1, 2, 2), (0, 1, 1)),
        stages=tuple(stages),
        pool_after_s2=PoolSpec((2, 1, 1), (2, 1, 1), (0, 0, 0)),
        head_pool=(num_frames // 2, crop // 32, crop // 32),
        head_in=dim_in, num_classes=1)


def _bottleneck_stage(sname: str, pathway: int, dim_in: int, dim_out: int, dim_inner: int, depth: int,
                      stride: int, tks) -> StageSpec:
    """ResStage of one pathway (resnet_helper.py:585-603): block 0 carries the stride and the projection."""
    blocks = []
    for bi in range(depth):
        bname = "%s.pathway%d_res%d" % (sname, pathway, bi)
        cin = dim_in if bi == 0 else dim_out
        s = stride if bi == 0 else 1
        tk = tks[bi]
        branch1 = None
        if cin != dim_out or s != 1:
            branch1 = ConvSpec(bname + ".branch1", bname + ".branch1_bn", cin, dim_out, (1, 1, 1), (1, s, s), (0, 0, 0),
                               relu=False)
        a = ConvSpec(bname + ".branch2.a", bname + ".branch2.a_bn", cin, dim_inner, (tk, 1, 1), (1, 1, 1),
                     (tk // 2, 0, 0), relu=True)
        b = ConvSpec(bname + ".branch2.b", bname + ".branch2.b_bn", dim_inner, dim_inner, (1, 3, 3), (1, s, s),
                     (0, 1, 1), relu=True)
        c = ConvSpec(bname + ".branch2.c", bname + ".branch2.c_bn", dim_inner, dim_out, (1, 1, 1), (1, 1, 1),
                     (0, 0, 0), relu=False, final_bn=True)
        blocks.append(BlockSpec(bname, branch1, a, b, c))
    return StageSpec(sname, tuple(blocks))


@dataclass(frozen=True)
class SlowFastSpec:
    """Two-pathway SlowFast-R50 (reference video_model_builder.py:146-387; never instantiated by a shipped plugin).
    Pathway 0 = Slow (every alpha-th frame, full width), pathway 1 = Fast (all frames, width / beta_inv).
    After s1..s4 the Fast tensor is fused into the Slow one: Conv3d([5,1,1], stride [alpha,1,1]) + BN + ReLU,
    concatenated by channel (FuseFastToSlow, :86-143)."""
    num_frames: int
    crop: int
    alpha: int
    stems: Tuple[ConvSpec, ConvSpec]
    stem_pool: PoolSpec
    fuses: Tuple[ConvSpec, ...]                       # after s1, s2, s3, s4
    stages: Tuple[Tuple[StageSpec, StageSpec], ...]   # (slow, fast) for s2..s5
    head_pools: Tuple[Tuple[int, int, int], Tuple[int, int, int]]
    head_in: int
    num_classes: int
    head: str = "resnet.head.projection"

    def convs(self) -> List[ConvSpec]:
        """state_dict order: s1 (both stems), s1_fuse, s2 (slow blocks, then fast blocks), s2_fuse, ..."""
        out = [self.stems[0], self.stems[1], self.fuses[0]]
        for si, (slow, fast) in enumerate(self.stages):
            for st in (slow, fast):
                for blk in st.blocks:
                    if blk.branch1 is not None:
                        out.append(blk.branch1)
                    out += [blk.a, blk.b, blk.c]
            if si + 1 < len(self.fuses):
                out.append(self.fuses[si + 1])
        return out


_SLOWFAST_TEMPORAL_BASIS = (((1,), (5,)), ((1,), (3,)), ((1,), (3,)), ((3,), (3,)), ((3,), (3,)))


def slowfast_r50_spec(num_frames: int = 32, crop: int = 224, alpha: int = 8, beta_inv: int = 8,
                      fusion_ratio: int = 2, fusion_kernel: int = 5) -> SlowFastSpec:
    wf = _WIDTH // beta_inv
    kt_s, kt_f = _SLOWFAST_TEMPORAL_BASIS[0][0][0], _SLOWFAST_TEMPORAL_BASIS[0][1][0]
    stems = (ConvSpec("resnet.s1.pathway0_stem.conv", "resnet.s1.pathway0_stem.bn", 3, _WIDTH, (kt_s, 7, 7), (1, 2, 2),
                      (kt_s // 2, 3, 3), relu=True),
             ConvSpec("resnet.s1.pathway1_stem.conv", "resnet.s1.pathway1_stem.bn", 3, wf, (kt_f, 7, 7), (1, 2, 2),
                      (kt_f // 2, 3, 3), relu=True))

    def fuse(name, cfast):
        return ConvSpec("resnet.%s.conv_f2s" % name, "resnet.%s.bn" % name, cfast, cfast * fusion_ratio,
                        (fusion_kernel, 1, 1), (alpha, 1, 1), (fusion_kernel // 2, 0, 0), relu=True)

    fuses = [fuse("s1_fuse", wf)]
    stages = []
    slow_in, fast_in = _WIDTH + wf * fusion_ratio, wf
    for si, depth in enumerate(_STAGE_DEPTH_R50):
        sname = "resnet.s%d" % (si + 2)
        inner = _WIDTH * (2 ** si)
        out = inner * 4
        stride = 1 if si == 0 else 2
        tk_s = _block_temporal_kernels(_SLOWFAST_TEMPORAL_BASIS[si + 1][0], depth, depth)
        tk_f = _block_temporal_kernels(_SLOWFAST_TEMPORAL_BASIS[si + 1][1], depth, depth)
        stages.append((_bottleneck_stage(sname, 0, slow_in, out, inner, depth, stride, tk_s),
                       _bottleneck_stage(sname, 1, fast_in, out // beta_inv, inner // beta_inv, depth, stride, tk_f)))
        if si < 3:
            fuses.append(fuse("s%d_fuse" % (si + 2), out // beta_inv))
        slow_in, fast_in = out + (out // beta_inv) * fusion_ratio, out // beta_inv
    c_slow, c_fast = _WIDTH * 32, _WIDTH * 32 // beta_inv
    return SlowFastSpec(
        num_frames=num_frames, crop=crop, alpha=alpha, stems=stems,
        stem_pool=PoolSpec((1, 3, 3), (1, 2, 2), (0, 1, 1)), fuses=tuple(fuses), stages=tuple(stages),
        head_pools=((num_frames // alpha, crop // 32, crop // 32), (num_frames, crop // 32, crop // 32)),
        head_in=c_slow + c_fast, num_classes=1)


@dataclass(frozen=True)
class FtcnTTSpec:
    """The repo's second classifier plugin, FTCN-TT (reference altfreezing/model/classifier/
    i3d_temporal_var_fix_dropout_tt_cfg.py:207-359 with setting/ftcn_tt.yaml): the I3D-R50 trunk with EVERY spatial
    kernel shrunk to 1x1 (`temporal_only_conv`, :207-288: a stride-2 conv becomes stride 1 + MaxPool3d((1,2,2)) placed
    after its BN, before the ReLU), s5 dropped (`stop_point: 5`), and a one-layer pre-norm transformer over the
    T/2 = 16 per-frame tokens (AvgPool3d((1,14,14)) of s4) + a class token as the head (:127-197,
    time_transformer.py:219-276: dim 1024, 16 heads x 64, MLP 2048, GELU)."""
    num_frames: int
    crop: int
    stem: ConvSpec
    stem_pool: PoolSpec
    stages: Tuple[StageSpec, ...]
    pool_after_s2: PoolSpec
    tokens: int                    # per-frame tokens (the class token is extra)
    token_pool: Tuple[int, int, int]
    dim: int
    heads: int
    dim_head: int
    mlp_dim: int
    num_classes: int
    head: str = "resnet.head.time_T"

    def convs(self) -> List[ConvSpec]:
        out = [self.stem]
        for st in self.stages:
            for blk in st.blocks:
                if blk.branch1 is not None:
                    out.append(blk.branch1)
                out += [blk.a, blk.b, blk.c]
        return out

    def head_layout(self):
        h, d, inner = self.head, self.dim, self.heads * self.dim_head
        l0, l1 = h + ".transformer.layers.0.0.fn", h + ".transformer.layers.0.1.fn"
        return [(h + ".pos_embedding", (1, self.tokens + 1, d), "float32"), (h + ".cls_token", (1, 1, d), "float32"),
                (l0 + ".norm.weight", (d,), "float32"), (l0 + ".norm.bias", (d,), "float32"),
                (l0 + ".fn.to_qkv.weight", (3 * inner, d), "float32"),
                (l0 + ".fn.to_out.0.weight", (d, inner), "float32"), (l0 + ".fn.to_out.0.bias", (d,), "float32"),
                (l1 + ".norm.weight", (d,), "float32"), (l1 + ".norm.bias", (d,), "float32"),
                (l1 + ".fn.net.0.weight", (self.mlp_dim, d), "float32"), (l1 + ".fn.net.0.bias", (self.mlp_dim,), "float32"),
                (l1 + ".fn.net.3.weight", (d, self.mlp_dim), "float32"), (l1 + ".fn.net.3.bias", (d,), "float32"),
                (h + ".mlp_head.0.weight", (d,), "float32"), (h + ".mlp_head.0.bias", (d,), "float32"),
                (h + ".mlp_head.1.weight", (self.num_classes, d), "float32"),
                (h + ".mlp_head.1.bias", (self.num_classes,), "float32")]

    def linear_prefixes(self):
        h = self.head
        return [h + ".transformer.layers.0.0.fn.fn.to_qkv", h + ".transformer.layers.0.0.fn.fn.to_out.0",
                h + ".transformer.layers.0.1.fn.fn.net.0", h + ".transformer.layers.0.1.fn.fn.net.3", h + ".mlp_head.1"]


def ftcn_tt_spec(num_frames: int = 32, crop: int = 224) -> FtcnTTSpec:
    pool2 = PoolSpec((1, 2, 2), (1, 2, 2), (0, 0, 0))
    kt = _I3D_TEMPORAL_BASIS[0][0]
    stem = ConvSpec("resnet.s1.pathway0_stem.conv", "resnet.s1.pathway0_stem.bn", 3, _WIDTH, (kt, 1, 1), (1, 1, 1),
                    (kt // 2, 0, 0), relu=True, pool_after_bn=pool2)
    stages = []
    dim_in = _WIDTH
    for si, depth in enumerate(_STAGE_DEPTH_R50[:3]):                 # s2..s4; s5 is an nn.Identity (stop_point 5)
        sname = "resnet.s%d" % (si + 2)
        inner, dim_out = _WIDTH * (2 ** si), _WIDTH * (2 ** si) * 4
        tks = _block_temporal_kernels(_I3D_TEMPORAL_BASIS[si + 1], depth, depth)
        blocks = []
        for bi in range(depth):
            bname = "%s.pathway0_res%d" % (sname, bi)
            cin = dim_in if bi == 0 else dim_out
            down = pool2 if (si > 0 and bi == 0) else None             # where the reference net strides by 2
            tk = tks[bi]
            branch1 = None
            if cin != dim_out or down is not None:
                branch1 = ConvSpec(bname + ".branch1", bname + ".branch1_bn", cin, dim_out, (1, 1, 1), (1, 1, 1),
                                   (0, 0, 0), relu=False, pool_after_bn=down)
            a = ConvSpec(bname + ".branch2.a", bname + ".branch2.a_bn", cin, inner, (tk, 1, 1), (1, 1, 1),
                         (tk // 2, 0, 0), relu=True)
            b = ConvSpec(bname + ".branch2.b", bname + ".branch2.b_bn", inner, inner, (1, 1, 1), (1, 1, 1), (0, 0, 0),
                         relu=True, pool_after_bn=down)
            c = ConvSpec(bname + ".branch2.c", bname + ".branch2.c_bn", inner, dim_out, (1, 1, 1), (1, 1, 1),
                         (0, 0, 0), relu=False, final_bn=True)
            blocks.append(BlockSpec(bname, branch1, a, b, c))
        stages.append(StageSpec(sname, tuple(blocks)))
        dim_in = dim_out
    return FtcnTTSpec(num_frames=num_frames, crop=crop, stem=stem, stem_pool=PoolSpec((1, 3, 3), (1, 2, 2), (0, 1, 1)),
                      stages=tuple(stages), pool_after_s2=PoolSpec((2, 1, 1), (2, 1, 1), (0, 0, 0)),
                      tokens=num_frames // 2, token_pool=(1, crop // 16, crop // 16), dim=dim_in, heads=16,
                      dim_head=64, mlp_dim=2048, num_classes=1)


def state_dict_layout(spec):
    """[(key, shape, dtype_name)] in the reference's ``network.state_dict()`` order
    (320 keys for the i3d 32x224 config; 662 keys for SlowFast-R50)."""
    out = []
    for cv in spec.convs():
        out.append((cv.conv + ".weight", cv.weight_shape, "float32"))
        out.append((cv.bn_key + ".weight", (cv.cout,), "float32"))
        out.append((cv.bn_key + ".bias", (cv.cout,), "float32"))
        out.append((cv.bn_key + ".running_mean", (cv.cout,), "float32"))
        out.append((cv.bn_key + ".running_var", (cv.cout,), "float32"))
        out.append((cv.bn_key + ".num_batches_tracked", (), "int64"))
    if isinstance(spec, FtcnTTSpec):
        return out + spec.head_layout()
    out.append((spec.head + ".weight", (spec.num_classes, spec.head_in), "float32"))
    out.append((spec.head + ".bias", (spec.num_classes,), "float32"))
    return out


def conv_macs_per_clip(spec: NetSpec):
    """Walks the activation shapes; returns (total_macs, [(ConvSpec, (T,H,W) in, macs)])."""
    t, h, w = spec.num_frames, spec.crop, spec.crop
    rows = []

    def pool(dims, p: PoolSpec):
        return tuple((d + 2 * pp - k) // s + 1 for d, k, s, pp in zip(dims, p.kernel, p.stride, p.pad))

    rows.append((spec.stem, (t, h, w), spec.stem.macs(t, h, w)))
    t, h, w = spec.stem.out_dims(t, h, w)
    t, h, w = pool((t, h, w), spec.stem_pool)
    for si, st in enumerate(spec.stages):
        for blk in st.blocks:
            if blk.branch1 is not None:
                rows.append((blk.branch1, (t, h, w), blk.branch1.macs(t, h, w)))
            rows.append((blk.a, (t, h, w), blk.a.macs(t, h, w)))
            ta, ha, wa = blk.a.out_dims(t, h, w)
            rows.append((blk.b, (ta, ha, wa), blk.b.macs(ta, ha, wa)))
            tb, hb, wb = blk.b.out_dims(ta, ha, wa)
            rows.append((blk.c, (tb, hb, wb), blk.c.macs(tb, hb, wb)))
            t, h, w = blk.c.out_dims(tb, hb, wb)
        if si == 0:
            t, h, w = pool((t, h, w), spec.pool_after_s2)
    return sum(r[2] for r in rows), rows

"""Host side of the HIP forward: turns a network spec + checkpoint tensors into a flat list of
libafhip ops over caller-owned HBM buffers, and runs it with ONE C call per forward.

What this orchestrates is the reference's ``ResNet.forward`` / ``SlowFast.forward``
(altfreezing/slowfast/models/video_model_builder.py:561-578, :370-387): stem(s) -> [lateral] -> s2 -> [lateral] ->
pool -> s3 -> ... -> head, with every Conv3d+BatchNorm3d(+add)(+ReLU)(+pool) group collapsed into one kernel
launch.  PyTorch is used only to own device memory and the stream.
"""
import ctypes as C
import dataclasses
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import Op, check, lib
from .arch import BN_EPS, ConvSpec, FtcnTTSpec, NetSpec, PoolSpec, SlowFastSpec

_TORCH_DTYPE = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}

# op tags (echoed by the timed runner; used by bench.py to attribute device time to kernel classes)
TAG_PACK, TAG_STEM, TAG_POOL, TAG_HEAD = 0, 1, 2, 3
TAG_CONV_1x1x1, TAG_CONV_Tx1x1, TAG_CONV_1x3x3, TAG_CONV_OTHER, TAG_CONV_BC, TAG_CONV_CA, TAG_BLOCK_ABC = 10, 11, 12, 13, 14, 15, 16
TAG_NAMES = {TAG_PACK: "input_pack", TAG_STEM: "stem_5x7x7", TAG_POOL: "maxpool", TAG_HEAD: "head",
             TAG_CONV_1x1x1: "conv_1x1x1", TAG_CONV_Tx1x1: "conv_3x1x1", TAG_CONV_1x3x3: "conv_1x3x3",
             TAG_CONV_OTHER: "conv_other", TAG_CONV_BC: "conv_1x3x3+1x1x1_fused", TAG_CONV_CA: "conv_1x1x1+3x1x1_fused",
             TAG_BLOCK_ABC: "bottleneck_block_fused"}
TAG_NAMES[TAG_STEM] = "stem"


def _conv_tag(cv: ConvSpec) -> int:
    k = tuple(cv.kernel)
    if k == (1, 1, 1):
        return TAG_CONV_1x1x1
    if k[1:] == (1, 1):
        return TAG_CONV_Tx1x1
    if k[0] == 1 and k[1:] == (3, 3):
        return TAG_CONV_1x3x3
    return TAG_CONV_OTHER


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _pool_out(dims, p: PoolSpec):
    return tuple((d + 2 * pp - k) // s + 1 for d, k, s, pp in zip(dims, p.kernel, p.stride, p.pad))


def _is_pool(p: PoolSpec, k, s, pad):
    return (tuple(p.kernel), tuple(p.stride), tuple(p.pad)) == (k, s, pad)


def _stages_of(spec):
    if isinstance(spec, SlowFastSpec):
        return [st for pair in spec.stages for st in pair]
    return list(spec.stages)

class PackedWeights:
    """Device-resident, kernel-ready copy of a checkpoint: per conv a packed weight in the compute
    dtype plus fp32 BatchNorm scale/shift; fp32 head.  Built with HIP kernels (af_pack.hip)."""

    def __init__(self, spec, state: Dict[str, torch.Tensor], dtype: str, device):
        self.dtype = dtype
        code = _lib.DTYPE_CODES[dtype]
        es = 4 if dtype == "f32" else 2
        st = _stream_ptr(device)
        stems = set(c.conv for c in (spec.stems if isinstance(spec, SlowFastSpec) else (spec.stem,)))
        self.w, self.scale, self.shift = {}, {}, {}
        for cv in spec.convs():
            w = state[cv.conv + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
            assert tuple(w.shape) == cv.weight_shape, (cv.conv, tuple(w.shape), cv.weight_shape)
            bn = [state[cv.bn_key + s].detach().to(device=device, dtype=torch.float32).contiguous()
                  for s in (".weight", ".bias", ".running_mean", ".running_var")]
            cpad = lib.af_padded_channels(cv.cout)           # kernels read scale/shift over the padded channel tile
            scale = torch.zeros(cpad, dtype=torch.float32, device=device)
            shift = torch.zeros(cpad, dtype=torch.float32, device=device)
            check(lib.af_fold_bn(_ptr(bn[0]), _ptr(bn[1]), _ptr(bn[2]), _ptr(bn[3]), BN_EPS, cv.cout,
                                 _ptr(scale), _ptr(shift), st), "af_fold_bn")
            kt, kh, kw = cv.kernel
            if cv.conv in stems and isinstance(spec, FtcnTTSpec):        # temporal stem: MFMA A-fragment image
                nbytes = lib.af_packed_tstem_weight_bytes(code)
                packed = torch.empty(nbytes // es, dtype=_TORCH_DTYPE[dtype], device=device)
                check(lib.af_pack_tstem_weight(_ptr(w), cv.cout, kt, code, _ptr(packed), st), "af_pack_tstem_weight")
            elif cv.conv in stems:
                nbytes = lib.af_packed_stem_weight_bytes(cv.cout, kt, kh, code)
                packed = torch.empty(nbytes // es, dtype=_TORCH_DTYPE[dtype], device=device)
                check(lib.af_pack_stem_weight(_ptr(w), cv.cout, kt, kh, kw, code, _ptr(packed), st),
                      "af_pack_stem_weight")
            else:
                nbytes = lib.af_packed_conv_weight_bytes(cv.cout, cv.cin, kt, kh, kw, code)
                packed = torch.empty(nbytes // es, dtype=_TORCH_DTYPE[dtype], device=device)
                check(lib.af_pack_conv_weight(_ptr(w), cv.cout, cv.cin, kt, kh, kw, code, _ptr(packed), st),
                      "af_pack_conv_weight")
            self.w[cv.conv], self.scale[cv.conv], self.shift[cv.conv] = packed, scale, shift
            if (cv.conv in stems and not isinstance(spec, FtcnTTSpec) and dtype != "f32" and cv.cout == 64
                    and (kh, kw) == (7, 7)):
                # the K-packed image of the same stem weights for the fused 16-bit stem (3 real channels, af_stem3.hip)
                nb3 = lib.af_packed_stem_weight_bytes_rgb3(kt, code)
                p3 = torch.empty(nb3 // es, dtype=_TORCH_DTYPE[dtype], device=device)
                check(lib.af_pack_stem_weight_rgb3(_ptr(w), cv.cout, kt, code, _ptr(p3), st), "af_pack_stem_weight_rgb3")
                self.w3 = getattr(self, "w3", {})
                self.w3[cv.conv] = p3
        # block 0 of every stage: last 1x1x1 + projection shortcut share one accumulator (af_conv3d_dual_bn_act):
        # both weights get their BN scale folded in (fp32, before the rounding), shifts are summed
        self.w_folded, self.shift_sum, self.ones = {}, {}, {}
        for stage in _stages_of(spec):
            for blk in stage.blocks:
                if blk.branch1 is None or blk.branch1.pool_after_bn is not None:
                    continue                 # (FTCN: a pooled shortcut cannot share the c conv's accumulator)
                for cv in (blk.c, blk.branch1):
                    w = state[cv.conv + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
                    kt, kh, kw = cv.kernel
                    nbytes = lib.af_packed_conv_weight_bytes(cv.cout, cv.cin, kt, kh, kw, code)
                    packed = torch.empty(nbytes // es, dtype=_TORCH_DTYPE[dtype], device=device)
                    check(lib.af_pack_conv_weight_scaled(_ptr(w), _ptr(self.scale[cv.conv]), cv.cout, cv.cin, kt, kh, kw,
                                                         code, _ptr(packed), st), "af_pack_conv_weight_scaled")
                    self.w_folded[cv.conv] = packed
                self.shift_sum[blk.c.conv] = (self.shift[blk.c.conv] + self.shift[blk.branch1.conv]).contiguous()
                self.ones[blk.c.conv] = torch.ones(lib.af_padded_channels(blk.c.cout), dtype=torch.float32, device=device)
        f32 = lambda k: state[k].detach().to(device=device, dtype=torch.float32).contiguous()
        if isinstance(spec, FtcnTTSpec):
            # transformer head, fp32 whatever the trunk's dtype: Linear weights packed for the fp32 conv kernel,
            # bias as its `shift` (scale = ones), LayerNorm / token parameters as they are
            h = spec.head
            l0, l1 = h + ".transformer.layers.0.0.fn", h + ".transformer.layers.0.1.fn"
            f32code = _lib.DTYPE_CODES["f32"]
            self.lin = {}
            for name, wkey, bkey in (("qkv", l0 + ".fn.to_qkv.weight", None),
                                     ("out", l0 + ".fn.to_out.0.weight", l0 + ".fn.to_out.0.bias"),
                                     ("ff1", l1 + ".fn.net.0.weight", l1 + ".fn.net.0.bias"),
                                     ("ff2", l1 + ".fn.net.3.weight", l1 + ".fn.net.3.bias")):
                w = f32(wkey)
                cout, cin = w.shape
                nbytes = lib.af_packed_conv_weight_bytes(cout, cin, 1, 1, 1, f32code)
                packed = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
                check(lib.af_pack_conv_weight(_ptr(w), cout, cin, 1, 1, 1, f32code, _ptr(packed), st), "af_pack_conv_weight")
                cpad = lib.af_padded_channels(cout)
                bias = torch.zeros(cpad, dtype=torch.float32, device=device)
                if bkey is not None:
                    bias[:cout] = f32(bkey)
                self.lin[name] = (packed, torch.ones(cpad, dtype=torch.float32, device=device), bias, cin, cout)
            self.ln = {"attn": (f32(l0 + ".norm.weight"), f32(l0 + ".norm.bias")),
                       "ff": (f32(l1 + ".norm.weight"), f32(l1 + ".norm.bias")),
                       "head": (f32(h + ".mlp_head.0.weight"), f32(h + ".mlp_head.0.bias"))}
            self.cls_token = f32(h + ".cls_token").reshape(-1)
            self.pos_embedding = f32(h + ".pos_embedding").reshape(spec.tokens + 1, spec.dim)
            self.fc_w, self.fc_b = f32(h + ".mlp_head.1.weight"), f32(h + ".mlp_head.1.bias")
        else:
            self.fc_w, self.fc_b = f32(spec.head + ".weight"), f32(spec.head + ".bias")
        torch.cuda.current_stream(device).synchronize()      # sources may be freed by the caller


def _fill_conv_desc(cd, batch, code, cv: ConvSpec, din, dout, relu):
    cd.n, (cd.t, cd.h, cd.w), cd.cin, cd.cout = batch, din, cv.cin, cv.cout
    cd.kt, cd.kh, cd.kw = cv.kernel
    cd.st, cd.sh, cd.sw = cv.stride
    cd.pt, cd.ph, cd.pw = cv.pad
    cd.to, cd.ho, cd.wo = dout
    cd.relu, cd.dtype = int(relu), code


class _Plan:
    """Accumulates plan entries (dicts) and the element count every named activation buffer must hold."""

    def __init__(self, batch, code=None):
        self.batch, self.entries, self.sizes, self.code = batch, [], {}, code
        self.boundary = None                 # set by a stage whose last launch already ran the next stage's first a conv

    def bc_fusable(self, b: ConvSpec, c: ConvSpec, da, db, dc) -> bool:
        """does the library run this bottleneck's b (1x3x3) and c (1x1x1 + residual) convs as one launch (af_conv3d_bc_bn_act)?
        OFF unless AF_FUSE_BC=1: measured slower (B=16, bf16: s3 0.205 ms against 0.068 + 0.080 ms for the two launches, s4
        0.134 against 0.056 + 0.070) - the b conv is MFMA-bound, the c conv with its residual an HBM stream, and inside one
        workgroup the two phases run back to back on every CU at the same time instead of overlapping."""
        if self.code is None or os.environ.get("AF_FUSE_BC") != "1":
            return False
        d1, d2 = _lib.ConvDesc(), _lib.ConvDesc()
        _fill_conv_desc(d1, self.batch, self.code, b, da, db, True)
        _fill_conv_desc(d2, self.batch, self.code, c, db, dc, True)
        return bool(lib.af_conv_bc_fusable(C.byref(d1), C.byref(d2)))

    def need(self, buf, dims, width):
        self.sizes[buf] = max(self.sizes.get(buf, 0), self.batch * dims[0] * dims[1] * dims[2] * width)

    def add(self, **e):
        self.entries.append(e)

    def ca_fusable(self, c: ConvSpec, a_next: ConvSpec, db, dc, branch1: ConvSpec = None, d_in=None) -> bool:
        """does the library run this block's c conv (+ residual or projection shortcut, + ReLU) and the NEXT block's a conv as one
        launch (af_conv3d_ca_bn_act)?"""
        if self.code is None or os.environ.get("AF_FUSE_CA", "1") != "1":
            return False
        d1, d2, d3 = _lib.ConvDesc(), _lib.ConvDesc(), _lib.ConvDesc()
        _fill_conv_desc(d1, self.batch, self.code, c, db, dc, True)
        _fill_conv_desc(d2, self.batch, self.code, a_next, dc, a_next.out_dims(*dc), True)
        if branch1 is not None:
            _fill_conv_desc(d3, self.batch, self.code, branch1, d_in, dc, True)
        return bool(lib.af_conv_ca_fusable(C.byref(d1), C.byref(d3) if branch1 is not None else None, C.byref(d2)))

    def cpa_fusable(self, c: ConvSpec, a_next: ConvSpec, db, dc, x_sub) -> bool:
        """does the library run the stage's last c conv (+ residual + ReLU), the temporal max-pool behind the stage and the NEXT
        stage's first a conv as one launch (af_conv3d_cpa_bn_act)?  AF_FUSE_CPA=0 switches it off (A/B runs)."""
        if self.code is None or os.environ.get("AF_FUSE_CPA", "0") != "1":
            return False
        dp = (dc[0] // 2,) + tuple(dc[1:])
        d1, d2 = _lib.ConvDesc(), _lib.ConvDesc()
        _fill_conv_desc(d1, self.batch, self.code, c, db, dc, True)
        d1.tpool = 1
        _fill_conv_desc(d2, self.batch, self.code, a_next, dp, a_next.out_dims(*dp), True)
        return bool(lib.af_conv_cpa_fusable(C.byref(d1), C.byref(d2), x_sub))

    def _boundary_fusable(self, c: ConvSpec, next_stage, db, dc) -> int:
        """0, or the x_sub of the fused stage-boundary launch: 2 when the next stage's block 0 reads the pooled trunk only through
        a 1x1x1 projection shortcut of stride (1,2,2) (and its a conv, which runs inside the launch), else 1."""
        nb = next_stage.blocks[0]
        if nb.a.pool_after_bn is not None or dc[0] % 2:
            return 0
        b1 = nb.branch1
        sub = 2 if (b1 is not None and b1.pool_after_bn is None and tuple(b1.kernel) == (1, 1, 1) and tuple(b1.stride) == (1, 2, 2)
                    and tuple(b1.pad) == (0, 0, 0) and dc[1] % 2 == 0 and dc[2] % 2 == 0) else 1
        if b1 is None:
            return 0                       # an identity shortcut would need the whole trunk AND block 0's residual add: not this launch
        return sub if self.cpa_fusable(c, nb.a, db, dc, sub) else 0

    def abc_fusable(self, blk, d) -> bool:
        """does the library run this whole block (a, b, c + identity or stride-1 projection shortcut + ReLU) as one launch
        (af_block_abc_bn_act: the narrow blocks of SlowFast's Fast pathway)?  AF_FUSE_ABC=0 switches it off (A/B runs)."""
        if self.code is None or os.environ.get("AF_FUSE_ABC", "1") != "1":
            return False
        if any(cv is not None and cv.pool_after_bn is not None for cv in (blk.a, blk.b, blk.c, blk.branch1)):
            return False
        da = blk.a.out_dims(*d)
        db = blk.b.out_dims(*da)
        dc = blk.c.out_dims(*db)
        d1, d2, d3, d4 = _lib.ConvDesc(), _lib.ConvDesc(), _lib.ConvDesc(), _lib.ConvDesc()
        _fill_conv_desc(d1, self.batch, self.code, blk.a, d, da, True)
        _fill_conv_desc(d2, self.batch, self.code, blk.b, da, db, True)
        _fill_conv_desc(d3, self.batch, self.code, blk.c, db, dc, True)
        if blk.branch1 is not None:
            if blk.branch1.out_dims(*d) != dc:
                return False
            _fill_conv_desc(d4, self.batch, self.code, blk.branch1, d, dc, True)
        return bool(lib.af_block_abc_fusable(C.byref(d1), C.byref(d2), C.byref(d3), C.byref(d4) if blk.branch1 is not None else None))

    def stage(self, stage, d, cur, nxt, a_buf, b_buf, last_ld=None, tpool_last=False, next_stage=None):
        """One pathway's ResStage.  ``last_ld``: row stride of the stage's final output (room for the lateral's
        channels); ``next_stage`` (with ``tpool_last``): the stage behind the temporal pool, whose first a conv may ride in this
        stage's last launch.  Returns (dims, channels, buffer holding the output, the other trunk buffer)."""
        c = None
        nblk = len(stage.blocks)
        # this block's a conv was run by the previous block's fused c -> a launch (or, block 0, by the previous STAGE's last one)
        bnd, self.boundary = self.boundary, None
        a_done = bnd is not None
        sub_first = bool(bnd and bnd.get("sub"))
        for bi, blk in enumerate(stage.blocks):
            last = bi == nblk - 1
            if not a_done and not (tpool_last and last) and self.abc_fusable(blk, d):
                # the whole block in one launch (narrow pathway): trunk in, trunk out, a and b stay on the CU
                ld = last_ld if (last and last_ld) else blk.c.cout
                self.add(kind="abc", cv=blk.a, cv2=blk.b, cv3=blk.c, cv4=blk.branch1, din=d, dout=d, src=cur, dst=nxt, ld=ld)
                self.need(nxt, d, ld)
                cur, nxt = nxt, cur
                c = blk.c.cout
                continue
            da = blk.a.out_dims(*d)
            if not a_done:
                self.add(kind="conv", cv=blk.a, din=d, dout=da, src=cur, dst=a_buf)
            self.need(a_buf, da, blk.a.cout)
            a_done = False
            db = blk.b.out_dims(*da)
            c_src, res_src = b_buf, cur
            pb = blk.b.pool_after_bn
            if pb is not None and _is_pool(pb, (1, 2, 2), (1, 2, 2), (0, 0, 0)) and db[1] % 2 == 0 and db[2] % 2 == 0:
                # FTCN: conv -> BN -> MaxPool3d((1,2,2)) -> ReLU in one launch (max and ReLU commute): the conv's tile
                # rows are ordered so that a 2x2 window is 4 adjacent rows and the epilogue stores their max
                dbp = _pool_out(db, pb)
                self.add(kind="conv", cv=blk.b, din=da, dout=db, src=a_buf, dst=b_buf, tpool=2); self.need(b_buf, dbp, blk.b.cout)
                db = dbp
            elif (pb is None and blk.branch1 is None and not (tpool_last and bi == nblk - 1) and
                  self.bc_fusable(blk.b, blk.c, da, db, blk.c.out_dims(*db))):
                # b + c + residual + ReLU in one launch: the b output of a frame stays in LDS (s3 / s4 at bench batch sizes)
                dc = blk.c.out_dims(*db)
                ld = last_ld if (bi == nblk - 1 and last_ld) else blk.c.cout
                self.add(kind="bc", cv=blk.b, cv2=blk.c, din=da, dmid=db, dout=dc, src=a_buf, dst=nxt, res=cur, ld=ld)
                self.need(nxt, dc, ld)
                cur, nxt = nxt, cur
                d, c = dc, blk.c.cout
                continue
            else:
                self.add(kind="conv", cv=blk.b, din=da, dout=db, src=a_buf, dst=b_buf); self.need(b_buf, db, blk.b.cout)
                if pb is not None:       # odd sizes / other pool shapes: the pool as its own launch into the free a buffer
                    dbp = _pool_out(db, pb)
                    self.add(kind="pool", pool=pb, ch=blk.b.cout, din=db, dout=dbp, src=b_buf, dst=a_buf)
                    self.need(a_buf, dbp, blk.b.cout)
                    db, c_src = dbp, a_buf
            if blk.branch1 is not None and blk.branch1.pool_after_bn is not None:
                # pooled projection shortcut: conv + BN, pool, then it is the residual of the c conv
                p1 = blk.branch1.pool_after_bn
                d1 = blk.branch1.out_dims(*d)
                d1p = _pool_out(d1, p1)
                if _is_pool(p1, (1, 2, 2), (1, 2, 2), (0, 0, 0)) and d1[1] % 2 == 0 and d1[2] % 2 == 0:
                    self.add(kind="conv", cv=blk.branch1, din=d, dout=d1, src=cur, dst="R1", tpool=2)
                else:
                    self.add(kind="conv", cv=blk.branch1, din=d, dout=d1, src=cur, dst="R0"); self.need("R0", d1, blk.branch1.cout)
                    self.add(kind="pool", pool=p1, ch=blk.branch1.cout, din=d1, dout=d1p, src="R0", dst="R1")
                self.need("R1", d1p, blk.branch1.cout)
                res_src = "R1"
            dc = blk.c.out_dims(*db)
            ld = last_ld if (last and last_ld) else blk.c.cout
            # the temporal max-pool after s2 rides in the epilogue of s2's last conv when it can
            tp = tpool_last and last and blk.branch1 is None and dc[0] % 2 == 0
            dstore = (dc[0] // 2,) + tuple(dc[1:]) if tp else dc
            if blk.branch1 is not None and blk.branch1.pool_after_bn is None and bi == 0 and sub_first:
                # the previous stage's last launch stored the trunk only where this stride-(1,2,2) shortcut reads it, packed:
                # the same weights as a stride-1 convolution over that tensor
                b1 = dataclasses.replace(blk.branch1, stride=(1, 1, 1))
                dsub = (d[0], d[1] // 2, d[2] // 2)
                assert b1.out_dims(*dsub) == dc
                self.add(kind="dual", cv=blk.c, cv2=b1, din=db, din2=dsub, dout=dc, src=c_src, src2=cur, dst=nxt, ld=ld)
            elif blk.branch1 is not None and blk.branch1.pool_after_bn is None:
                # c conv + projection shortcut in one launch; no shortcut tensor
                assert blk.branch1.out_dims(*d) == dc
                if (not last and ld == blk.c.cout and stage.blocks[bi + 1].a.pool_after_bn is None
                        and self.ca_fusable(blk.c, stage.blocks[bi + 1].a, db, dc, blk.branch1, d)):
                    # ... and the next block's a conv behind them (s2): the trunk slab never comes back from HBM
                    nxa = stage.blocks[bi + 1].a
                    self.add(kind="ca", cv=blk.c, cv2=nxa, cv3=blk.branch1, din=db, din3=d, dout=dc, src=c_src, src3=cur, dst=nxt, dst2=a_buf)
                    self.need(a_buf, nxa.out_dims(*dc), nxa.cout)
                    a_done = True
                else:
                    self.add(kind="dual", cv=blk.c, cv2=blk.branch1, din=db, din2=d, dout=dc, src=c_src, src2=cur, dst=nxt, ld=ld)
            elif (not last and not tp and res_src == cur and ld == blk.c.cout and stage.blocks[bi + 1].a.pool_after_bn is None
                  and self.ca_fusable(blk.c, stage.blocks[bi + 1].a, db, dc)):
                # c + residual + ReLU of this block and the a conv of the next one in one launch (s2): the trunk slab is
                # produced in LDS, stored once, and multiplied by the temporal taps without being read back
                nxa = stage.blocks[bi + 1].a
                self.add(kind="ca", cv=blk.c, cv2=nxa, din=db, dout=dc, src=c_src, res=res_src, dst=nxt, dst2=a_buf)
                self.need(a_buf, nxa.out_dims(*dc), nxa.cout)
                a_done = True
            elif tp and next_stage is not None and res_src == cur and ld == blk.c.cout and self._boundary_fusable(blk.c, next_stage, db, dc):
                # s2 -> s3: c + residual + ReLU, the temporal pool and the next stage's first a conv in one launch; the pooled trunk
                # is stored only where the next stage's projection shortcut reads it
                nb = next_stage.blocks[0]
                sub = self._boundary_fusable(blk.c, next_stage, db, dc)
                self.add(kind="cpa", cv=blk.c, cv2=nb.a, din=db, dout=dc, src=c_src, res=res_src, dst=nxt, dst2=a_buf, x_sub=sub, tpool=True)
                self.need(a_buf, nb.a.out_dims(*dstore), nb.a.cout)
                self.boundary = {"sub": sub == 2}
            else:
                self.add(kind="conv", cv=blk.c, din=db, dout=dc, src=c_src, dst=nxt, res=res_src, ld=ld, tpool=tp)
            self.need(nxt, dstore, ld)
            cur, nxt = nxt, cur
            d, c = dstore, blk.c.cout
        return d, c, cur, nxt


class Engine:
    """Op list + activation buffers for one (batch, dtype).  ``run_*`` enqueue the whole forward on
    the current torch stream and return the engine-owned logits / pooled-feature tensors."""

    def __init__(self, spec, weights: PackedWeights, batch: int, device, dims=None):
        self.spec, self.weights, self.batch, self.device = spec, weights, batch, device
        self.dtype = weights.dtype
        self.code = _lib.DTYPE_CODES[self.dtype]
        T, H, W = dims or (spec.num_frames, spec.crop, spec.crop)
        self.in_dims = (T, H, W)
        plan = _Plan(batch, self.code)
        if isinstance(spec, SlowFastSpec):
            self._plan_slowfast(plan, spec, T, H, W)
        elif isinstance(spec, FtcnTTSpec):
            self._plan_ftcn(plan, spec, T, H, W)
        else:
            self._plan_i3d(plan, spec, T, H, W)
        self._materialise(plan)

    # -- plans -----------------------------------------------------------------------------------------
    def _plan_i3d(self, plan: _Plan, spec: NetSpec, T, H, W):
        self.inputs = [("IN", (T, H, W), 1)]                       # (stem-input buffer, dims, temporal stride)
        cur, nxt = "P0", "P1"
        d = spec.stem.out_dims(T, H, W)
        d2 = _pool_out(d, spec.stem_pool)
        fuse_stem_pool = (self.dtype != "f32" and d[2] <= 128 and spec.stem.cout == 64 and
                          _is_pool(spec.stem_pool, (1, 3, 3), (1, 2, 2), (0, 1, 1)))
        self.rgb3 = bool(fuse_stem_pool and spec.stem.conv in getattr(self.weights, "w3", {}))
        if fuse_stem_pool:        # conv + BN + ReLU + max-pool in one launch; the conv output never reaches HBM
            plan.add(kind="stem3_pool" if self.rgb3 else "stem_pool", cv=spec.stem, din=(T, H, W), dout=d, src="IN", dst=cur)
            plan.need(cur, d2, spec.stem.cout)
        else:
            plan.add(kind="stem", cv=spec.stem, din=(T, H, W), dout=d, src="IN", dst=cur); plan.need(cur, d, spec.stem.cout)
            plan.add(kind="pool", pool=spec.stem_pool, ch=spec.stem.cout, din=d, dout=d2, src=cur, dst=nxt)
            plan.need(nxt, d2, spec.stem.cout)
            cur, nxt = nxt, cur
        d = d2
        fuse_tpool = _is_pool(spec.pool_after_s2, (2, 1, 1), (2, 1, 1), (0, 0, 0))
        c = spec.stem.cout
        for si, stage in enumerate(spec.stages):
            d, c, cur, nxt = plan.stage(stage, d, cur, nxt, "A", "B", tpool_last=(fuse_tpool and si == 0),
                                        next_stage=spec.stages[1] if (si == 0 and len(spec.stages) > 1) else None)
            if si == 0 and not plan.entries[-1].get("tpool"):
                # pathway0_pool as its own launch (odd frame counts / other pool shapes)
                d2 = _pool_out(d, spec.pool_after_s2)
                plan.add(kind="pool", pool=spec.pool_after_s2, ch=c, din=d, dout=d2, src=cur, dst=nxt); plan.need(nxt, d2, c)
                cur, nxt = nxt, cur
                d = d2
        hp = tuple(spec.head_pool)
        dh = tuple(di - k + 1 for di, k in zip(d, hp))
        if min(dh) < 1:
            raise ValueError("input %s too small for the head pool %s" % ((T, H, W), hp))
        self.head_dims, self.head_width = dh, c
        plan.add(kind="head", pool=hp, ch=c, din=d, dout=dh, src=cur)

    def _plan_ftcn(self, plan: _Plan, spec: FtcnTTSpec, T, H, W):
        """FTCN-TT (reference i3d_temporal_var_fix_dropout_tt_cfg.py:290-359): temporal stem, s2..s4 with every spatial
        kernel 1x1, per-frame average-pooled tokens, one pre-norm transformer layer, LayerNorm + Linear on the class
        token (time_transformer.py:268-281)."""
        self.inputs = [("IN", (T, H, W), 1)]
        cur, nxt = "P0", "P1"
        d = _pool_out(spec.stem.out_dims(T, H, W), spec.stem.pool_after_bn)
        d2 = _pool_out(d, spec.stem_pool)
        if (self.dtype != "f32" and os.environ.get("AF_FUSE_TSTEM_POOL", "1") == "1"
                and _is_pool(spec.stem_pool, (1, 3, 3), (1, 2, 2), (0, 1, 1))):
            # 16-bit: the stem's own 1x3x3 / stride-2 max-pool rides behind the temporal stem (the half-resolution tensor -
            # 822 MB at 16 clips - is neither written nor read back)
            plan.add(kind="tstem_pool3", cv=spec.stem, din=(T, H, W), dout=d, src="IN", dst=cur); plan.need(cur, d2, spec.stem.cout)
            d = d2
        else:
            plan.add(kind="tstem", cv=spec.stem, din=(T, H, W), dout=d, src="IN", dst=cur); plan.need(cur, d, spec.stem.cout)
            plan.add(kind="pool", pool=spec.stem_pool, ch=spec.stem.cout, din=d, dout=d2, src=cur, dst=nxt); plan.need(nxt, d2, spec.stem.cout)
            cur, nxt, d = nxt, cur, d2
        fuse_tpool = _is_pool(spec.pool_after_s2, (2, 1, 1), (2, 1, 1), (0, 0, 0))
        c = spec.stem.cout
        for si, stage in enumerate(spec.stages):
            d, c, cur, nxt = plan.stage(stage, d, cur, nxt, "A", "B", tpool_last=(fuse_tpool and si == 0))
            if si == 0 and not plan.entries[-1].get("tpool"):
                dp = _pool_out(d, spec.pool_after_s2)
                plan.add(kind="pool", pool=spec.pool_after_s2, ch=c, din=d, dout=dp, src=cur, dst=nxt); plan.need(nxt, dp, c)
                cur, nxt, d = nxt, cur, dp
        # TransformerHead 'time' patches (:133-135): AvgPool3d((1, S, S)) sized from the crop, one token per frame
        if tuple(d[1:]) != tuple(spec.token_pool[1:]) or d[0] != spec.tokens or c != spec.dim:
            raise ValueError("FTCN-TT head expects a (%d,%d,%d)x%d feature map, got %s x %d"
                             % (spec.tokens, spec.token_pool[1], spec.token_pool[2], spec.dim, d, c))
        self.head_dims, self.head_width = (1, 1, 1), spec.dim
        plan.add(kind="tt_head", ch=c, din=d, src=cur)

    def _plan_slowfast(self, plan: _Plan, spec: SlowFastSpec, T, H, W):
        if T % spec.alpha:
            raise ValueError("SlowFast needs a frame count divisible by alpha=%d (got %d)" % (spec.alpha, T))
        Ts = T // spec.alpha
        self.inputs = [("IN_S", (Ts, H, W), spec.alpha), ("IN_F", (T, H, W), 1)]
        fuse_w = [f.cout for f in spec.fuses]                         # channels each lateral adds to the Slow tensor
        # stems (+ max-pool); the Slow pool writes at the widened row stride so the lateral can append its channels
        ds = spec.stems[0].out_dims(Ts, H, W); ds2 = _pool_out(ds, spec.stem_pool)
        df = spec.stems[1].out_dims(T, H, W); df2 = _pool_out(df, spec.stem_pool)
        ld0 = spec.stems[0].cout + fuse_w[0]
        self.rgb3_inputs = set()
        if (spec.stems[0].conv in getattr(self.weights, "w3", {}) and os.environ.get("AF_SLOWFAST_STEM3", "1") == "1"
                and _is_pool(spec.stem_pool, (1, 3, 3), (1, 2, 2), (0, 1, 1)) and ds[2] <= 128):
            # 16-bit: the Slow stem + its max-pool as the K-packed fused stem (3-channel input layout), pooled rows at the widened stride
            self.rgb3_inputs.add("IN_S")
            plan.add(kind="stem3_pool", cv=spec.stems[0], din=(Ts, H, W), dout=ds, src="IN_S", dst="S0", ld=ld0)
            plan.need("S1", ds2, spec.stems[0].cout)
        else:
            plan.add(kind="stem", cv=spec.stems[0], din=(Ts, H, W), dout=ds, src="IN_S", dst="S1"); plan.need("S1", ds, spec.stems[0].cout)
            plan.add(kind="pool", pool=spec.stem_pool, ch=spec.stems[0].cout, din=ds, dout=ds2, src="S1", dst="S0", ld=ld0)
        plan.need("S0", ds2, ld0)
        plan.add(kind="stem", cv=spec.stems[1], din=(T, H, W), dout=df, src="IN_F", dst="F1"); plan.need("F1", df, spec.stems[1].cout)
        plan.add(kind="pool", pool=spec.stem_pool, ch=spec.stems[1].cout, din=df, dout=df2, src="F1", dst="F0")
        plan.need("F0", df2, spec.stems[1].cout)
        scur, snxt, fcur, fnxt = "S0", "S1", "F0", "F1"
        ds, df, cs = ds2, df2, spec.stems[0].cout

        def lateral(fz: ConvSpec, dfast, fbuf, sbuf, c_slow, ld):
            do = fz.out_dims(*dfast)                                   # FuseFastToSlow: conv + BN + ReLU into the Slow rows
            plan.add(kind="conv", cv=fz, din=dfast, dout=do, src=fbuf, dst=sbuf, ld=ld, ch_off=c_slow)
            return do

        if lateral(spec.fuses[0], df, fcur, scur, cs, ld0) != ds:
            raise ValueError("the Fast->Slow lateral does not land on the Slow grid")
        cf = spec.stems[1].cout
        for si, (slow, fast) in enumerate(spec.stages):
            has_fuse = si + 1 < len(spec.fuses)
            ld = slow.blocks[-1].c.cout + (fuse_w[si + 1] if has_fuse else 0)
            ds, cs, scur, snxt = plan.stage(slow, ds, scur, snxt, "A_S", "B_S", last_ld=ld)
            df, cf, fcur, fnxt = plan.stage(fast, df, fcur, fnxt, "A_F", "B_F")
            if has_fuse and lateral(spec.fuses[si + 1], df, fcur, scur, cs, ld) != ds:
                raise ValueError("the Fast->Slow lateral does not land on the Slow grid")
        hs, hf = (tuple(p) for p in spec.head_pools)
        dhs = tuple(di - k + 1 for di, k in zip(ds, hs))
        dhf = tuple(di - k + 1 for di, k in zip(df, hf))
        if min(dhs) < 1 or dhs != dhf:
            raise ValueError("input %s does not fit the SlowFast head pools %s / %s" % ((T, H, W), hs, hf))
        self.head_dims, self.head_width = dhs, cs + cf
        plan.add(kind="avgpool", pool=hs, ch=cs, din=ds, dout=dhs, src=scur, ch_off=0)
        plan.add(kind="avgpool", pool=hf, ch=cf, din=df, dout=dhf, src=fcur, ch_off=cs)
        plan.add(kind="linear")

    # -- buffers + op list ------------------------------------------------------------------------------
    def _materialise(self, plan: _Plan):
        batch, device, weights, spec = self.batch, self.device, self.weights, self.spec
        tdt = _TORCH_DTYPE[self.dtype]
        es = 4 if self.dtype == "f32" else 2
        self.buf = {k: torch.empty(max(v, 8), dtype=tdt, device=device) for k, v in plan.sizes.items()}
        self.rgb3 = getattr(self, "rgb3", False)
        # inputs that feed a K-packed stem take the 3-channel layout (the single input of the i3d engine; SlowFast: the Slow one)
        r3 = set(getattr(self, "rgb3_inputs", ())) | ({self.inputs[0][0]} if self.rgb3 else set())
        self.k_pack_f32 = [_lib.AF_OP_PACK3_F32 if name in r3 else _lib.AF_OP_PACK_F32 for name, _, _ in self.inputs]
        self.k_pack_u8 = _lib.AF_OP_PACK3_U8 if self.rgb3 else _lib.AF_OP_PACK_U8
        for name, dims, _ in self.inputs:                           # padded stem inputs: halos stay zero forever
            nbytes = (lib.af_stem_input_bytes_rgb3 if name in r3 else lib.af_stem_input_bytes)(batch, dims[0], dims[1], dims[2], self.code)
            self.buf[name] = torch.zeros(nbytes // es, dtype=tdt, device=device)
        self.head_positions = self.head_dims[0] * self.head_dims[1] * self.head_dims[2]
        self.pooled = torch.empty((batch, self.head_positions, self.head_width), dtype=torch.float32, device=device)
        self.logits = torch.empty((batch, self.head_positions * spec.num_classes), dtype=torch.float32, device=device)
        # the callers' score epilogue (sigmoid / softmax[:,1], test/af_realtime.py:88-95) rides behind the head's Linear
        self.scores = (torch.empty((batch, self.head_positions), dtype=torch.float32, device=device)
                       if spec.num_classes in (1, 2) else None)

        n_pack = len(self.inputs)
        self.n_ops = n_pack + len(plan.entries) + (self.TT_HEAD_OPS - 1 if isinstance(spec, FtcnTTSpec) else 0)
        self.ops = (Op * self.n_ops)()
        self.op_names: List[str] = []
        self.op_macs: List[int] = []
        self.op_dst: List[Optional[str]] = [name for name, _, _ in self.inputs] + [e.get("dst") for e in plan.entries]
        for i, (name, dims, _) in enumerate(self.inputs):
            pk = self.ops[i]
            pk.kind, pk.tag = self.k_pack_f32[i], TAG_PACK
            pk.conv.n, (pk.conv.t, pk.conv.h, pk.conv.w), pk.conv.dtype = batch, dims, self.code
            pk.out = self.buf[name].data_ptr()
            self.op_names.append("input_pack" if n_pack == 1 else "input_pack_" + name)
            self.op_macs.append(0)

        def fill_conv(cd, cv: ConvSpec, din, dout, relu):
            _fill_conv_desc(cd, batch, self.code, cv, din, dout, relu)

        def fill_pool(pd, din, ch, kernel, stride, pad, dout):
            pd.n, (pd.t, pd.h, pd.w), pd.c = batch, din, ch
            pd.kt, pd.kh, pd.kw = kernel
            pd.st, pd.sh, pd.sw = stride
            pd.pt, pd.ph, pd.pw = pad
            pd.to, pd.ho, pd.wo = dout
            pd.dtype = self.code

        for k, e in enumerate(plan.entries):
            op = self.ops[n_pack + k]
            kind = e["kind"]
            if "src" in e:
                op.in_ = self.buf[e["src"]].data_ptr()
            if "dst" in e:
                op.out = self.buf[e["dst"]].data_ptr() + e.get("ch_off", 0) * es
            if kind == "tt_head":
                self._materialise_tt_head(n_pack + k, e)
                continue
            if kind in ("stem", "stem_pool", "stem3_pool", "conv", "tstem", "tstem_pool3"):
                cv: ConvSpec = e["cv"]
                op.kind = {"stem": _lib.AF_OP_STEM, "stem_pool": _lib.AF_OP_STEM_POOL, "stem3_pool": _lib.AF_OP_STEM3_POOL,
                           "tstem": _lib.AF_OP_TSTEM, "tstem_pool3": _lib.AF_OP_TSTEM_POOL3}.get(kind, _lib.AF_OP_CONV)
                op.tag = TAG_STEM if kind != "conv" else _conv_tag(cv)
                # a, b, stems and laterals carry their own ReLU; c (final_bn) takes the block's add + ReLU; the
                # projection shortcut has neither (resnet_helper.py:311-326, 438-444; video_model_builder.py:136-143)
                fill_conv(op.conv, cv, e["din"], e["dout"], cv.relu or cv.final_bn)
                op.conv.tpool = int(e.get("tpool") or 0)
                op.weight = (weights.w3 if kind == "stem3_pool" else weights.w)[cv.conv].data_ptr()
                op.scale = weights.scale[cv.conv].data_ptr()
                op.shift = weights.shift[cv.conv].data_ptr()
                op.residual = self.buf[e["res"]].data_ptr() if e.get("res") else None
                op.out_ld = e.get("ld", cv.cout)
                self.op_names.append(cv.conv)
                self.op_macs.append(batch * cv.macs(*e["din"]))
            elif kind == "ca":
                cvc, cva = e["cv"], e["cv2"]
                op.kind, op.tag = _lib.AF_OP_CONV_CA, TAG_CONV_CA
                fill_conv(op.conv, cvc, e["din"], e["dout"], True)
                fill_conv(op.conv2, cva, e["dout"], cva.out_dims(*e["dout"]), True)
                op.weight2, op.scale2, op.shift2 = (weights.w[cva.conv].data_ptr(), weights.scale[cva.conv].data_ptr(),
                                                    weights.shift[cva.conv].data_ptr())
                macs = cvc.macs(*e["din"]) + cva.macs(*e["dout"])
                if e.get("cv3") is not None:                 # projection block: folded weights, scale = ones, summed shifts
                    cv1 = e["cv3"]
                    fill_conv(op.conv3, cv1, e["din3"], e["dout"], True)
                    op.weight, op.weight3 = weights.w_folded[cvc.conv].data_ptr(), weights.w_folded[cv1.conv].data_ptr()
                    op.in3 = self.buf[e["src3"]].data_ptr()
                    op.scale, op.shift = weights.ones[cvc.conv].data_ptr(), weights.shift_sum[cvc.conv].data_ptr()
                    op.residual = None
                    macs += cv1.macs(*e["din3"])
                else:
                    op.weight, op.scale, op.shift = (weights.w[cvc.conv].data_ptr(), weights.scale[cvc.conv].data_ptr(),
                                                     weights.shift[cvc.conv].data_ptr())
                    op.residual = self.buf[e["res"]].data_ptr()
                op.aux = self.buf[e["dst2"]].data_ptr()
                op.out_ld = cvc.cout
                self.op_names.append(cvc.conv + ("+branch1" if e.get("cv3") is not None else "") + "->" + cva.conv.split("resnet.")[-1])
                self.op_macs.append(batch * macs)
            elif kind == "cpa":
                cvc, cva = e["cv"], e["cv2"]
                dp = (e["dout"][0] // 2,) + tuple(e["dout"][1:])
                op.kind, op.tag = _lib.AF_OP_CONV_CPA, TAG_CONV_CA
                fill_conv(op.conv, cvc, e["din"], e["dout"], True)
                op.conv.tpool = 1
                fill_conv(op.conv2, cva, dp, cva.out_dims(*dp), True)
                op.weight, op.scale, op.shift = (weights.w[cvc.conv].data_ptr(), weights.scale[cvc.conv].data_ptr(),
                                                 weights.shift[cvc.conv].data_ptr())
                op.weight2, op.scale2, op.shift2 = (weights.w[cva.conv].data_ptr(), weights.scale[cva.conv].data_ptr(),
                                                    weights.shift[cva.conv].data_ptr())
                op.residual = self.buf[e["res"]].data_ptr()
                op.aux = self.buf[e["dst2"]].data_ptr()
                op.out_ld, op.x_sub = cvc.cout, e["x_sub"]
                self.op_names.append(cvc.conv + "+pool->" + cva.conv.split("resnet.")[-1])
                self.op_macs.append(batch * (cvc.macs(*e["din"]) + cva.macs(*dp)))
            elif kind == "abc":
                cva, cvb, cvc = e["cv"], e["cv2"], e["cv3"]
                op.kind, op.tag = _lib.AF_OP_BLOCK_ABC, TAG_BLOCK_ABC
                fill_conv(op.conv, cva, e["din"], e["din"], True)
                fill_conv(op.conv2, cvb, e["din"], e["din"], True)
                fill_conv(op.conv3, cvc, e["din"], e["din"], True)       # final_bn: takes the block's add + ReLU
                op.weight, op.scale, op.shift = (weights.w[cva.conv].data_ptr(), weights.scale[cva.conv].data_ptr(),
                                                 weights.shift[cva.conv].data_ptr())
                op.weight2, op.scale2, op.shift2 = (weights.w[cvb.conv].data_ptr(), weights.scale[cvb.conv].data_ptr(),
                                                    weights.shift[cvb.conv].data_ptr())
                macs = cva.macs(*e["din"]) + cvb.macs(*e["din"]) + cvc.macs(*e["din"])
                if e.get("cv4") is not None:                 # projection block: folded weights, scale = ones, summed shifts
                    cv1 = e["cv4"]
                    fill_conv(op.conv4, cv1, e["din"], e["din"], True)
                    op.weight3, op.weight4 = weights.w_folded[cvc.conv].data_ptr(), weights.w_folded[cv1.conv].data_ptr()
                    op.scale3, op.shift3 = weights.ones[cvc.conv].data_ptr(), weights.shift_sum[cvc.conv].data_ptr()
                    macs += cv1.macs(*e["din"])
                else:
                    op.weight3, op.scale3, op.shift3 = (weights.w[cvc.conv].data_ptr(), weights.scale[cvc.conv].data_ptr(),
                                                        weights.shift[cvc.conv].data_ptr())
                op.out_ld = e.get("ld", cvc.cout)
                self.op_names.append(cva.conv.rsplit(".", 1)[0] + ".a+b+c" + ("+branch1" if e.get("cv4") is not None else ""))
                self.op_macs.append(batch * macs)
            elif kind == "bc":
                cvb, cvc = e["cv"], e["cv2"]
                op.kind, op.tag = _lib.AF_OP_CONV_BC, TAG_CONV_BC
                fill_conv(op.conv, cvb, e["din"], e["dmid"], True)
                fill_conv(op.conv2, cvc, e["dmid"], e["dout"], True)      # final_bn: takes the block's add + ReLU
                op.weight, op.scale, op.shift = (weights.w[cvb.conv].data_ptr(), weights.scale[cvb.conv].data_ptr(),
                                                 weights.shift[cvb.conv].data_ptr())
                op.weight2, op.scale2, op.shift2 = (weights.w[cvc.conv].data_ptr(), weights.scale[cvc.conv].data_ptr(),
                                                    weights.shift[cvc.conv].data_ptr())
                op.residual = self.buf[e["res"]].data_ptr()
                op.out_ld = e.get("ld", cvc.cout)
                self.op_names.append(cvb.conv + "+c")
                self.op_macs.append(batch * (cvb.macs(*e["din"]) + cvc.macs(*e["dmid"])))
            elif kind == "dual":
                cvc, cv1 = e["cv"], e["cv2"]
                op.kind, op.tag = _lib.AF_OP_CONV_DUAL, _conv_tag(cvc)
                fill_conv(op.conv, cvc, e["din"], e["dout"], True)
                fill_conv(op.conv2, cv1, e["din2"], e["dout"], True)
                op.weight = weights.w_folded[cvc.conv].data_ptr()
                op.weight2 = weights.w_folded[cv1.conv].data_ptr()
                op.in2 = self.buf[e["src2"]].data_ptr()
                op.scale = weights.ones[cvc.conv].data_ptr()
                op.shift = weights.shift_sum[cvc.conv].data_ptr()
                op.residual = None
                op.out_ld = e.get("ld", cvc.cout)
                self.op_names.append(cvc.conv + "+branch1")
                self.op_macs.append(batch * (cvc.macs(*e["din"]) + cv1.macs(*e["din2"])))
            elif kind == "pool":
                p = e["pool"]
                op.kind, op.tag = _lib.AF_OP_MAXPOOL, TAG_POOL
                fill_pool(op.pool, e["din"], e["ch"], p.kernel, p.stride, p.pad, e["dout"])
                op.pool.out_ld = e.get("ld", 0)
                self.op_names.append("maxpool_%dx%dx%d" % tuple(p.kernel))
                self.op_macs.append(0)
            elif kind == "head":
                op.kind, op.tag = _lib.AF_OP_HEAD, TAG_HEAD
                fill_pool(op.pool, e["din"], e["ch"], e["pool"], (1, 1, 1), (0, 0, 0), e["dout"])
                op.weight = weights.fc_w.data_ptr()
                op.scale = weights.fc_b.data_ptr()
                op.aux = self.pooled.data_ptr()
                op.out = self.logits.data_ptr()
                op.scores = self.scores.data_ptr() if self.scores is not None else None
                op.num_classes = spec.num_classes
                self.op_names.append("head")
                self.op_macs.append(0)
            elif kind == "avgpool":
                op.kind, op.tag = _lib.AF_OP_AVGPOOL, TAG_HEAD
                fill_pool(op.pool, e["din"], e["ch"], e["pool"], (1, 1, 1), (0, 0, 0), e["dout"])
                op.out = self.pooled.data_ptr() + e["ch_off"] * 4
                op.out_ld = self.head_width
                self.op_names.append("head_avgpool_%d" % e["ch_off"])
                self.op_macs.append(0)
            elif kind == "linear":
                op.kind, op.tag = _lib.AF_OP_LINEAR, TAG_HEAD
                op.in_ = self.pooled.data_ptr()
                op.weight = weights.fc_w.data_ptr()
                op.scale = weights.fc_b.data_ptr()
                op.pool.n, op.pool.c = batch * self.head_positions, self.head_width
                op.num_classes = spec.num_classes
                op.out = self.logits.data_ptr()
                op.scores = self.scores.data_ptr() if self.scores is not None else None
                self.op_names.append("head_linear")
                self.op_macs.append(0)
            else:
                raise KeyError(kind)
        # split-K scratch (small batches, long-K layers): sized by the library, owned by this engine (one per engine =
        # one per device and per stream of work, since an engine enqueues its forward on one stream); consecutive
        # launches of one forward reuse it in stream order
        need = 0
        for i in range(self.n_ops):
            if self.ops[i].kind == _lib.AF_OP_CONV:
                need = max(need, int(lib.af_conv_workspace_bytes(C.byref(self.ops[i].conv))))
        self.workspace = torch.empty(need // 4, dtype=torch.float32, device=device) if need else None
        for i in range(self.n_ops):
            if self.ops[i].kind == _lib.AF_OP_CONV and need:
                self.ops[i].workspace = self.workspace.data_ptr()
                self.ops[i].workspace_bytes = need

    TT_HEAD_OPS = 12

    def _materialise_tt_head(self, first: int, e):
        """The FTCN-TT head as TT_HEAD_OPS ops starting at ops[first] (fp32 buffers of its own)."""
        spec, w, batch, dev = self.spec, self.weights, self.batch, self.device
        n1, dim, inner = spec.tokens + 1, spec.dim, spec.heads * spec.dim_head
        rows = batch * n1
        fb = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self.tokens_pooled = fb(batch, spec.tokens, dim)
        self.fbuf = {"x": fb(rows, dim), "h": fb(rows, dim), "qkv": fb(rows, 3 * inner), "att": fb(rows, inner),
                     "x2": fb(rows, dim), "ff": fb(rows, spec.mlp_dim), "x3": fb(rows, dim)}
        f32code = _lib.DTYPE_CODES["f32"]
        i = [first]

        def nxt(name, tag=TAG_HEAD, macs=0):
            op = self.ops[i[0]]
            i[0] += 1
            op.tag = tag
            self.op_names.append("tt_head." + name)
            self.op_macs.append(macs)
            if len(self.op_dst) < len(self.op_names):
                self.op_dst.append(None)
            return op

        def linear(name, key, src, dst, res=None):
            packed, ones, bias, cin, cout = w.lin[key]
            op = nxt(name, TAG_CONV_1x1x1, rows * cin * cout)
            op.kind = _lib.AF_OP_CONV
            cd = op.conv
            cd.n, cd.t, cd.h, cd.w, cd.cin, cd.cout = 1, 1, 1, rows, cin, cout
            cd.kt = cd.kh = cd.kw = cd.st = cd.sh = cd.sw = 1
            cd.pt = cd.ph = cd.pw = 0
            cd.to, cd.ho, cd.wo = 1, 1, rows
            cd.relu, cd.dtype, cd.tpool = 0, f32code, 0
            op.in_, op.out = self.fbuf[src].data_ptr(), self.fbuf[dst].data_ptr()
            op.weight, op.scale, op.shift = packed.data_ptr(), ones.data_ptr(), bias.data_ptr()
            op.residual = self.fbuf[res].data_ptr() if res else None
            op.out_ld = cout

        def layernorm(name, key, src, dst, n_rows, x_stride, y_stride):
            op = nxt(name)
            op.kind = _lib.AF_OP_LAYERNORM
            op.in_, op.out = src.data_ptr(), dst.data_ptr()
            op.scale, op.shift = w.ln[key][0].data_ptr(), w.ln[key][1].data_ptr()
            op.pool.n, op.pool.c, op.pool.h, op.pool.w = n_rows, dim, x_stride, y_stride

        op = nxt("avgpool_tokens")                                         # (B, T/2, dim) per-frame tokens
        op.kind = _lib.AF_OP_AVGPOOL
        pd = op.pool
        pd.n, (pd.t, pd.h, pd.w), pd.c = batch, e["din"], e["ch"]
        pd.kt, pd.kh, pd.kw = spec.token_pool
        pd.st = pd.sh = pd.sw = 1
        pd.pt = pd.ph = pd.pw = 0
        pd.to, pd.ho, pd.wo = e["din"][0], 1, 1
        pd.dtype = self.code
        op.in_, op.out, op.out_ld = self.buf[e["src"]].data_ptr(), self.tokens_pooled.data_ptr(), dim
        op = nxt("tokens")                                                 # class token + position embedding
        op.kind = _lib.AF_OP_TOKENS
        op.in_, op.weight, op.scale = self.tokens_pooled.data_ptr(), w.cls_token.data_ptr(), w.pos_embedding.data_ptr()
        op.pool.n, op.pool.t, op.pool.c = batch, spec.tokens, dim
        op.out = self.fbuf["x"].data_ptr()
        layernorm("attn_norm", "attn", self.fbuf["x"], self.fbuf["h"], rows, dim, dim)
        linear("to_qkv", "qkv", "h", "qkv")
        op = nxt("attention")
        op.kind = _lib.AF_OP_ATTENTION
        op.in_, op.out = self.fbuf["qkv"].data_ptr(), self.fbuf["att"].data_ptr()
        op.pool.n, op.pool.t, op.pool.h, op.pool.w = batch, n1, spec.heads, spec.dim_head
        linear("to_out", "out", "att", "x2", res="x")                      # Residual(PreNorm(Attention))
        layernorm("ff_norm", "ff", self.fbuf["x2"], self.fbuf["h"], rows, dim, dim)
        linear("ff1", "ff1", "h", "ff")
        op = nxt("gelu")
        op.kind = _lib.AF_OP_GELU
        op.out = self.fbuf["ff"].data_ptr()
        op.pool.n, op.pool.c = rows, spec.mlp_dim
        linear("ff2", "ff2", "ff", "x3", res="x2")                         # Residual(PreNorm(FeedForward))
        # mlp_head on the class token (row 0 of every clip): LayerNorm -> Linear; `pooled` = the Linear's input
        layernorm("head_norm", "head", self.fbuf["x3"], self.pooled, batch, n1 * dim, dim)
        op = nxt("head_linear")
        op.kind = _lib.AF_OP_LINEAR
        op.in_, op.weight, op.scale = self.pooled.data_ptr(), w.fc_w.data_ptr(), w.fc_b.data_ptr()
        op.pool.n, op.pool.c = batch, dim
        op.num_classes = spec.num_classes
        op.out = self.logits.data_ptr()
        op.scores = self.scores.data_ptr() if self.scores is not None else None
        assert i[0] - first == self.TT_HEAD_OPS

    # -- input binding -------------------------------------------------------------------------------------
    def _bind_f32(self, i: int, x: torch.Tensor, tstride: int = 1):
        """Binds network input i to x.  ``tstride`` > 1 takes every tstride-th frame of x (the Slow pathway of a
        single-clip SlowFast call): only the element stride handed to the pack kernel changes, nothing is copied."""
        name, dims, _ = self.inputs[i]
        B, Cc, T, H, W = x.shape
        if (B, Cc, H, W) != (self.batch, 3, dims[1], dims[2]) or T != dims[0] * tstride:
            raise ValueError("expected input (%d,3,%d,%d,%d), got %s"
                             % (self.batch, dims[0] * tstride, dims[1], dims[2], tuple(x.shape)))
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError("inputs must be fp32 HIP tensors")
        pk = self.ops[i]
        pk.kind = self.k_pack_f32[i]
        pk.in_ = x.data_ptr()
        st = list(x.stride())
        st[2] *= tstride
        for k, s in enumerate(st):
            pk.in_strides[k] = s

    def run_f32(self, *xs: torch.Tensor):
        """One (B,3,T,H,W) fp32 device tensor per network input, any strides (the callers' normalised clip; SlowFast:
        slow, fast).  A single tensor given to a two-input network is split by frame striding."""
        if len(xs) == 1 and len(self.inputs) > 1:
            for i, (_, _, ts) in enumerate(self.inputs):
                self._bind_f32(i, xs[0], ts)
        else:
            if len(xs) != len(self.inputs):
                raise ValueError("network takes %d inputs, got %d" % (len(self.inputs), len(xs)))
            for i, x in enumerate(xs):
                self._bind_f32(i, x)
        check(lib.af_run_ops(self.ops, self.n_ops, _stream_ptr(self.device)), "af_run_ops")
        return self.logits, self.pooled

    def run_u8(self, clips: torch.Tensor, mean, std):
        """clips: (B,T,H,W,3) uint8 device tensor in caller layout; normalisation fused into the prologue."""
        assert clips.dtype == torch.uint8 and clips.is_cuda
        if len(self.inputs) != 1:
            raise ValueError("the uint8 prologue is wired for single-input networks")
        B, T, H, W, Cc = clips.shape
        if (B, T, H, W, Cc) != (self.batch,) + self.in_dims + (3,) or not clips.is_contiguous():
            raise ValueError("expected contiguous uint8 clips (%d,%d,%d,%d,3)" % ((self.batch,) + self.in_dims))
        pk = self.ops[0]
        pk.kind = self.k_pack_u8
        pk.in_ = clips.data_ptr()
        for i in range(3):
            pk.mean[i], pk.std_[i] = float(mean[i]), float(std[i])
        check(lib.af_run_ops(self.ops, self.n_ops, _stream_ptr(self.device)), "af_run_ops")
        return self.logits, self.pooled

    def run_prefix(self, n_ops: int):
        """Runs ops[0:n_ops] of the currently bound forward (tests read intermediate activations)."""
        check(lib.af_run_ops(self.ops, n_ops, _stream_ptr(self.device)), "af_run_ops")

    def run_timed(self, first_op: int = 0):
        """Re-runs the currently bound forward with hipEvents around every op; returns ms per op."""
        ms = (C.c_float * self.n_ops)()
        n = self.n_ops - first_op
        ops = C.cast(C.addressof(self.ops) + first_op * C.sizeof(Op), C.POINTER(Op))
        msp = C.cast(C.addressof(ms) + first_op * C.sizeof(C.c_float), C.POINTER(C.c_float))
        check(lib.af_run_ops_timed(ops, n, _stream_ptr(self.device), msp), "af_run_ops_timed")
        return [float(v) for v in ms]

    def activation(self, op_index: int) -> torch.Tensor:
        """Output of op ``op_index`` as an (N,T,H,W,C) view of its buffer (valid until overwritten; ops that write
        into a wider, concatenated row return the full-width rows)."""
        op = self.ops[op_index]
        if op.kind in (_lib.AF_OP_STEM_POOL, _lib.AF_OP_STEM3_POOL, _lib.AF_OP_TSTEM_POOL3):
            shape = (op.conv.n, op.conv.to, (op.conv.ho - 1) // 2 + 1, (op.conv.wo - 1) // 2 + 1, op.conv.cout)
        elif op.kind == _lib.AF_OP_CONV_BC:
            shape = (op.conv2.n, op.conv2.to, op.conv2.ho, op.conv2.wo, op.out_ld or op.conv2.cout)
        elif op.kind == _lib.AF_OP_BLOCK_ABC:
            shape = (op.conv3.n, op.conv3.to, op.conv3.ho, op.conv3.wo, op.out_ld or op.conv3.cout)
        elif op.kind == _lib.AF_OP_CONV_CPA:       # the pooled trunk, or (x_sub == 2) its even (h, w) positions
            shape = (op.conv.n, op.conv.to // 2, op.conv.ho // op.x_sub, op.conv.wo // op.x_sub, op.conv.cout)
        elif op.kind == _lib.AF_OP_CONV_CA:        # the trunk
            shape = (op.conv.n, op.conv.to, op.conv.ho, op.conv.wo, op.conv.cout)
        elif op.kind in (_lib.AF_OP_STEM, _lib.AF_OP_CONV, _lib.AF_OP_CONV_DUAL, _lib.AF_OP_TSTEM):
            ld = op.out_ld or op.conv.cout
            q = 2 if op.conv.tpool == 2 else 1
            shape = (op.conv.n, op.conv.to // 2 if op.conv.tpool == 1 else op.conv.to, op.conv.ho // q, op.conv.wo // q, ld)
        elif op.kind == _lib.AF_OP_MAXPOOL:
            shape = (op.pool.n, op.pool.to, op.pool.ho, op.pool.wo, op.pool.out_ld or op.pool.c)
        else:
            raise ValueError("op %d has no NDHWC output" % op_index)
        numel = 1
        for s in shape:
            numel *= s
        return self.buf[self.op_dst[op_index]][:numel].view(shape)      # rows from channel 0 of the destination buffer

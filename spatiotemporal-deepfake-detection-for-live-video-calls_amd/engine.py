"""Host side of the HIP forward: turns a NetSpec + checkpoint tensors into a flat list of
libafhip ops over caller-owned HBM buffers, and runs it with ONE C call per forward.

What this orchestrates is the reference's ``ResNet.forward``
(altfreezing/slowfast/models/video_model_builder.py:561-578): s1 -> s2 -> pathway0_pool ->
s3 -> s4 -> s5 -> head, with every Conv3d+BatchNorm3d(+add)(+ReLU) group collapsed into one
kernel launch.  PyTorch is used only to own device memory and the stream.
"""
import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ConvDesc, Op, PoolDesc, check, lib
from .arch import BN_EPS, ConvSpec, NetSpec, PoolSpec

_TORCH_DTYPE = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}

# op tags (echoed by the timed runner; used by bench.py to attribute device time to kernel classes)
TAG_PACK, TAG_STEM, TAG_POOL, TAG_HEAD = 0, 1, 2, 3
TAG_CONV_1x1x1, TAG_CONV_Tx1x1, TAG_CONV_1x3x3, TAG_CONV_OTHER = 10, 11, 12, 13
TAG_NAMES = {TAG_PACK: "input_pack", TAG_STEM: "stem_5x7x7", TAG_POOL: "maxpool", TAG_HEAD: "head",
             TAG_CONV_1x1x1: "conv_1x1x1", TAG_CONV_Tx1x1: "conv_3x1x1", TAG_CONV_1x3x3: "conv_1x3x3",
             TAG_CONV_OTHER: "conv_other"}


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


class PackedWeights:
    """Device-resident, kernel-ready copy of a checkpoint: per conv a packed weight in the compute
    dtype plus fp32 BatchNorm scale/shift; fp32 head.  Built with HIP kernels (af_pack.hip)."""

    def __init__(self, spec: NetSpec, state: Dict[str, torch.Tensor], dtype: str, device):
        self.dtype = dtype
        code = _lib.DTYPE_CODES[dtype]
        es = 4 if dtype == "f32" else 2
        st = _stream_ptr(device)
        self.w, self.scale, self.shift = {}, {}, {}
        for cv in spec.convs():
            w = state[cv.conv + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
            assert tuple(w.shape) == cv.weight_shape, (cv.conv, tuple(w.shape), cv.weight_shape)
            bn = [state[cv.bn + s].detach().to(device=device, dtype=torch.float32).contiguous()
                  for s in (".weight", ".bias", ".running_mean", ".running_var")]
            cpad = lib.af_padded_channels(cv.cout)           # kernels read scale/shift over the padded channel tile
            scale = torch.zeros(cpad, dtype=torch.float32, device=device)
            shift = torch.zeros(cpad, dtype=torch.float32, device=device)
            check(lib.af_fold_bn(_ptr(bn[0]), _ptr(bn[1]), _ptr(bn[2]), _ptr(bn[3]), BN_EPS, cv.cout,
                                 _ptr(scale), _ptr(shift), st), "af_fold_bn")
            kt, kh, kw = cv.kernel
            if cv is spec.stem:
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
        # block 0 of every stage: last 1x1x1 + projection shortcut share one accumulator (af_conv3d_dual_bn_act):
        # both weights get their BN scale folded in (fp32, before the rounding), shifts are summed
        self.w_folded, self.shift_sum, self.ones = {}, {}, {}
        for stage in spec.stages:
            for blk in stage.blocks:
                if blk.branch1 is None:
                    continue
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
        self.fc_w = state[spec.head + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
        self.fc_b = state[spec.head + ".bias"].detach().to(device=device, dtype=torch.float32).contiguous()
        torch.cuda.current_stream(device).synchronize()      # sources may be freed by the caller


class Engine:
    """Op list + activation buffers for one (batch, dtype).  ``run_*`` enqueue the whole forward on
    the current torch stream and return the engine-owned logits / pooled-feature tensors."""

    def __init__(self, spec: NetSpec, weights: PackedWeights, batch: int, device, dims=None):
        self.spec, self.weights, self.batch, self.device = spec, weights, batch, device
        self.dtype = weights.dtype
        self.code = _lib.DTYPE_CODES[self.dtype]
        tdt = _TORCH_DTYPE[self.dtype]
        T, H, W = dims or (spec.num_frames, spec.crop, spec.crop)
        self.in_dims = (T, H, W)
        es = 4 if self.dtype == "f32" else 2

        # ---- walk the network once to size the buffers --------------------------------------
        plan = []           # (kind, spec, in_dims, out_dims, in_buf, out_buf, res_buf)
        sizes = {"P0": 0, "P1": 0, "A": 0, "B": 0}

        def need(buf, dims, c):
            sizes[buf] = max(sizes[buf], batch * dims[0] * dims[1] * dims[2] * c)

        def pool_out(dims, p: PoolSpec):
            return tuple((d + 2 * pp - k) // s + 1 for d, k, s, pp in zip(dims, p.kernel, p.stride, p.pad))

        cur, nxt = "P0", "P1"
        d = spec.stem.out_dims(T, H, W)
        d2 = pool_out(d, spec.stem_pool)
        sp_ = spec.stem_pool
        fuse_stem_pool = (self.dtype != "f32" and d[2] <= 128 and spec.stem.cout == 64 and
                          (tuple(sp_.kernel), tuple(sp_.stride), tuple(sp_.pad)) == ((1, 3, 3), (1, 2, 2), (0, 1, 1)))
        if fuse_stem_pool:        # conv + BN + ReLU + max-pool in one launch; the conv output never reaches HBM
            plan.append(("stem_pool", spec.stem, (T, H, W), d, "IN", cur, None)); need(cur, d2, spec.stem.cout)
        else:
            plan.append(("stem", spec.stem, (T, H, W), d, "IN", cur, None)); need(cur, d, spec.stem.cout)
            plan.append(("pool", (spec.stem_pool, spec.stem.cout), d, d2, cur, nxt, None)); need(nxt, d2, spec.stem.cout)
            cur, nxt = nxt, cur
        d, c = d2, spec.stem.cout
        p2 = spec.pool_after_s2
        fuse_tpool = (tuple(p2.kernel), tuple(p2.stride), tuple(p2.pad)) == ((2, 1, 1), (2, 1, 1), (0, 0, 0))
        for si, stage in enumerate(spec.stages):
            for bi, blk in enumerate(stage.blocks):
                res = cur
                da = blk.a.out_dims(*d)
                plan.append(("conv", blk.a, d, da, cur, "A", None)); need("A", da, blk.a.cout)
                db = blk.b.out_dims(*da)
                plan.append(("conv", blk.b, da, db, "A", "B", None)); need("B", db, blk.b.cout)
                dc = blk.c.out_dims(*db)
                # the temporal max-pool after s2 rides in the epilogue of s2's last conv when it can
                tp = fuse_tpool and si == 0 and bi == len(stage.blocks) - 1 and blk.branch1 is None and dc[0] % 2 == 0
                dstore = (dc[0] // 2,) + tuple(dc[1:]) if tp else dc
                if blk.branch1 is not None:      # c conv + projection shortcut in one launch; no shortcut tensor
                    assert blk.branch1.out_dims(*d) == dc
                    plan.append(("dual", (blk.c, blk.branch1, d), db, dc, "B", nxt, cur)); need(nxt, dc, blk.c.cout)
                else:
                    plan.append(("conv_tpool" if tp else "conv", blk.c, db, dc, "B", nxt, res)); need(nxt, dstore, blk.c.cout)
                cur, nxt = nxt, cur
                d, c = dstore, blk.c.cout
            if si == 0 and not tp:
                d2 = pool_out(d, spec.pool_after_s2)
                plan.append(("pool", (spec.pool_after_s2, c), d, d2, cur, nxt, None)); need(nxt, d2, c)
                cur, nxt = nxt, cur
                d = d2
        hp = tuple(spec.head_pool)
        dh = tuple(di - k + 1 for di, k in zip(d, hp))
        if min(dh) < 1:
            raise ValueError("input %s too small for the head pool %s" % ((T, H, W), hp))
        plan.append(("head", (hp, c), d, dh, cur, "LOGITS", None))
        self.head_positions = dh[0] * dh[1] * dh[2]
        self.head_dims = dh

        # ---- buffers (caller-owned HBM, allocated once) ------------------------------------------
        stem_in_bytes = lib.af_stem_input_bytes(batch, T, H, W, self.code)
        self.buf = {k: torch.empty(max(v, 8), dtype=tdt, device=device) for k, v in sizes.items()}
        self.buf["IN"] = torch.zeros(stem_in_bytes // es, dtype=tdt, device=device)   # halos stay zero forever
        self.pooled = torch.empty((batch, self.head_positions, c), dtype=torch.float32, device=device)
        self.logits = torch.empty((batch, self.head_positions * spec.num_classes), dtype=torch.float32, device=device)
        self.buf["LOGITS"] = self.logits

        # ---- op list ----------------------------------------------------------------------------------
        n_ops = len(plan) + 1
        self.ops = (Op * n_ops)()
        self.op_names: List[str] = ["input_pack"]
        self.op_macs: List[int] = [0]
        pk = self.ops[0]
        pk.kind = _lib.AF_OP_PACK_F32
        pk.tag = TAG_PACK
        pk.conv.n, pk.conv.t, pk.conv.h, pk.conv.w, pk.conv.dtype = batch, T, H, W, self.code
        pk.out = self.buf["IN"].data_ptr()
        for i, (kind, sp, din, dout, bi, bo, br) in enumerate(plan, start=1):
            op = self.ops[i]
            op.in_ = self.buf[bi].data_ptr()
            op.out = self.buf[bo].data_ptr()
            if kind in ("stem", "stem_pool", "conv", "conv_tpool"):
                cv: ConvSpec = sp
                op.kind = {"stem": _lib.AF_OP_STEM, "stem_pool": _lib.AF_OP_STEM_POOL}.get(kind, _lib.AF_OP_CONV)
                op.conv.tpool = 1 if kind == "conv_tpool" else 0
                op.tag = TAG_STEM if kind in ("stem", "stem_pool") else _conv_tag(cv)
                cd = op.conv
                cd.n, (cd.t, cd.h, cd.w), cd.cin, cd.cout = batch, din, cv.cin, cv.cout
                cd.kt, cd.kh, cd.kw = cv.kernel
                cd.st, cd.sh, cd.sw = cv.stride
                cd.pt, cd.ph, cd.pw = cv.pad
                cd.to, cd.ho, cd.wo = dout
                # a, b and the stem carry their own ReLU; c (final_bn) takes the block's add + ReLU;
                # the projection shortcut has neither (resnet_helper.py:311-326, 438-444)
                cd.relu = 1 if (cv.relu or cv.final_bn) else 0
                cd.dtype = self.code
                op.weight = weights.w[cv.conv].data_ptr()
                op.scale = weights.scale[cv.conv].data_ptr()
                op.shift = weights.shift[cv.conv].data_ptr()
                op.residual = self.buf[br].data_ptr() if br is not None else None
                op.out_ld = cv.cout
                self.op_names.append(cv.conv)
                self.op_macs.append(batch * cv.macs(*din))
            elif kind == "dual":
                cvc, cv1, din1 = sp
                op.kind, op.tag = _lib.AF_OP_CONV_DUAL, _conv_tag(cvc)
                for cd, cv, dd in ((op.conv, cvc, din), (op.conv2, cv1, din1)):
                    cd.n, (cd.t, cd.h, cd.w), cd.cin, cd.cout = batch, dd, cv.cin, cv.cout
                    cd.kt, cd.kh, cd.kw = cv.kernel
                    cd.st, cd.sh, cd.sw = cv.stride
                    cd.pt, cd.ph, cd.pw = cv.pad
                    cd.to, cd.ho, cd.wo = dout
                    cd.relu, cd.dtype = 1, self.code
                op.weight = weights.w_folded[cvc.conv].data_ptr()
                op.weight2 = weights.w_folded[cv1.conv].data_ptr()
                op.in2 = self.buf[br].data_ptr()
                op.scale = weights.ones[cvc.conv].data_ptr()
                op.shift = weights.shift_sum[cvc.conv].data_ptr()
                op.residual = None
                op.out_ld = cvc.cout
                self.op_names.append(cvc.conv + "+branch1")
                self.op_macs.append(batch * (cvc.macs(*din) + cv1.macs(*din1)))
            elif kind == "pool":
                p, ch = sp
                op.kind, op.tag = _lib.AF_OP_MAXPOOL, TAG_POOL
                pd = op.pool
                pd.n, (pd.t, pd.h, pd.w), pd.c = batch, din, ch
                pd.kt, pd.kh, pd.kw = p.kernel
                pd.st, pd.sh, pd.sw = p.stride
                pd.pt, pd.ph, pd.pw = p.pad
                pd.to, pd.ho, pd.wo = dout
                pd.dtype = self.code
                self.op_names.append("maxpool_%dx%dx%d" % tuple(p.kernel))
                self.op_macs.append(0)
            else:
                hp_, ch = sp
                op.kind, op.tag = _lib.AF_OP_HEAD, TAG_HEAD
                pd = op.pool
                pd.n, (pd.t, pd.h, pd.w), pd.c = batch, din, ch
                pd.kt, pd.kh, pd.kw = hp_
                pd.st = pd.sh = pd.sw = 1
                pd.pt = pd.ph = pd.pw = 0
                pd.to, pd.ho, pd.wo = dout
                pd.dtype = self.code
                op.weight = weights.fc_w.data_ptr()
                op.scale = weights.fc_b.data_ptr()
                op.aux = self.pooled.data_ptr()
                op.num_classes = spec.num_classes
                self.op_names.append("head")
                self.op_macs.append(0)
        self.n_ops = n_ops

    # -- input binding -------------------------------------------------------------------------------------
    def _bind_f32(self, x: torch.Tensor):
        B, Cc, T, H, W = x.shape
        if (B, Cc, T, H, W) != (self.batch, 3) + self.in_dims:
            raise ValueError("expected input (%d,3,%d,%d,%d), got %s" % ((self.batch,) + self.in_dims + (tuple(x.shape),)))
        pk = self.ops[0]
        pk.kind = _lib.AF_OP_PACK_F32
        pk.in_ = x.data_ptr()
        for i, s in enumerate(x.stride()):
            pk.in_strides[i] = s

    def _bind_u8(self, clips: torch.Tensor, mean, std):
        B, T, H, W, Cc = clips.shape
        if (B, T, H, W, Cc) != (self.batch,) + self.in_dims + (3,) or not clips.is_contiguous():
            raise ValueError("expected contiguous uint8 clips (%d,%d,%d,%d,3)" % ((self.batch,) + self.in_dims))
        pk = self.ops[0]
        pk.kind = _lib.AF_OP_PACK_U8
        pk.in_ = clips.data_ptr()
        for i in range(3):
            pk.mean[i], pk.std_[i] = float(mean[i]), float(std[i])

    def run_f32(self, x: torch.Tensor):
        """x: (B,3,T,H,W) fp32 device tensor, any strides (the callers' normalised clip)."""
        assert x.dtype == torch.float32 and x.is_cuda
        self._bind_f32(x)
        check(lib.af_run_ops(self.ops, self.n_ops, _stream_ptr(self.device)), "af_run_ops")
        return self.logits, self.pooled

    def run_u8(self, clips: torch.Tensor, mean, std):
        """clips: (B,T,H,W,3) uint8 device tensor in caller layout; normalisation fused into the prologue."""
        assert clips.dtype == torch.uint8 and clips.is_cuda
        self._bind_u8(clips, mean, std)
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
        """Output of op ``op_index`` as an (N,T,H,W,C) view of its buffer (valid until overwritten)."""
        op = self.ops[op_index]
        if op.kind == _lib.AF_OP_STEM_POOL:
            shape = (op.conv.n, op.conv.to, (op.conv.ho - 1) // 2 + 1, (op.conv.wo - 1) // 2 + 1, op.conv.cout)
        elif op.kind in (_lib.AF_OP_STEM, _lib.AF_OP_CONV, _lib.AF_OP_CONV_DUAL):
            shape = (op.conv.n, op.conv.to // 2 if op.conv.tpool else op.conv.to, op.conv.ho, op.conv.wo, op.conv.cout)
        elif op.kind == _lib.AF_OP_MAXPOOL:
            shape = (op.pool.n, op.pool.to, op.pool.ho, op.pool.wo, op.pool.c)
        else:
            raise ValueError("op %d has no NDHWC output" % op_index)
        numel = 1
        for s in shape:
            numel *= s
        for t in self.buf.values():
            if t.data_ptr() == op.out:
                return t[:numel].view(shape)
        raise KeyError(op_index)

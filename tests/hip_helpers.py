"""Thin test-side wrappers that call libafhip.so through its C ABI (af_mi355x._lib) on torch-owned
device buffers.  Layout conversions (NCDHW <-> NDHWC) are done with torch: they are test scaffolding."""
import ctypes as C

import torch

TORCH_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# stated tolerances of the per-layer and whole-net parity tests, relative to max|reference output|
LAYER_TOL = {"f32": 2e-5, "f16": 4e-3, "bf16": 3e-2}


def lib():
    from af_mi355x import _lib
    return _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_ndhwc(x_ncdhw, dtype):
    return x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().to(device="cuda", dtype=TORCH_DT[dtype])


def to_ncdhw(x_ndhwc):
    return x_ndhwc.permute(0, 4, 1, 2, 3).float().cpu()


def fold_bn(sd, prefix):
    L = lib()
    ts = [sd[prefix + s].float().cuda().contiguous() for s in (".weight", ".bias", ".running_mean", ".running_var")]
    c = ts[0].numel()
    cpad = L.lib.af_padded_channels(c)            # the conv kernels read scale/shift over the padded channel tile
    scale = torch.zeros(cpad, device="cuda")
    shift = torch.zeros(cpad, device="cuda")
    L.check(L.lib.af_fold_bn(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), 1e-5, c, _p(scale), _p(shift), _stream()), "fold_bn")
    return scale, shift


def conv_bn_act(x_ndhwc, w_oidhw, scale, shift, stride, pad, relu, dtype, residual=None, out=None, out_ld=0, tpool=False,
                workspace="auto"):
    """workspace: "auto" = a caller-owned scratch of af_conv_workspace_bytes(d) bytes (the split-K path of small layers),
    None = no workspace (the layer must then run unsplit)."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w, cin = x_ndhwc.shape
    cout, cin2, kt, kh, kw = w_oidhw.shape
    assert cin == cin2
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, cin, cout
    d.kt, d.kh, d.kw = kt, kh, kw
    d.st, d.sh, d.sw = stride
    d.pt, d.ph, d.pw = pad
    d.to, d.ho, d.wo = [(a + 2 * p - k) // s + 1 for a, p, k, s in zip((t, h, w), pad, (kt, kh, kw), stride)]
    d.relu, d.dtype, d.tpool = int(relu), code, int(tpool)
    conv_bn_act.last_variant = L.lib.af_conv_variant(C.byref(d), None)     # which kernel the library picks for this layer
    wsrc = w_oidhw.float().cuda().contiguous()
    nbytes = L.lib.af_packed_conv_weight_bytes(cout, cin, kt, kh, kw, code)
    packed = torch.empty(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_conv_weight(_p(wsrc), cout, cin, kt, kh, kw, code, _p(packed), _stream()), "pack_conv_weight")
    if out is None:
        q = 2 if int(tpool) == 2 else 1
        out = torch.empty((n, d.to // 2 if int(tpool) == 1 else d.to, d.ho // q, d.wo // q, cout), dtype=TORCH_DT[dtype], device="cuda")
    ws_bytes = L.lib.af_conv_workspace_bytes(C.byref(d)) if workspace == "auto" else 0
    ws = torch.empty(max(ws_bytes // 4, 4), dtype=torch.float32, device="cuda") if ws_bytes else None
    conv_bn_act.last_workspace_bytes = ws_bytes
    L.check(L.lib.af_conv3d_bn_act(C.byref(d), _p(x_ndhwc), _p(packed), _p(scale), _p(shift), _p(residual), _p(out),
                                   out_ld, _p(ws), ws_bytes, _stream()), "conv3d_bn_act")
    torch.cuda.current_stream().synchronize()          # ws / packed stay referenced until the launch has run
    return out


def _pack_plain(w_oidhw, dtype):
    L = lib()
    code = L.DTYPE_CODES[dtype]
    cout, cin, kt, kh, kw = w_oidhw.shape
    wsrc = w_oidhw.float().cuda().contiguous()
    nbytes = L.lib.af_packed_conv_weight_bytes(cout, cin, kt, kh, kw, code)
    packed = torch.empty(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_conv_weight(_p(wsrc), cout, cin, kt, kh, kw, code, _p(packed), _stream()), "pack_conv_weight")
    torch.cuda.current_stream().synchronize()
    return packed


def conv_bc(x_ndhwc, wb_oidhw, bn_b, wc_oidhw, bn_c, residual, dtype):
    """relu(bn_c(conv1x1x1(relu(bn_b(conv1x3x3(x))))) + residual) as one af_conv3d_bc_bn_act launch; None if the library
    does not fuse this pair (af_conv_bc_fusable)."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w, cin = x_ndhwc.shape
    cmid, cout = wb_oidhw.shape[0], wc_oidhw.shape[0]
    db, dc = L.ConvDesc(), L.ConvDesc()
    db.n, db.t, db.h, db.w, db.cin, db.cout = n, t, h, w, cin, cmid
    db.kt, db.kh, db.kw, db.st, db.sh, db.sw, db.pt, db.ph, db.pw = 1, 3, 3, 1, 1, 1, 0, 1, 1
    db.to, db.ho, db.wo, db.relu, db.dtype = t, h, w, 1, code
    dc.n, dc.t, dc.h, dc.w, dc.cin, dc.cout = n, t, h, w, cmid, cout
    dc.kt = dc.kh = dc.kw = dc.st = dc.sh = dc.sw = 1
    dc.to, dc.ho, dc.wo, dc.relu, dc.dtype = t, h, w, 1, code
    if not L.lib.af_conv_bc_fusable(C.byref(db), C.byref(dc)):
        return None
    pb, pc = _pack_plain(wb_oidhw, dtype), _pack_plain(wc_oidhw, dtype)
    out = torch.empty((n, t, h, w, cout), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_conv3d_bc_bn_act(C.byref(db), _p(x_ndhwc), _p(pb), _p(bn_b[0]), _p(bn_b[1]), C.byref(dc), _p(pc), _p(bn_c[0]),
                                      _p(bn_c[1]), _p(residual), _p(out), 0, _stream()), "conv3d_bc_bn_act")
    torch.cuda.current_stream().synchronize()
    return out


def _pack_scaled(w_oidhw, scale, dtype):
    """BN scale folded into the packed weight in fp32, before the one rounding (af_pack_conv_weight_scaled)"""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    cout, cin, kt, kh, kw = w_oidhw.shape
    wsrc = w_oidhw.float().cuda().contiguous()
    nbytes = L.lib.af_packed_conv_weight_bytes(cout, cin, kt, kh, kw, code)
    packed = torch.empty(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_conv_weight_scaled(_p(wsrc), _p(scale), cout, cin, kt, kh, kw, code, _p(packed), _stream()), "pack_conv_weight_scaled")
    torch.cuda.current_stream().synchronize()
    return packed


def block_abc(x_ndhwc, wa_oidhw, bn_a, wb_oidhw, bn_b, wc_oidhw, bn_c, dtype, out_ld=0, w1_oidhw=None, bn_1=None):
    """relu(shortcut(x) + bn_c(c(relu(bn_b(b(relu(bn_a(a(x)))))))))  as one af_block_abc_bn_act launch (a: kT x 1 x 1, b: 1x3x3,
    c: 1x1x1; shortcut = x, or bn_1(conv1x1x1_1(x)) with w1 / bn_1: the projection form with folded weights); None if the
    library does not fuse this block (af_block_abc_fusable)."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w, cin = x_ndhwc.shape
    inner, kta, cout = wa_oidhw.shape[0], wa_oidhw.shape[2], wc_oidhw.shape[0]
    da, db, dc, d1 = L.ConvDesc(), L.ConvDesc(), L.ConvDesc(), L.ConvDesc()
    for d, (ci, co, k, p) in zip((da, db, dc, d1), ((cin, inner, (kta, 1, 1), (kta // 2, 0, 0)), (inner, inner, (1, 3, 3), (0, 1, 1)),
                                                    (inner, cout, (1, 1, 1), (0, 0, 0)), (cin, cout, (1, 1, 1), (0, 0, 0)))):
        d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, ci, co
        d.kt, d.kh, d.kw = k
        d.st = d.sh = d.sw = 1
        d.pt, d.ph, d.pw = p
        d.to, d.ho, d.wo, d.relu, d.dtype = t, h, w, 1, code
    proj = w1_oidhw is not None
    if not L.lib.af_block_abc_fusable(C.byref(da), C.byref(db), C.byref(dc), C.byref(d1) if proj else None):
        return None
    pa, pb = _pack_plain(wa_oidhw, dtype), _pack_plain(wb_oidhw, dtype)
    if proj:
        pc, p1 = _pack_scaled(wc_oidhw, bn_c[0], dtype), _pack_scaled(w1_oidhw, bn_1[0], dtype)
        sc3, sh3 = torch.ones_like(bn_c[0]), (bn_c[1] + bn_1[1]).contiguous()
    else:
        pc, p1, sc3, sh3 = _pack_plain(wc_oidhw, dtype), None, bn_c[0], bn_c[1]
    ld = out_ld or cout
    out = torch.full((n, t, h, w, ld), 7.0, dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_block_abc_bn_act(C.byref(da), _p(x_ndhwc), _p(pa), _p(bn_a[0]), _p(bn_a[1]), C.byref(db), _p(pb), _p(bn_b[0]),
                                      _p(bn_b[1]), C.byref(dc), _p(pc), _p(sc3), _p(sh3), C.byref(d1) if proj else None, _p(p1),
                                      _p(out), out_ld, _stream()), "block_abc_bn_act")
    torch.cuda.current_stream().synchronize()
    return out


def conv_ca(b_ndhwc, wc_oidhw, bn_c, res_ndhwc, wa_oidhw, bn_a, dtype, x0_ndhwc=None, w1_oidhw=None, bn_1=None):
    """x = relu(bn_c(conv1x1x1(b)) + res), a_out = relu(bn_a(conv3x1x1(x))) as one af_conv3d_ca_bn_act launch -> (x, a_out);
    with x0 / w1 / bn_1 (a projection block) x = relu(bn_c(c(b)) + bn_1(conv1x1x1_1(x0))) and res must be None.
    None if the library does not fuse this pair (af_conv_ca_fusable)."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w, cmid = b_ndhwc.shape
    ctrunk, cout_a = wc_oidhw.shape[0], wa_oidhw.shape[0]
    dc, da = L.ConvDesc(), L.ConvDesc()
    dc.n, dc.t, dc.h, dc.w, dc.cin, dc.cout = n, t, h, w, cmid, ctrunk
    dc.kt = dc.kh = dc.kw = dc.st = dc.sh = dc.sw = 1
    dc.to, dc.ho, dc.wo, dc.relu, dc.dtype = t, h, w, 1, code
    da.n, da.t, da.h, da.w, da.cin, da.cout = n, t, h, w, ctrunk, cout_a
    da.kt, da.kh, da.kw, da.st, da.sh, da.sw, da.pt, da.ph, da.pw = 3, 1, 1, 1, 1, 1, 1, 0, 0
    da.to, da.ho, da.wo, da.relu, da.dtype = t, h, w, 1, code
    d1 = None
    if x0_ndhwc is not None:
        d1 = L.ConvDesc()
        d1.n, d1.t, d1.h, d1.w, d1.cin, d1.cout = n, t, h, w, x0_ndhwc.shape[-1], ctrunk
        d1.kt = d1.kh = d1.kw = d1.st = d1.sh = d1.sw = 1
        d1.to, d1.ho, d1.wo, d1.relu, d1.dtype = t, h, w, 1, code
    if not L.lib.af_conv_ca_fusable(C.byref(dc), None if d1 is None else C.byref(d1), C.byref(da)):
        return None
    pa = _pack_plain(wa_oidhw, dtype)
    x = torch.empty((n, t, h, w, ctrunk), dtype=TORCH_DT[dtype], device="cuda")
    a_out = torch.empty((n, t, h, w, cout_a), dtype=TORCH_DT[dtype], device="cuda")
    if d1 is None:
        pc, p1, sc, sf = _pack_plain(wc_oidhw, dtype), None, bn_c[0], bn_c[1]
    else:           # both weight sets carry their BN scale, scale = ones, shift = the summed shifts (as the engine does)
        pc, p1 = _pack_scaled(wc_oidhw, bn_c[0], dtype), _pack_scaled(w1_oidhw, bn_1[0], dtype)
        sc, sf = torch.ones(L.lib.af_padded_channels(ctrunk), device="cuda"), (bn_c[1] + bn_1[1]).contiguous()
    L.check(L.lib.af_conv3d_ca_bn_act(C.byref(dc), _p(b_ndhwc), _p(pc), None if d1 is None else C.byref(d1), _p(x0_ndhwc), _p(p1), _p(sc),
                                      _p(sf), _p(res_ndhwc), _p(x), C.byref(da), _p(pa), _p(bn_a[0]), _p(bn_a[1]), _p(a_out), _stream()),
            "conv3d_ca_bn_act")
    torch.cuda.current_stream().synchronize()
    return x, a_out


def conv_cpa(b_ndhwc, wc_oidhw, bn_c, res_ndhwc, wa_oidhw, bn_a, dtype, x_sub):
    """x = relu(bn_c(conv1x1x1(b)) + res), xp = max over frame pairs, a_out = relu(bn_a(conv3x1x1(xp))) as one
    af_conv3d_cpa_bn_act launch -> (xp or its even (h, w) positions, a_out); None if the library does not fuse this pair."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w, cmid = b_ndhwc.shape
    ctrunk, cout_a = wc_oidhw.shape[0], wa_oidhw.shape[0]
    dc, da = L.ConvDesc(), L.ConvDesc()
    dc.n, dc.t, dc.h, dc.w, dc.cin, dc.cout = n, t, h, w, cmid, ctrunk
    dc.kt = dc.kh = dc.kw = dc.st = dc.sh = dc.sw = 1
    dc.to, dc.ho, dc.wo, dc.relu, dc.dtype, dc.tpool = t, h, w, 1, code, 1
    da.n, da.t, da.h, da.w, da.cin, da.cout = n, t // 2, h, w, ctrunk, cout_a
    da.kt, da.kh, da.kw, da.st, da.sh, da.sw, da.pt, da.ph, da.pw = 3, 1, 1, 1, 1, 1, 1, 0, 0
    da.to, da.ho, da.wo, da.relu, da.dtype = t // 2, h, w, 1, code
    if not L.lib.af_conv_cpa_fusable(C.byref(dc), C.byref(da), x_sub):
        return None
    pc, pa = _pack_plain(wc_oidhw, dtype), _pack_plain(wa_oidhw, dtype)
    xs = (n, t // 2, h // 2, w // 2, ctrunk) if x_sub == 2 else (n, t // 2, h, w, ctrunk)
    x = torch.full(xs, 7.0, dtype=TORCH_DT[dtype], device="cuda")
    a_out = torch.full((n, t // 2, h, w, cout_a), 7.0, dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_conv3d_cpa_bn_act(C.byref(dc), _p(b_ndhwc), _p(pc), _p(bn_c[0]), _p(bn_c[1]), _p(res_ndhwc), _p(x), x_sub,
                                       C.byref(da), _p(pa), _p(bn_a[0]), _p(bn_a[1]), _p(a_out), _stream()), "conv3d_cpa_bn_act")
    torch.cuda.current_stream().synchronize()
    return x, a_out


def _pack_scaled(w_oidhw, row_scale, dtype):
    L = lib()
    code = L.DTYPE_CODES[dtype]
    cout, cin, kt, kh, kw = w_oidhw.shape
    wsrc = w_oidhw.float().cuda().contiguous()
    nbytes = L.lib.af_packed_conv_weight_bytes(cout, cin, kt, kh, kw, code)
    packed = torch.empty(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_conv_weight_scaled(_p(wsrc), _p(row_scale), cout, cin, kt, kh, kw, code, _p(packed), _stream()),
            "pack_conv_weight_scaled")
    return packed


def conv_dual(x_ndhwc, w_oidhw, bn, x2_ndhwc, w2_oidhw, bn2, stride2, dtype):
    """relu(bn(conv1x1x1(x)) + bn2(conv1x1x1_strided(x2))) as one af_conv3d_dual_bn_act launch."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    (scale, shift), (scale2, shift2) = bn, bn2
    n, t, h, w, cin = x_ndhwc.shape
    cout = w_oidhw.shape[0]
    d, d2 = L.ConvDesc(), L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, cin, cout
    d.kt = d.kh = d.kw = d.st = d.sh = d.sw = 1
    d.to, d.ho, d.wo, d.relu, d.dtype = t, h, w, 1, code
    n2, t2, h2, w2, cin2 = x2_ndhwc.shape
    d2.n, d2.t, d2.h, d2.w, d2.cin, d2.cout = n2, t2, h2, w2, cin2, cout
    d2.kt = d2.kh = d2.kw = 1
    d2.st, d2.sh, d2.sw = stride2
    d2.to, d2.ho, d2.wo, d2.relu, d2.dtype = t, h, w, 1, code
    out = torch.empty((n, t, h, w, cout), dtype=TORCH_DT[dtype], device="cuda")
    ones = torch.ones(L.lib.af_padded_channels(cout), device="cuda")
    # keep every device buffer referenced until the launch has been enqueued (the caching allocator would
    # otherwise hand the first packed weight's memory to the second)
    pw, pw2, shift_sum = _pack_scaled(w_oidhw, scale, dtype), _pack_scaled(w2_oidhw, scale2, dtype), (shift + shift2).contiguous()
    conv_dual.last_variant = L.lib.af_conv_variant(C.byref(d), C.byref(d2))
    L.check(L.lib.af_conv3d_dual_bn_act(C.byref(d), _p(x_ndhwc), _p(pw), C.byref(d2), _p(x2_ndhwc), _p(pw2), _p(ones),
                                        _p(shift_sum), _p(out), 0, _stream()), "conv3d_dual_bn_act")
    torch.cuda.current_stream().synchronize()
    return out


def pack_input_f32(x_ncdhw_dev, dtype):
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, c, t, h, w = x_ncdhw_dev.shape
    nbytes = L.lib.af_stem_input_bytes(n, t, h, w, code)
    buf = torch.zeros(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    s = x_ncdhw_dev.stride()
    L.check(L.lib.af_pack_input_f32(_p(x_ncdhw_dev), n, t, h, w, s[0], s[1], s[2], s[3], s[4], code, _p(buf), _stream()),
            "pack_input_f32")
    return buf


def pack_input_u8(clips_dev, mean, std, dtype):
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w, _ = clips_dev.shape
    nbytes = L.lib.af_stem_input_bytes(n, t, h, w, code)
    buf = torch.zeros(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    L.check(L.lib.af_pack_input_u8(_p(clips_dev), n, t, h, w, m, s, code, _p(buf), _stream()), "pack_input_u8")
    return buf


def stem_conv(stem_in, dims, w_oidhw, scale, shift, dtype):
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w = dims
    cout, _, kt, kh, kw = w_oidhw.shape
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, 3, cout
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, kh, kw, 1, 2, 2, kt // 2, 3, 3
    d.to, d.ho, d.wo = t, (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    d.relu, d.dtype = 1, code
    wsrc = w_oidhw.float().cuda().contiguous()
    nbytes = L.lib.af_packed_stem_weight_bytes(cout, kt, kh, code)
    packed = torch.empty(nbytes // (4 if dtype == "f32" else 2), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_stem_weight(_p(wsrc), cout, kt, kh, kw, code, _p(packed), _stream()), "pack_stem_weight")
    out = torch.empty((n, d.to, d.ho, d.wo, cout), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_stem_conv_bn_relu(C.byref(d), _p(stem_in), _p(packed), _p(scale), _p(shift), _p(out), _stream()),
            "stem_conv_bn_relu")
    return out


def stem_conv_pool(stem_in, dims, w_oidhw, scale, shift, dtype):
    """conv + BN + ReLU + max-pool [1,3,3]/[1,2,2]/[0,1,1] as one launch (16-bit dtypes)."""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    n, t, h, w = dims
    cout, _, kt, kh, kw = w_oidhw.shape
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, 3, cout
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, kh, kw, 1, 2, 2, kt // 2, 3, 3
    d.to, d.ho, d.wo = t, (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    d.relu, d.dtype = 1, code
    wsrc = w_oidhw.float().cuda().contiguous()
    nbytes = L.lib.af_packed_stem_weight_bytes(cout, kt, kh, code)
    packed = torch.empty(nbytes // 2, dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_stem_weight(_p(wsrc), cout, kt, kh, kw, code, _p(packed), _stream()), "pack_stem_weight")
    out = torch.empty((n, d.to, (d.ho - 1) // 2 + 1, (d.wo - 1) // 2 + 1, cout), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_stem_conv_bn_relu_maxpool(C.byref(d), _p(stem_in), _p(packed), _p(scale), _p(shift), _p(out), _stream()),
            "stem_conv_bn_relu_maxpool")
    torch.cuda.current_stream().synchronize()
    return out


def stem3_conv_pool(x_ncdhw_dev, w_oidhw, scale, shift, dtype, u8=None, mean=None, std=None):
    """the K-packed fused stem (rgb3 input layout): pack (fp32 strided source, or uint8 clips) + conv + BN + ReLU + max-pool"""
    L = lib()
    code = L.DTYPE_CODES[dtype]
    if u8 is None:
        n, _, t, h, w = x_ncdhw_dev.shape
    else:
        n, t, h, w, _ = u8.shape
    nbytes = L.lib.af_stem_input_bytes_rgb3(n, t, h, w, code)
    buf = torch.zeros(nbytes // 2, dtype=TORCH_DT[dtype], device="cuda")
    if u8 is None:
        s = x_ncdhw_dev.stride()
        L.check(L.lib.af_pack_input_f32_rgb3(_p(x_ncdhw_dev), n, t, h, w, s[0], s[1], s[2], s[3], s[4], code, _p(buf), _stream()),
                "pack_input_f32_rgb3")
    else:
        m, sd_ = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        L.check(L.lib.af_pack_input_u8_rgb3(_p(u8), n, t, h, w, m, sd_, code, _p(buf), _stream()), "pack_input_u8_rgb3")
    cout, _, kt, kh, kw = w_oidhw.shape
    d = L.ConvDesc()
    d.n, d.t, d.h, d.w, d.cin, d.cout = n, t, h, w, 3, cout
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, kh, kw, 1, 2, 2, kt // 2, 3, 3
    d.to, d.ho, d.wo = t, (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    d.relu, d.dtype = 1, code
    wsrc = w_oidhw.float().cuda().contiguous()
    packed = torch.empty(L.lib.af_packed_stem_weight_bytes_rgb3(kt, code) // 2, dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_pack_stem_weight_rgb3(_p(wsrc), cout, kt, code, _p(packed), _stream()), "pack_stem_weight_rgb3")
    out = torch.empty((n, d.to, (d.ho - 1) // 2 + 1, (d.wo - 1) // 2 + 1, cout), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_stem_conv_bn_relu_maxpool_rgb3(C.byref(d), _p(buf), _p(packed), _p(scale), _p(shift), _p(out), _stream()),
            "stem_conv_bn_relu_maxpool_rgb3")
    torch.cuda.current_stream().synchronize()
    return out


def maxpool(x_ndhwc, kernel, stride, pad, dtype):
    L = lib()
    n, t, h, w, c = x_ndhwc.shape
    d = L.PoolDesc()
    d.n, d.t, d.h, d.w, d.c = n, t, h, w, c
    d.kt, d.kh, d.kw = kernel
    d.st, d.sh, d.sw = stride
    d.pt, d.ph, d.pw = pad
    d.to, d.ho, d.wo = [(a + 2 * p - k) // s + 1 for a, p, k, s in zip((t, h, w), pad, kernel, stride)]
    d.dtype = L.DTYPE_CODES[dtype]
    out = torch.empty((n, d.to, d.ho, d.wo, c), dtype=TORCH_DT[dtype], device="cuda")
    L.check(L.lib.af_maxpool3d(C.byref(d), _p(x_ndhwc), _p(out), _stream()), "maxpool3d")
    return out


def avgpool_fc(x_ndhwc, pool, fc_w, fc_b, dtype):
    L = lib()
    n, t, h, w, c = x_ndhwc.shape
    d = L.PoolDesc()
    d.n, d.t, d.h, d.w, d.c = n, t, h, w, c
    d.kt, d.kh, d.kw = pool
    d.st = d.sh = d.sw = 1
    d.pt = d.ph = d.pw = 0
    d.to, d.ho, d.wo = t - pool[0] + 1, h - pool[1] + 1, w - pool[2] + 1
    d.dtype = L.DTYPE_CODES[dtype]
    k = fc_w.shape[0]
    pos = d.to * d.ho * d.wo
    pooled = torch.empty((n, pos, c), device="cuda")
    logits = torch.empty((n, pos * k), device="cuda")
    fw, fb = fc_w.float().cuda().contiguous(), fc_b.float().cuda().contiguous()
    L.check(L.lib.af_avgpool_fc(C.byref(d), _p(x_ndhwc), _p(fw), _p(fb), k, _p(pooled), _p(logits), _stream()), "avgpool_fc")
    return pooled, logits

"""ctypes binding of libafhip.so (include/af_hip.h).  There is NO fallback: if the HIP library is
missing or does not export the ABI this file declares, importing fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AF_HIP_LIB") or os.path.join(HERE, "libafhip.so")   # override: kernel experiments

AF_F32, AF_BF16, AF_F16 = 0, 1, 2
(AF_OP_STEM, AF_OP_CONV, AF_OP_MAXPOOL, AF_OP_HEAD, AF_OP_PACK_F32, AF_OP_PACK_U8, AF_OP_CONV_DUAL, AF_OP_STEM_POOL, AF_OP_AVGPOOL,
 AF_OP_LINEAR, AF_OP_TSTEM, AF_OP_TOKENS, AF_OP_LAYERNORM, AF_OP_ATTENTION, AF_OP_GELU, AF_OP_CONV_BC, AF_OP_PACK3_F32,
 AF_OP_PACK3_U8, AF_OP_STEM3_POOL, AF_OP_CONV_CA, AF_OP_BLOCK_ABC, AF_OP_TSTEM_POOL3, AF_OP_CONV_CPA) = range(23)
AF_ABI_VERSION = 4
STEM_PAD_T, STEM_PAD_H, STEM_PAD_W_LEFT, STEM_PAD_W_TOTAL, STEM_CPAD = 2, 3, 3, 8, 4

DTYPE_CODES = {"f32": AF_F32, "bf16": AF_BF16, "f16": AF_F16}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "t", "h", "w", "cin", "cout", "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw",
        "to", "ho", "wo", "relu", "dtype", "tpool")]


class StageRect(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst_offset", C.c_int64), ("src_pitch", C.c_int64), ("rows", C.c_int32), ("row_bytes", C.c_int32)]


class AlignFrame(C.Structure):
    _fields_ = [("offset", C.c_int64), ("ih", C.c_int32), ("iw", C.c_int32), ("x", C.c_int32), ("y", C.c_int32)]


class AlignCrop(C.Structure):
    _fields_ = [("src", C.c_void_p), ("pitch", C.c_int64), ("h", C.c_int32), ("w", C.c_int32), ("x", C.c_int32), ("y", C.c_int32)]


ALIGN_MAX_FRAMES = 64


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "t", "h", "w", "c", "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw", "to", "ho", "wo", "dtype",
        "out_ld")]


class Op(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("out_ld", C.c_int32),
        ("conv", ConvDesc), ("pool", PoolDesc),
        ("in_", C.c_void_p), ("weight", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("residual", C.c_void_p), ("out", C.c_void_p), ("aux", C.c_void_p),
        ("num_classes", C.c_int32), ("tag", C.c_int32),
        ("conv2", ConvDesc), ("in2", C.c_void_p), ("weight2", C.c_void_p),
        ("in_strides", C.c_int64 * 5),
        ("mean", C.c_float * 3), ("std_", C.c_float * 3),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("scores", C.c_void_p),
        ("scale2", C.c_void_p), ("shift2", C.c_void_p),
        ("conv3", ConvDesc), ("in3", C.c_void_p), ("weight3", C.c_void_p),
        ("scale3", C.c_void_p), ("shift3", C.c_void_p),
        ("conv4", ConvDesc), ("weight4", C.c_void_p),
        ("x_sub", C.c_int32), ("reserved0", C.c_int32),
    ]


# name -> (restype, argtypes); tests/test_host_cpu.py::test_c_abi_exports_every_declared_symbol checks this table against
# include/af_hip.h and the built library
ABI = {
    "af_version": (C.c_int, []),
    "af_last_error": (C.c_char_p, []),
    "af_device_count": (C.c_int, []),
    "af_fold_bn": (C.c_int, [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "af_packed_conv_weight_bytes": (C.c_int64, [C.c_int] * 6),
    "af_padded_channels": (C.c_int, [C.c_int]),
    "af_pack_conv_weight": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]),
    "af_pack_conv_weight_scaled": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]),
    "af_packed_stem_weight_bytes": (C.c_int64, [C.c_int] * 4),
    "af_pack_stem_weight": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "af_stem_input_bytes": (C.c_int64, [C.c_int] * 5),
    "af_pack_input_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_int64] * 5 + [C.c_int, C.c_void_p, C.c_void_p]),
    "af_pack_input_u8": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                                  C.c_int, C.c_void_p, C.c_void_p]),
    "af_stem_conv_bn_relu": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6),
    "af_stem_conv_bn_relu_maxpool": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6),
    "af_stem_input_bytes_rgb3": (C.c_int64, [C.c_int] * 5),
    "af_pack_input_f32_rgb3": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_int64] * 5 + [C.c_int, C.c_void_p, C.c_void_p]),
    "af_pack_input_u8_rgb3": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                                       C.c_int, C.c_void_p, C.c_void_p]),
    "af_packed_stem_weight_bytes_rgb3": (C.c_int64, [C.c_int] * 2),
    "af_pack_stem_weight_rgb3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "af_stem_conv_bn_relu_maxpool_rgb3": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6),
    "af_stem_conv_bn_relu_maxpool_rgb3_ld": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "af_conv_workspace_bytes": (C.c_int64, [C.POINTER(ConvDesc)]),
    "af_conv3d_bn_act": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "af_conv3d_dual_bn_act": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.POINTER(ConvDesc)] + [C.c_void_p] * 5
                              + [C.c_int, C.c_void_p]),
    "af_conv_ca_fusable": (C.c_int, [C.POINTER(ConvDesc)] * 3),
    "af_stage_rows_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "af_align_plan_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "af_conv_cpa_fusable": (C.c_int, [C.POINTER(ConvDesc)] * 2 + [C.c_int]),
    "af_conv3d_cpa_bn_act": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(ConvDesc)] + [C.c_void_p] * 5),
    "af_conv3d_ca_bn_act": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.POINTER(ConvDesc)] + [C.c_void_p] * 6
                            + [C.POINTER(ConvDesc)] + [C.c_void_p] * 5),
    "af_conv_bc_fusable": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "af_conv3d_bc_bn_act": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 4 + [C.POINTER(ConvDesc)] + [C.c_void_p] * 5
                            + [C.c_int, C.c_void_p]),
    "af_block_abc_fusable": (C.c_int, [C.POINTER(ConvDesc)] * 4),
    "af_block_abc_bn_act": (C.c_int, ([C.POINTER(ConvDesc)] + [C.c_void_p] * 4) + ([C.POINTER(ConvDesc)] + [C.c_void_p] * 3) * 2
                            + [C.POINTER(ConvDesc), C.c_void_p] + [C.c_void_p, C.c_int, C.c_void_p]),
    "af_conv_variant": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "af_conv_variant_name": (C.c_char_p, [C.c_int]),
    "af_maxpool3d": (C.c_int, [C.POINTER(PoolDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "af_avgpool_fc": (C.c_int, [C.POINTER(PoolDesc)] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3),
    "af_avgpool_fc_scores": (C.c_int, [C.POINTER(PoolDesc)] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4),
    "af_linear_scores": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "af_masked_mean_proj": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int, C.c_void_p]),
    "af_mlp_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "af_avgpool": (C.c_int, [C.POINTER(PoolDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "af_linear": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]),
    "af_pack_tstem_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "af_packed_tstem_weight_bytes": (C.c_longlong, [C.c_int]),
    "af_tstem_conv_bn_pool_relu": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6),
    "af_tstem_conv_bn_pool_relu_maxpool": (C.c_int, [C.POINTER(ConvDesc)] + [C.c_void_p] * 6),
    "af_tokens_assemble": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]),
    "af_layernorm": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                               C.c_longlong, C.c_void_p]),
    "af_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "af_gelu": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "af_dual_branch_weight_floats": (C.c_longlong, [C.c_int] * 4),
    "af_transpose_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "af_dual_branch_encoders": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p,
                                           C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "af_gated_moe": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "af_warp_affine_clip_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int,
                                         C.c_void_p, C.c_void_p]),
    "af_dual_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "af_run_ops": (C.c_int, [C.POINTER(Op), C.c_int, C.c_void_p]),
    "af_run_ops_timed": (C.c_int, [C.POINTER(Op), C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
}


class AfError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libafhip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the HIP path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.af_version() != AF_ABI_VERSION:
        raise ImportError("libafhip.so ABI version %d, binding expects %d" % (lib.af_version(), AF_ABI_VERSION))
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise AfError("%s failed (%d): %s" % (what or "libafhip call", rc, lib.af_last_error().decode()))

// C-ABI glue: version / error text / device probe and the whole-forward op-list runner.
#include "af_common.h"

#include <vector>

namespace af {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int device_cus() {
    static int cus[kMaxDevices] = {};
    const int dev = current_device();
    if (dev >= 0 && cus[dev] > 0) return cus[dev];
    int n = 0, d = dev < 0 ? 0 : dev;
    if (dev < 0) (void)hipGetDevice(&d);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;   // MI355X
    if (dev >= 0) cus[dev] = n;
    return n;
}

static int run_one(const af_op& op, hipStream_t s) {
    switch (op.kind) {
        case AF_OP_STEM:
            return af_stem_conv_bn_relu(&op.conv, op.in, op.weight, op.scale, op.shift, op.out, s);
        case AF_OP_STEM_POOL:
            return af_stem_conv_bn_relu_maxpool(&op.conv, op.in, op.weight, op.scale, op.shift, op.out, s);
        case AF_OP_CONV:
            return af_conv3d_bn_act(&op.conv, op.in, op.weight, op.scale, op.shift, op.residual, op.out, op.out_ld, op.workspace,
                                    op.workspace_bytes, s);
        case AF_OP_CONV_DUAL:
            return af_conv3d_dual_bn_act(&op.conv, op.in, op.weight, &op.conv2, op.in2, op.weight2, op.scale, op.shift,
                                         op.out, op.out_ld, s);
        case AF_OP_CONV_BC:
            return af_conv3d_bc_bn_act(&op.conv, op.in, op.weight, op.scale, op.shift, &op.conv2, op.weight2, op.scale2, op.shift2,
                                       op.residual, op.out, op.out_ld, s);
        case AF_OP_CONV_CPA:    /* out = the pooled trunk (whole or its even positions), aux = the next stage's `a` output */
            return af_conv3d_cpa_bn_act(&op.conv, op.in, op.weight, op.scale, op.shift, op.residual, op.out, op.x_sub, &op.conv2,
                                        op.weight2, op.scale2, op.shift2, op.aux, s);
        case AF_OP_CONV_CA:     /* out = the trunk, aux = the next block's `a` output */
            return af_conv3d_ca_bn_act(&op.conv, op.in, op.weight, op.in3 ? &op.conv3 : nullptr, op.in3, op.weight3, op.scale, op.shift,
                                       op.residual, op.out, &op.conv2, op.weight2, op.scale2, op.shift2, op.aux, s);
        case AF_OP_BLOCK_ABC:   /* in = the trunk: input and residual */
            return af_block_abc_bn_act(&op.conv, op.in, op.weight, op.scale, op.shift, &op.conv2, op.weight2, op.scale2, op.shift2,
                                       &op.conv3, op.weight3, op.scale3, op.shift3, op.weight4 ? &op.conv4 : nullptr, op.weight4,
                                       op.out, op.out_ld, s);
        case AF_OP_MAXPOOL:
            return af_maxpool3d(&op.pool, op.in, op.out, s);
        case AF_OP_HEAD:
            return af_avgpool_fc_scores(&op.pool, op.in, (const float*)op.weight, op.scale, op.num_classes, (float*)op.aux,
                                        (float*)op.out, op.scores, s);
        case AF_OP_AVGPOOL:      /* out = pooled (+ channel offset), out_ld = pooled row stride */
            return af_avgpool(&op.pool, op.in, (float*)op.out, op.out_ld, s);
        case AF_OP_LINEAR:       /* in = x, weight = w, scale = bias, pool.n = rows, pool.c = in_features */
            return af_linear_scores((const float*)op.in, (const float*)op.weight, op.scale, op.pool.n, op.pool.c, op.num_classes,
                                    (float*)op.out, op.scores, s);
        case AF_OP_PACK_F32:
            return af_pack_input_f32((const float*)op.in, op.conv.n, op.conv.t, op.conv.h, op.conv.w, op.in_strides[0],
                                     op.in_strides[1], op.in_strides[2], op.in_strides[3], op.in_strides[4],
                                     op.conv.dtype, op.out, s);
        case AF_OP_PACK_U8:
            return af_pack_input_u8((const uint8_t*)op.in, op.conv.n, op.conv.t, op.conv.h, op.conv.w, op.mean, op.std_,
                                    op.conv.dtype, op.out, s);
        case AF_OP_STEM3_POOL:
            return af_stem_conv_bn_relu_maxpool_rgb3_ld(&op.conv, op.in, op.weight, op.scale, op.shift, op.out, op.out_ld, s);
        case AF_OP_PACK3_F32:
            return af_pack_input_f32_rgb3((const float*)op.in, op.conv.n, op.conv.t, op.conv.h, op.conv.w, op.in_strides[0],
                                          op.in_strides[1], op.in_strides[2], op.in_strides[3], op.in_strides[4],
                                          op.conv.dtype, op.out, s);
        case AF_OP_PACK3_U8:
            return af_pack_input_u8_rgb3((const uint8_t*)op.in, op.conv.n, op.conv.t, op.conv.h, op.conv.w, op.mean, op.std_,
                                         op.conv.dtype, op.out, s);
        case AF_OP_TSTEM:
            return af_tstem_conv_bn_pool_relu(&op.conv, op.in, op.weight, op.scale, op.shift, op.out, s);
        case AF_OP_TSTEM_POOL3:
            return af_tstem_conv_bn_pool_relu_maxpool(&op.conv, op.in, op.weight, op.scale, op.shift, op.out, s);
        case AF_OP_TOKENS:       /* in = pooled, weight = cls token, scale = position embedding; pool.n = clips, pool.t = tokens, pool.c = dim */
            return af_tokens_assemble((const float*)op.in, (const float*)op.weight, op.scale, op.pool.n, op.pool.t, op.pool.c,
                                      (float*)op.out, s);
        case AF_OP_LAYERNORM:    /* in = x, scale = gamma, shift = beta; pool.n = rows, pool.c = dim, pool.h / pool.w = x / y row strides */
            return af_layernorm((const float*)op.in, op.pool.h, op.scale, op.shift, op.pool.n, op.pool.c, 1e-5f, (float*)op.out,
                                op.pool.w, s);
        case AF_OP_ATTENTION:    /* in = qkv; pool.n = clips, pool.t = tokens, pool.h = heads, pool.w = dim_head */
            return af_attention((const float*)op.in, op.pool.n, op.pool.t, op.pool.h, op.pool.w, (float*)op.out, s);
        case AF_OP_GELU:         /* out = x (in place); pool.n * pool.c elements */
            return af_gelu((float*)op.out, (long long)op.pool.n * op.pool.c, s);
        default:
            return set_error(AF_ERR_ARG, "run_ops: unknown op kind %d", op.kind);
    }
}

}  // namespace af

extern "C" int af_version(void) { return AF_ABI_VERSION; }

extern "C" const char* af_last_error(void) { return af::g_err; }

extern "C" int af_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return af::set_error(AF_ERR_NO_DEVICE, "no HIP device visible");
    return n;
}

extern "C" int af_run_ops(const af_op* ops, int n_ops, void* stream) {
    AF_REQUIRE(ops && n_ops > 0, "run_ops: empty op list");
    for (int i = 0; i < n_ops; ++i) {
        int rc = af::run_one(ops[i], (hipStream_t)stream);
        if (rc != AF_OK) return rc;
    }
    return AF_OK;
}

extern "C" int af_run_ops_timed(const af_op* ops, int n_ops, void* stream, float* ms) {
    AF_REQUIRE(ops && n_ops > 0 && ms, "run_ops_timed: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(n_ops + 1);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return af::set_error(AF_ERR_LAUNCH, "run_ops_timed: hipEventCreate failed");
    int rc = AF_OK;
    (void)hipEventRecord(ev[0], s);
    for (int i = 0; i < n_ops && rc == AF_OK; ++i) {
        rc = af::run_one(ops[i], s);
        (void)hipEventRecord(ev[i + 1], s);
    }
    hipError_t e = hipStreamSynchronize(s);
    if (rc == AF_OK && e != hipSuccess) rc = af::set_error(AF_ERR_LAUNCH, "run_ops_timed: %s", hipGetErrorString(e));
    if (rc == AF_OK)
        for (int i = 0; i < n_ops; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    for (auto& ev_ : ev) (void)hipEventDestroy(ev_);
    return rc;
}

/* A plain C99 caller of libafhip.so: what a non-Python host of the reference's classifier path would link against.
 * No torch, no C++: device memory through the HIP runtime's C API, every kernel through include/af_hip.h.
 *   1. ABI version, the thread-local error text of a refused call (a NULL argument never launches anything);
 *   2. nn.BatchNorm3d(eval) folding (af_fold_bn) against the same arithmetic on the host;
 *   3. Conv3d 3x1x1 (64 -> 64, pad [1,0,0]) + BN + residual + ReLU in exact-fp32 mode (af_pack_conv_weight, af_conv3d_bn_act)
 *      against a naive host loop - resnet_helper.py:267-281, 438-444;
 *   4. MaxPool3d [1,3,3] / [1,2,2] / [0,1,1] (af_maxpool3d) against a host loop - stem_helper.py:168-170.
 * Prints "c_abi_smoke OK" and returns 0, or says what differed.  Built and run by tests/test_host_cpu.py (link check, CPU)
 * and tests/test_hip_layers.py (run, GPU). */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "af_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_AF(x) do { int r_ = (x); if (r_ != AF_OK) { printf("af error %d (%s) at %s:%d\n", r_, af_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static unsigned long long rng_state = 88172645463325252ULL;
static float frand(void) {                      /* xorshift: uniform in [-1, 1) */
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

static void* to_device(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    if (host && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    if (af_version() != AF_ABI_VERSION) { printf("ABI %d, header %d\n", af_version(), AF_ABI_VERSION); return 1; }
    if (af_device_count() < 1) { printf("no device\n"); return 1; }

    /* 1. a refused call: error code + text, nothing launched */
    {
        af_conv_desc d; memset(&d, 0, sizeof d);
        int rc = af_conv3d_bn_act(&d, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, 0, NULL);
        if (rc != AF_ERR_ARG || strlen(af_last_error()) == 0) { printf("NULL arguments were not refused (rc %d)\n", rc); return 1; }
    }

    /* 2. BatchNorm folding */
    enum { C = 64 };
    float gamma[C], beta[C], mean[C], var[C], scale[C], shift[C];
    for (int i = 0; i < C; ++i) { gamma[i] = 1.0f + 0.5f * frand(); beta[i] = 0.3f * frand(); mean[i] = 0.2f * frand(); var[i] = 0.5f + 0.4f * (frand() + 1.0f); }
    float *d_gamma = to_device(gamma, sizeof gamma), *d_beta = to_device(beta, sizeof beta), *d_mean = to_device(mean, sizeof mean),
          *d_var = to_device(var, sizeof var), *d_scale = to_device(NULL, sizeof scale), *d_shift = to_device(NULL, sizeof shift);
    if (!d_gamma || !d_beta || !d_mean || !d_var || !d_scale || !d_shift) { printf("hipMalloc failed\n"); return 2; }
    CHECK_AF(af_fold_bn(d_gamma, d_beta, d_mean, d_var, 1e-5f, C, d_scale, d_shift, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(scale, d_scale, sizeof scale, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(shift, d_shift, sizeof shift, hipMemcpyDeviceToHost));
    for (int i = 0; i < C; ++i) {
        const float s = gamma[i] * (1.0f / sqrtf(var[i] + 1e-5f)), b = beta[i] - mean[i] * s;
        if (fabsf(scale[i] - s) > 1e-6f * fabsf(s) + 1e-7f || fabsf(shift[i] - b) > 1e-6f) { printf("fold_bn[%d]: %g %g vs %g %g\n", i, scale[i], shift[i], s, b); return 1; }
    }

    /* 3. conv 3x1x1 + BN + residual + ReLU, fp32 */
    enum { N = 1, T = 4, H = 6, W = 10, KT = 3 };
    const int pos = N * T * H * W;
    float* x = malloc(sizeof(float) * pos * C); float* res = malloc(sizeof(float) * pos * C); float* y = malloc(sizeof(float) * pos * C);
    float* w = malloc(sizeof(float) * C * C * KT);                         /* OIDHW: [cout][cin][kt][1][1] */
    for (int i = 0; i < pos * C; ++i) { x[i] = frand(); res[i] = frand(); }
    for (int i = 0; i < C * C * KT; ++i) w[i] = 0.1f * frand();
    af_conv_desc d; memset(&d, 0, sizeof d);
    d.n = N; d.t = T; d.h = H; d.w = W; d.cin = C; d.cout = C; d.kt = KT; d.kh = 1; d.kw = 1; d.st = d.sh = d.sw = 1; d.pt = 1;
    d.to = T; d.ho = H; d.wo = W; d.relu = 1; d.dtype = AF_F32;
    const int64_t wbytes = af_packed_conv_weight_bytes(C, C, KT, 1, 1, AF_F32);
    if (wbytes <= 0) { printf("packed weight bytes %lld\n", (long long)wbytes); return 1; }
    float* d_w = to_device(w, sizeof(float) * C * C * KT); void* d_wp = to_device(NULL, (size_t)wbytes);
    float *d_x = to_device(x, sizeof(float) * pos * C), *d_res = to_device(res, sizeof(float) * pos * C), *d_y = to_device(NULL, sizeof(float) * pos * C);
    if (!d_w || !d_wp || !d_x || !d_res || !d_y) { printf("hipMalloc failed\n"); return 2; }
    CHECK_AF(af_pack_conv_weight(d_w, C, C, KT, 1, 1, AF_F32, d_wp, NULL));
    CHECK_AF(af_conv3d_bn_act(&d, d_x, d_wp, d_scale, d_shift, d_res, d_y, 0, NULL, 0, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(y, d_y, sizeof(float) * pos * C, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int t = 0; t < T; ++t)
        for (int p = 0; p < H * W; ++p)
            for (int co = 0; co < C; ++co) {
                double acc = 0.0;
                for (int dt = 0; dt < KT; ++dt) {
                    const int ti = t + dt - 1;
                    if (ti < 0 || ti >= T) continue;
                    for (int ci = 0; ci < C; ++ci) acc += (double)w[(co * C + ci) * KT + dt] * x[((size_t)ti * H * W + p) * C + ci];
                }
                double v = acc * scale[co] + shift[co] + res[((size_t)t * H * W + p) * C + co];
                if (v < 0.0) v = 0.0;
                const double e = fabs(v - y[((size_t)t * H * W + p) * C + co]);
                if (e > worst) worst = e;
            }
    if (worst > 2e-4) { printf("conv3x1x1 + BN + residual + ReLU: max |d| %.3e\n", worst); return 1; }

    /* 4. max-pool [1,3,3] / [1,2,2] / [0,1,1] on the conv output */
    af_pool_desc pd; memset(&pd, 0, sizeof pd);
    pd.n = N; pd.t = T; pd.h = H; pd.w = W; pd.c = C; pd.kt = 1; pd.kh = 3; pd.kw = 3; pd.st = 1; pd.sh = 2; pd.sw = 2; pd.pt = 0; pd.ph = 1; pd.pw = 1;
    pd.to = T; pd.ho = (H + 2 - 3) / 2 + 1; pd.wo = (W + 2 - 3) / 2 + 1; pd.dtype = AF_F32;
    const int opos = N * pd.to * pd.ho * pd.wo;
    float* q = malloc(sizeof(float) * opos * C); float* d_q = to_device(NULL, sizeof(float) * opos * C);
    if (!d_q) { printf("hipMalloc failed\n"); return 2; }
    CHECK_AF(af_maxpool3d(&pd, d_y, d_q, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(q, d_q, sizeof(float) * opos * C, hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
        for (int ho = 0; ho < pd.ho; ++ho)
            for (int wo = 0; wo < pd.wo; ++wo)
                for (int c = 0; c < C; ++c) {
                    float m = -INFINITY;
                    for (int dh = 0; dh < 3; ++dh)
                        for (int dw = 0; dw < 3; ++dw) {
                            const int hi = 2 * ho + dh - 1, wi = 2 * wo + dw - 1;
                            if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                            const float v = y[(((size_t)t * H + hi) * W + wi) * C + c];
                            if (v > m) m = v;
                        }
                    if (m != q[(((size_t)t * pd.ho + ho) * pd.wo + wo) * C + c]) { printf("maxpool (%d,%d,%d,%d): %g vs %g\n", t, ho, wo, c, q[(((size_t)t * pd.ho + ho) * pd.wo + wo) * C + c], m); return 1; }
                }
    printf("c_abi_smoke OK (conv max |d| %.2e)\n", worst);
    return 0;
}

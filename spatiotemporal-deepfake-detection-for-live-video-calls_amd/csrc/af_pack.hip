// One-time weight transforms (model load) and the input prologue.  All HBM-bound elementwise work.
#include "af_common.h"

namespace af {

__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               float eps, int c, float* scale, float* shift) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        // same operation order as ATen's eval batch_norm: invstd = 1/sqrt(var+eps); w*invstd; b - mean*that
        float invstd = 1.0f / sqrtf(var[i] + eps);
        float s = gamma[i] * invstd;
        scale[i] = s;
        shift[i] = beta[i] - mean[i] * s;
    }
}

// OIDHW fp32 -> [o_pad][tap][i_pad] in DT: cout padded to 64 rows, cin to the K-step (64 / 32 elements); zeros
template <int DT>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, const float* __restrict__ row_scale, int cout,
                                        int cin, int taps, int coutp, int cinp, typename Elem<DT>::type* __restrict__ out) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over output elements
    long long total = (long long)coutp * taps * cinp;
    if (idx >= total) return;
    int i = (int)(idx % cinp); long long r = idx / cinp;
    int tap = (int)(r % taps); int o = (int)(r / taps);
    float v = 0.f;
    if (o < cout && i < cin) {
        v = w[((long long)o * cin + i) * taps + tap];
        if (row_scale) v *= row_scale[o];           // BN scale folded in fp32, before the one rounding
    }
    out[idx] = Elem<DT>::from_f32(v);
}

// (cout,3,kt,kh,7) fp32 -> [kt][kh][NCH][cout][EPC]: K-row (dt,dh) = 8 pixels x 4 channels, zero padded
template <int DT>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, int cout, int coutp, int kt, int kh, int kw,
                                        typename Elem<DT>::type* __restrict__ out) {
    constexpr int EPC = Elem<DT>::EPC;
    constexpr int NCH = 32 / EPC;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)kt * kh * NCH * coutp * EPC;
    if (idx >= total) return;
    int e = (int)(idx % EPC); long long r = idx / EPC;
    int o = (int)(r % coutp); r /= coutp;
    int ch = (int)(r % NCH); r /= NCH;
    int dh = (int)(r % kh); int dt = (int)(r / kh);
    int k = ch * EPC + e;               // 0..31 within the K-row
    int dw = k >> 2, c = k & 3;
    float v = 0.f;
    if (dw < kw && c < 3 && o < cout) v = w[((((long long)o * 3 + c) * kt + dt) * kh + dh) * kw + dw];
    out[idx] = Elem<DT>::from_f32(v);
}

// interior of the padded stem input; one thread per pixel (4 channels = one 8/16-byte store)
template <int DT, typename Src>
__global__ void pack_input_kernel(Src src, int n, int t, int h, int w, char* __restrict__ out) {
    constexpr int ES = 16 / Elem<DT>::EPC;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n * t * h * w;
    if (idx >= total) return;
    int x = (int)(idx % w); long long r = idx / w;
    int y = (int)(r % h); r /= h;
    int z = (int)(r % t); long long b = r / t;
    f32x4 v;
    v[0] = src(b, 0, z, y, x); v[1] = src(b, 1, z, y, x); v[2] = src(b, 2, z, y, x); v[3] = 0.f;
    const long long Tp = t + 2 * AF_STEM_PAD_T, Hp = h + 2 * AF_STEM_PAD_H, Wp = w + AF_STEM_PAD_W_TOTAL;
    long long o = (((b * Tp + z + AF_STEM_PAD_T) * Hp + y + AF_STEM_PAD_H) * Wp + x + AF_STEM_PAD_W_LEFT) * 4;
    Vec4<DT>::store(out + o * ES, v);
}

struct SrcF32 {
    const float* p; long long sn, sc, st, sh, sw;
    __device__ float operator()(long long b, int c, int z, int y, int x) const {
        return p[b * sn + c * sc + z * st + y * sh + x * sw];
    }
};
struct SrcU8 {
    const uint8_t* p; int t, h, w; float mean[3], stdv[3];
    __device__ float operator()(long long b, int c, int z, int y, int x) const {
        // (float(u8) - mean) / std : the callers' x.sub(mean).div(std) on float32 (af_realtime.py:83)
        float v = (float)p[(((b * t + z) * h + y) * w + x) * 3 + c];
        return (v - mean[c]) / stdv[c];
    }
};

static inline unsigned grid_for(long long total, int block) { return (unsigned)((total + block - 1) / block); }

}  // namespace af

using namespace af;

extern "C" int af_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          int channels, float* scale, float* shift, void* stream) {
    AF_REQUIRE(gamma && beta && mean && var && scale && shift && channels > 0, "fold_bn: bad argument");
    hipLaunchKernelGGL(fold_bn_kernel, dim3(grid_for(channels, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       mean, var, eps, channels, scale, shift);
    AF_CHECK_LAUNCH("fold_bn_kernel");
    return AF_OK;
}

static inline int pad_cout(int cout) { return (cout + 63) / 64 * 64; }
static inline int pad_cin(int cin, int dtype) { const int bk = dtype == AF_F32 ? 32 : 64; return (cin + bk - 1) / bk * bk; }

extern "C" int64_t af_packed_conv_weight_bytes(int cout, int cin, int kt, int kh, int kw, int dtype) {
    if (!dtype_ok(dtype) || cout <= 0 || cin <= 0 || kt <= 0 || kh <= 0 || kw <= 0) return AF_ERR_ARG;
    return (int64_t)pad_cout(cout) * pad_cin(cin, dtype) * kt * kh * kw * dtype_size(dtype);
}

extern "C" int af_padded_channels(int cout) { return cout > 0 ? pad_cout(cout) : AF_ERR_ARG; }

extern "C" int af_pack_conv_weight_scaled(const float* w, const float* row_scale, int cout, int cin, int kt, int kh,
                                          int kw, int dtype, void* packed, void* stream) {
    AF_REQUIRE(w && packed && dtype_ok(dtype) && cout > 0 && cin > 0 && kt > 0 && kh > 0 && kw > 0,
               "pack_conv_weight: bad argument");
    const int taps = kt * kh * kw, coutp = pad_cout(cout), cinp = pad_cin(cin, dtype);
    const long long total = (long long)coutp * cinp * taps;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(total, 256)), b(256);
    if (dtype == AF_F32) hipLaunchKernelGGL((pack_conv_weight_kernel<AF_F32>), g, b, 0, s, w, row_scale, cout, cin, taps, coutp, cinp, (float*)packed);
    else if (dtype == AF_BF16) hipLaunchKernelGGL((pack_conv_weight_kernel<AF_BF16>), g, b, 0, s, w, row_scale, cout, cin, taps, coutp, cinp, (__bf16*)packed);
    else hipLaunchKernelGGL((pack_conv_weight_kernel<AF_F16>), g, b, 0, s, w, row_scale, cout, cin, taps, coutp, cinp, (_Float16*)packed);
    AF_CHECK_LAUNCH("pack_conv_weight_kernel");
    return AF_OK;
}

extern "C" int af_pack_conv_weight(const float* w, int cout, int cin, int kt, int kh, int kw, int dtype, void* packed,
                                   void* stream) {
    return af_pack_conv_weight_scaled(w, nullptr, cout, cin, kt, kh, kw, dtype, packed, stream);
}

static inline int pad_stem_cout(int cout) { return cout <= 16 ? 16 : 64; }     // channel tiles of the stem kernels

extern "C" int64_t af_packed_stem_weight_bytes(int cout, int kt, int kh, int dtype) {
    if (!dtype_ok(dtype) || cout <= 0 || cout > 64 || kt <= 0 || kh <= 0) return AF_ERR_ARG;
    return (int64_t)kt * kh * 32 * pad_stem_cout(cout) * dtype_size(dtype);
}

extern "C" int af_pack_stem_weight(const float* w, int cout, int kt, int kh, int kw, int dtype, void* packed,
                                   void* stream) {
    AF_REQUIRE(w && packed && dtype_ok(dtype) && cout > 0 && cout <= 64 && kt > 0 && kh > 0 && kw > 0 && kw <= 8,
               "pack_stem_weight: bad argument");
    const int coutp = pad_stem_cout(cout);
    const long long total = (long long)kt * kh * 32 * coutp;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(total, 256)), b(256);
    if (dtype == AF_F32) hipLaunchKernelGGL((pack_stem_weight_kernel<AF_F32>), g, b, 0, s, w, cout, coutp, kt, kh, kw, (float*)packed);
    else if (dtype == AF_BF16) hipLaunchKernelGGL((pack_stem_weight_kernel<AF_BF16>), g, b, 0, s, w, cout, coutp, kt, kh, kw, (__bf16*)packed);
    else hipLaunchKernelGGL((pack_stem_weight_kernel<AF_F16>), g, b, 0, s, w, cout, coutp, kt, kh, kw, (_Float16*)packed);
    AF_CHECK_LAUNCH("pack_stem_weight_kernel");
    return AF_OK;
}

extern "C" int64_t af_stem_input_bytes(int n, int t, int h, int w, int dtype) {
    if (!dtype_ok(dtype) || n <= 0 || t <= 0 || h <= 0 || w <= 0) return AF_ERR_ARG;
    return (int64_t)n * (t + 2 * AF_STEM_PAD_T) * (h + 2 * AF_STEM_PAD_H) * (w + AF_STEM_PAD_W_TOTAL) * AF_STEM_CPAD *
           dtype_size(dtype);
}

template <typename Src>
static int launch_pack_input(const Src& src, int n, int t, int h, int w, int dtype, void* out, hipStream_t s) {
    const long long total = (long long)n * t * h * w;
    dim3 g(grid_for(total, 256)), b(256);
    if (dtype == AF_F32) hipLaunchKernelGGL((pack_input_kernel<AF_F32, Src>), g, b, 0, s, src, n, t, h, w, (char*)out);
    else if (dtype == AF_BF16) hipLaunchKernelGGL((pack_input_kernel<AF_BF16, Src>), g, b, 0, s, src, n, t, h, w, (char*)out);
    else hipLaunchKernelGGL((pack_input_kernel<AF_F16, Src>), g, b, 0, s, src, n, t, h, w, (char*)out);
    AF_CHECK_LAUNCH("pack_input_kernel");
    return AF_OK;
}

extern "C" int af_pack_input_f32(const float* x, int n, int t, int h, int w, int64_t stride_n, int64_t stride_c,
                                 int64_t stride_t, int64_t stride_h, int64_t stride_w, int dtype, void* stem_in,
                                 void* stream) {
    AF_REQUIRE(x && stem_in && dtype_ok(dtype) && n > 0 && t > 0 && h > 0 && w > 0, "pack_input_f32: bad argument");
    AF_REQUIRE(aligned16(stem_in), "pack_input_f32: output must be 16-byte aligned");
    SrcF32 src{x, stride_n, stride_c, stride_t, stride_h, stride_w};
    return launch_pack_input(src, n, t, h, w, dtype, stem_in, (hipStream_t)stream);
}

extern "C" int af_pack_input_u8(const uint8_t* clips, int n, int t, int h, int w, const float mean[3],
                                const float std_[3], int dtype, void* stem_in, void* stream) {
    AF_REQUIRE(clips && stem_in && mean && std_ && dtype_ok(dtype) && n > 0 && t > 0 && h > 0 && w > 0,
               "pack_input_u8: bad argument");
    AF_REQUIRE(aligned16(stem_in), "pack_input_u8: output must be 16-byte aligned");
    SrcU8 src;
    src.p = clips; src.t = t; src.h = h; src.w = w;
    for (int i = 0; i < 3; ++i) { src.mean[i] = mean[i]; src.stdv[i] = std_[i]; }
    return launch_pack_input(src, n, t, h, w, dtype, stem_in, (hipStream_t)stream);
}

// NDHWC max-pool and the head (avg-pool + Linear).  HBM-bound; 16-byte vector accesses, one channel
// group per thread so consecutive lanes touch consecutive addresses.
#include "af_common.h"

namespace af {

struct PoolArgs {
    const char* in; char* out;
    int T, H, W, C;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int To, Ho, Wo;
    int out_ld;          // channel stride of the output rows
    long long total;     // N*To*Ho*Wo*(C/V)
};

// nn.MaxPool3d: implicit -inf padding (stem_helper.py:168-170, video_model_builder.py:474-480)
template <int DT>
__global__ void maxpool_kernel(const PoolArgs a) {
    typedef typename Elem<DT>::type elem_t;
    constexpr int V = Elem<DT>::EPC;           // channels per thread (16 bytes)
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.total) return;
    const int cg = a.C / V;
    int c = (int)(idx % cg) * V; long long r = idx / cg;
    int wo = (int)(r % a.Wo); r /= a.Wo;
    int ho = (int)(r % a.Ho); r /= a.Ho;
    int to = (int)(r % a.To); long long n = r / a.To;
    float best[V];
#pragma unroll
    for (int i = 0; i < V; ++i) best[i] = -INFINITY;
    for (int dt = 0; dt < a.kt; ++dt) {
        int ti = to * a.st - a.pt + dt;
        if ((unsigned)ti >= (unsigned)a.T) continue;
        for (int dh = 0; dh < a.kh; ++dh) {
            int hi = ho * a.sh - a.ph + dh;
            if ((unsigned)hi >= (unsigned)a.H) continue;
            for (int dw = 0; dw < a.kw; ++dw) {
                int wi = wo * a.sw - a.pw + dw;
                if ((unsigned)wi >= (unsigned)a.W) continue;
                const long long off = ((((n * a.T + ti) * a.H + hi) * a.W + wi) * a.C + c) * sizeof(elem_t);
                uint4 raw = *reinterpret_cast<const uint4*>(a.in + off);
                const elem_t* e = reinterpret_cast<const elem_t*>(&raw);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float v = Elem<DT>::to_f32(e[i]);
                    best[i] = max_nan(best[i], v);                        // NaN propagates like ATen
                }
            }
        }
    }
    uint4 o;
    elem_t* eo = reinterpret_cast<elem_t*>(&o);
#pragma unroll
    for (int i = 0; i < V; ++i) eo[i] = Elem<DT>::from_f32(best[i]);
    const long long ooff = ((((n * a.To + to) * a.Ho + ho) * a.Wo + wo) * a.out_ld + c) * sizeof(elem_t);
    *reinterpret_cast<uint4*>(a.out + ooff) = o;
}

// nn.AvgPool3d(kernel, stride=1), no padding: pooled[n][pos][c] = sum(window) / count (fp32).
// 64 channels per workgroup as 8 lanes x 16 bytes; the 32 lane groups take interleaved window positions (16-byte
// loads, 32 positions in flight per pass), partial sums meet in LDS.  (One clip = one 784-position window per channel:
// with 2-byte loads and 4 position streams this took 54 us of a 1.07-ms forward.)
template <int DT>
__global__ __launch_bounds__(256) void avgpool_kernel(const char* __restrict__ in, float* __restrict__ pooled, int pooled_ld,
                                                      int T, int H, int W, int C, int kt, int kh, int kw, int To, int Ho,
                                                      int Wo) {
    typedef Elem<DT> E;
    typedef typename E::type elem_t;
    constexpr int EPC = E::EPC, LPC = 64 / EPC;            // lanes covering the 64 channels; 16-bit: 8, fp32: 16
    constexpr int SLOTS = 256 / LPC;
    __shared__ float part[SLOTS][64 + 1];
    const int cg = threadIdx.x % LPC, slot = threadIdx.x / LPC;
    const int c = blockIdx.x * 64 + cg * EPC;
    long long r = blockIdx.y;                   // n*To*Ho*Wo + pos
    int wo = (int)(r % Wo); r /= Wo;
    int ho = (int)(r % Ho); r /= Ho;
    int to = (int)(r % To); long long n = r / To;
    const int win = kt * kh * kw, khw = kh * kw;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    if (c < C) {                                // C % EPC == 0 (host-checked): a lane's chunk is inside or outside
#pragma unroll 2
        for (int p = slot; p < win; p += SLOTS) {
            int dt = p / khw, q = p - dt * khw;
            int dh = q / kw, dw = q - dh * kw;
            const uint4 raw = *reinterpret_cast<const uint4*>(in + ((((n * T + to + dt) * H + ho + dh) * W + wo + dw) * (long long)C + c) * (16 / EPC));
            const elem_t* pe = reinterpret_cast<const elem_t*>(&raw);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += E::to_f32(pe[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) part[slot][cg * EPC + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < SLOTS; ++k) t += part[k][threadIdx.x];
        pooled[(long long)blockIdx.y * pooled_ld + blockIdx.x * 64 + threadIdx.x] = t / (float)win;
    }
}

// nn.Linear on the pooled vector: logits[row][k] = dot(pooled[row], w[k]) + b[k]
// `scores` (optional): the callers' score epilogue (test/af_realtime.py:88-95) - sigmoid(logit) for one class,
// softmax(logits)[1] for two - one value per row
__global__ void fc_kernel(const float* __restrict__ pooled, const float* __restrict__ w, const float* __restrict__ b,
                          int C, int num_classes, float* __restrict__ logits, float* __restrict__ scores) {   // pooled rows are C-contiguous
    __shared__ float red[4];
    const long long row = blockIdx.x;
    float l0 = 0.f, l1 = 0.f;
    for (int k = 0; k < num_classes; ++k) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) s += pooled[row * C + c] * w[(long long)k * C + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float l = red[0] + red[1] + red[2] + red[3] + b[k];
            logits[row * num_classes + k] = l;
            if (k == 0) l0 = l; else if (k == 1) l1 = l;
        }
        __syncthreads();
    }
    if (scores && threadIdx.x == 0)
        scores[row] = num_classes == 1 ? 1.f / (1.f + expf(-l0)) : 1.f / (1.f + expf(l0 - l1));    // softmax([l0, l1])[1]
}

}  // namespace af

using namespace af;

static int check_pool(const af_pool_desc* d, const char* what, bool need_vec) {
    AF_REQUIRE(d && dtype_ok(d->dtype), "%s: bad descriptor", what);
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0 && d->c > 0, "%s: bad dims", what);
    AF_REQUIRE(d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0 && d->pt >= 0 &&
                   d->ph >= 0 && d->pw >= 0, "%s: bad window", what);
    AF_REQUIRE(2 * d->pt <= d->kt && 2 * d->ph <= d->kh && 2 * d->pw <= d->kw, "%s: pad larger than half the window", what);
    const int to = (d->t + 2 * d->pt - d->kt) / d->st + 1, ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1,
              wo = (d->w + 2 * d->pw - d->kw) / d->sw + 1;
    AF_REQUIRE(to == d->to && ho == d->ho && wo == d->wo && to > 0 && ho > 0 && wo > 0,
               "%s: output dims (%d,%d,%d) do not match the descriptor (%d,%d,%d)", what, to, ho, wo, d->to, d->ho, d->wo);
    if (need_vec) {
        const int v = d->dtype == AF_F32 ? 4 : 8;
        AF_REQUIRE(d->c % v == 0, "%s: channels must be a multiple of %d", what, v);
    }
    return AF_OK;
}

extern "C" int af_maxpool3d(const af_pool_desc* d, const void* in, void* out, void* stream) {
    AF_REQUIRE(in && out, "maxpool: null buffer");
    int rc = check_pool(d, "maxpool", true);
    if (rc) return rc;
    AF_REQUIRE(aligned16(in) && aligned16(out), "maxpool: buffers must be 16-byte aligned");
    PoolArgs a;
    a.in = (const char*)in; a.out = (char*)out;
    a.T = d->t; a.H = d->h; a.W = d->w; a.C = d->c;
    a.kt = d->kt; a.kh = d->kh; a.kw = d->kw; a.st = d->st; a.sh = d->sh; a.sw = d->sw;
    a.pt = d->pt; a.ph = d->ph; a.pw = d->pw; a.To = d->to; a.Ho = d->ho; a.Wo = d->wo;
    const int v = d->dtype == AF_F32 ? 4 : 8;
    a.out_ld = d->out_ld ? d->out_ld : d->c;
    AF_REQUIRE(a.out_ld >= d->c && a.out_ld % v == 0, "maxpool: bad out_ld %d", a.out_ld);
    a.total = (long long)d->n * d->to * d->ho * d->wo * (d->c / v);
    const long long blocks = (a.total + 255) / 256;
    AF_REQUIRE(blocks <= 0x7fffffffLL, "maxpool: grid too large");
    dim3 g((unsigned)blocks), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == AF_F32) hipLaunchKernelGGL((maxpool_kernel<AF_F32>), g, b, 0, s, a);
    else if (d->dtype == AF_BF16) hipLaunchKernelGGL((maxpool_kernel<AF_BF16>), g, b, 0, s, a);
    else hipLaunchKernelGGL((maxpool_kernel<AF_F16>), g, b, 0, s, a);
    AF_CHECK_LAUNCH("maxpool_kernel");
    return AF_OK;
}

static int launch_avgpool(const af_pool_desc* d, const void* in, float* pooled, int pooled_ld, hipStream_t s) {
    const long long rows = (long long)d->n * d->to * d->ho * d->wo;
    AF_REQUIRE(rows <= 65535, "avgpool: too many output positions (%lld)", rows);
    dim3 g((d->c + 63) / 64, (unsigned)rows), b(256);
    if (d->dtype == AF_F32)
        hipLaunchKernelGGL((avgpool_kernel<AF_F32>), g, b, 0, s, (const char*)in, pooled, pooled_ld, d->t, d->h, d->w, d->c, d->kt, d->kh, d->kw, d->to, d->ho, d->wo);
    else if (d->dtype == AF_BF16)
        hipLaunchKernelGGL((avgpool_kernel<AF_BF16>), g, b, 0, s, (const char*)in, pooled, pooled_ld, d->t, d->h, d->w, d->c, d->kt, d->kh, d->kw, d->to, d->ho, d->wo);
    else
        hipLaunchKernelGGL((avgpool_kernel<AF_F16>), g, b, 0, s, (const char*)in, pooled, pooled_ld, d->t, d->h, d->w, d->c, d->kt, d->kh, d->kw, d->to, d->ho, d->wo);
    AF_CHECK_LAUNCH("avgpool_kernel");
    return AF_OK;
}

static int check_avgpool(const af_pool_desc* d, const char* what) {
    int rc = check_pool(d, what, true);          // 16-byte loads: channels a multiple of 8 (fp32: 4)
    if (rc) return rc;
    AF_REQUIRE(d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0,
               "%s: AvgPool3d(kernel, stride=1, padding=0) only (head_helper.py:54)", what);
    return AF_OK;
}

extern "C" int af_avgpool(const af_pool_desc* d, const void* in, float* pooled, int pooled_ld, void* stream) {
    AF_REQUIRE(in && pooled, "avgpool: null argument");
    int rc = check_avgpool(d, "avgpool");
    if (rc) return rc;
    AF_REQUIRE(pooled_ld >= d->c, "avgpool: pooled_ld %d < channels %d", pooled_ld, d->c);
    return launch_avgpool(d, in, pooled, pooled_ld, (hipStream_t)stream);
}

extern "C" int af_linear_scores(const float* x, const float* w, const float* b, int rows, int in_features, int out_features,
                                float* y, float* scores, void* stream) {
    AF_REQUIRE(x && w && b && y && rows > 0 && in_features > 0 && out_features > 0, "linear: bad argument");
    AF_REQUIRE(!scores || out_features == 1 || out_features == 2, "linear: scores are defined for 1 (sigmoid) or 2 (softmax[:,1]) classes, got %d",
               out_features);
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, w, b, in_features, out_features, y, scores);
    AF_CHECK_LAUNCH("fc_kernel");
    return AF_OK;
}

extern "C" int af_linear(const float* x, const float* w, const float* b, int rows, int in_features, int out_features,
                         float* y, void* stream) {
    return af_linear_scores(x, w, b, rows, in_features, out_features, y, nullptr, stream);
}

extern "C" int af_avgpool_fc_scores(const af_pool_desc* d, const void* in, const float* fc_w, const float* fc_b,
                                    int num_classes, float* pooled, float* logits, float* scores, void* stream) {
    AF_REQUIRE(in && fc_w && fc_b && pooled && logits && num_classes > 0, "avgpool_fc: null argument");
    int rc = check_avgpool(d, "avgpool_fc");
    if (rc) return rc;
    rc = launch_avgpool(d, in, pooled, d->c, (hipStream_t)stream);
    if (rc) return rc;
    return af_linear_scores(pooled, fc_w, fc_b, d->n * d->to * d->ho * d->wo, d->c, num_classes, logits, scores, stream);
}

extern "C" int af_avgpool_fc(const af_pool_desc* d, const void* in, const float* fc_w, const float* fc_b,
                             int num_classes, float* pooled, float* logits, void* stream) {
    return af_avgpool_fc_scores(d, in, fc_w, fc_b, num_classes, pooled, logits, nullptr, stream);
}

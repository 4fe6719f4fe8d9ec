// dualrun AU / landmark dual encoder (reference dualrun/model/dual_encoder.py:53-198), fp32.
//
// A clip is T <= 16 frames of a few dozen features; the whole BranchEncoder (Linear + LayerNorm, difference /
// high-pass mix, dilated depthwise pyramid + pointwise conv + GELU, positions, `depth` pre-norm transformer layers,
// attention pooling) is ONE launch: a 256-thread workgroup per clip keeps every activation in LDS ([T][256] rows,
// [T][768] for qkv / the MLP hidden) and streams the 2.4 M weights of the branch from L2 in a flat, pre-transposed
// image (W^T, so that consecutive threads read consecutive output columns: coalesced, each weight read once per
// workgroup and reused for all T rows from registers).  One thread = one channel of d_model = 256 for the
// channel-wise steps.  The 16 clips of a batch are 16 workgroups reading the same weights (L2 hits).
// The classification head (LayerNorm -> Linear -> GELU -> Linear on the concatenated clip vectors) is a second
// one-workgroup-per-clip kernel.
#include "af_common.h"

namespace af {

constexpr int DUAL_D = 256;            // d_model the kernels are built for
constexpr int DUAL_THREADS = 1024;     // 16 waves: the weight stream wants many loads in flight (a branch is latency-bound)
constexpr int DUAL_MAXT = 16;
constexpr int DUAL_MAXW = 768;         // widest row kept in LDS (3 * d_model, dim_feedforward)

constexpr int DUAL_MAXB = 4;           // modalities encoded by one launch (blockIdx.y)
struct DualArgs {
    const float* x[DUAL_MAXB];   // [clips][T][din] per branch
    const float* w[DUAL_MAXB];   // flat weight image per branch (layout: dual_branch_weight_floats)
    int din[DUAL_MAXB];
    const int* lengths;    // [clips] valid frames, or null
    const float* pe;       // [T][D] sinusoidal positions
    float* z;              // clip vectors: z[clip * z_ld + branch * D + 0..D)
    int T, depth, heads, ff, z_ld;
    float inv_tau;
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// y[t][n] = act(bias[n] + sum_k x[t][k] * Wt[k][n]) (+ res[t][n]);  x, y, res in LDS, Wt / bias in global memory.
// The 1024 threads cover the N columns once (N = 768) or G = 1024 / N times (N = 256: the K range is split over G thread
// groups whose partial sums meet in `part` [G][TT][N], G = 4 for T <= 8, else 2); ends with the workgroup synchronised.
template <int TT>
__device__ __forceinline__ void linear_rows(const float* xs, int ldx, int K, const float* Wt, const float* bias, int N,
                                            float* ys, int ldy, int T, bool gelu, const float* res, int ldr, float* part) {
    const int tid = threadIdx.x;
    constexpr int GMAX = TT <= 8 ? 4 : 2;                                // `part` holds GMAX x TT x 256 floats (32 KB)
    const int G = N * GMAX <= DUAL_THREADS ? GMAX : (N * 2 <= DUAL_THREADS ? 2 : 1);     // N = 256 -> K groups; N = 768 -> 1
    const int n = tid % N, kg = tid / N;
    const int kspan = ((K / G + 3) / 4) * 4;                             // K slice per group, a multiple of 4
    float acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = 0.f;
    if (kg < G) {
        const int k0 = kg * kspan, k1 = k0 + kspan < K ? k0 + kspan : K;
#pragma unroll 4                                                                         // 16 weight loads in flight per thread
        for (int k = k0; k < k1; k += 4) {                                              // K % 4 == 0 (host-checked)
            const float w0 = Wt[(long long)k * N + n], w1 = Wt[(long long)(k + 1) * N + n],
                        w2 = Wt[(long long)(k + 2) * N + n], w3 = Wt[(long long)(k + 3) * N + n];
#pragma unroll
            for (int t = 0; t < TT; ++t) {                                               // rows >= T hold zeros
                const float4 xv = *reinterpret_cast<const float4*>(xs + t * ldx + k);    // one broadcast ds_read_b128
                acc[t] = fmaf(xv.w, w3, fmaf(xv.z, w2, fmaf(xv.y, w1, fmaf(xv.x, w0, acc[t]))));
            }
        }
    }
    if (G > 1) {
        if (kg < G) {
#pragma unroll
            for (int t = 0; t < TT; ++t) part[(kg * TT + t) * N + n] = acc[t];
        }
        __syncthreads();
        for (int i = tid; i < T * N; i += DUAL_THREADS) {
            const int t = i / N, c = i % N;
            float v = bias[c];
            for (int g = 0; g < G; ++g) v += part[(g * TT + t) * N + c];
            if (gelu) v = gelu_erf(v);
            if (res) v += res[t * ldr + c];
            ys[t * ldy + c] = v;
        }
    } else if (kg < G) {
        const float b = bias[n];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < T) {
                float v = gelu ? gelu_erf(acc[t] + b) : acc[t] + b;
                if (res) v += res[t * ldr + n];
                ys[t * ldy + n] = v;
            }
    }
    __syncthreads();
}

// nn.LayerNorm(256) over the rows of an LDS matrix; wave w takes rows w, w + 4, ...
__device__ __forceinline__ void layernorm_rows(const float* xs, float* ys, int T, const float* gamma, const float* beta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += DUAL_THREADS / 64) {
        float v[4], s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = xs[t * DUAL_D + lane + 64 * i]; s += v[i]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.f / DUAL_D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * (1.f / DUAL_D) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            ys[t * DUAL_D + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
        }
    }
}

template <int TT>
__global__ __launch_bounds__(DUAL_THREADS) void dual_branch_kernel(const DualArgs a) {
    extern __shared__ float sm[];
    constexpr int D = DUAL_D;
    float* h = sm;                          // [TT][D]   the residual stream
    float* y = h + TT * D;                  // [TT][D]   LayerNorm output / scratch
    float* o = y + TT * D;                  // [TT][D]   attention output / input rows
    float* big = o + TT * D;                // [TT][768] qkv, MLP hidden
    float* part = big + TT * DUAL_MAXW;     // [G][TT][D]  partial sums of the K-split linears (32 KB)
    float* sc = part + 32 * D;          // [heads][TT][TT] attention probabilities; pooling weights
    const int tid = threadIdx.x, clip = blockIdx.x, br = blockIdx.y, T = a.T;
    const int din = a.din[br];
    int len = a.lengths ? a.lengths[clip] : T;
    len = len < 1 ? 1 : (len > T ? T : len);                     // a clip without valid frames keeps frame 0 (:162-166)

    // zero the row buffers once: rows >= T feed the (unrolled) row loops with zeros
    for (int i = tid; i < 3 * TT * D + TT * DUAL_MAXW; i += DUAL_THREADS) sm[i] = 0.f;
    __syncthreads();
    const float* xg = a.x[br] + (long long)clip * T * din;
    for (int i = tid; i < T * din; i += DUAL_THREADS) o[(i / din) * D + (i % din)] = xg[i];                // din <= 256 (host-checked)
    __syncthreads();

    const float* w = a.w[br];
    // ---- h = ln_in(proj(x))                                                     (dual_encoder.py:75)
    linear_rows<TT>(o, D, din, w, w + (long long)din * D, D, y, D, T, false, nullptr, 0, part);
    w += (long long)din * D + D;
    layernorm_rows(y, h, T, w, w + D);
    w += 2 * D;
    __syncthreads();
    // ---- first difference + moving-average high-pass mix, depthwise dilated pyramid (thread = channel)   (:77-91)
    if (tid < D) {
        const int c = tid;
        float v[TT], m[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) v[t] = h[t * D + c];                       // zeros beyond T = the convs' zero padding
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float ma = 0.f;                                                      // avg_pool1d(5, pad 2), pad counted
#pragma unroll
            for (int j = -2; j <= 2; ++j) ma += (t + j >= 0 && t + j < TT) ? v[t + j] : 0.f;
            const float delta = t == 0 ? 0.f : v[t] - v[t - 1];
            m[t] = v[t] + 0.5f * delta + 0.5f * (v[t] - ma * 0.2f);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) if (t >= T) m[t] = 0.f;                    // the sequence ends at T (zero padding)
        float p[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) p[t] = m[t];                                // + skip (:89)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int d = 1 << i;
            const float* wd = w + i * 4 * D;
            const float w0 = wd[c * 3 + 0], w1 = wd[c * 3 + 1], w2 = wd[c * 3 + 2], b = wd[3 * D + c];
#pragma unroll
            for (int t = 0; t < TT; ++t)
                p[t] += b + (t - d >= 0 ? w0 * m[t - d] : 0.f) + w1 * m[t] + (t + d < TT ? w2 * m[t + d] : 0.f);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) y[t * D + c] = t < T ? p[t] : 0.f;
    }
    w += 12 * D;
    __syncthreads();
    // ---- pointwise conv (a Linear over channels) + GELU, + positions                                       (:90-94)
    linear_rows<TT>(y, D, D, w, w + D * D, D, h, D, T, true, a.pe, D, part);
    w += D * D + D;

    const int dh = D / a.heads;
    const float qscale = rsqrtf((float)dh);
    for (int l = 0; l < a.depth; ++l) {
        // ---- x = x + self_attn(norm1(x))       (nn.TransformerEncoderLayer, norm_first; key padding mask on the keys)
        layernorm_rows(h, y, T, w, w + D);
        w += 2 * D;
        __syncthreads();
        linear_rows<TT>(y, D, D, w, w + 3 * D * D, 3 * D, big, DUAL_MAXW, T, false, nullptr, 0, part);
        w += 3 * D * D + 3 * D;
        for (int i = tid; i < a.heads * T * T; i += DUAL_THREADS) {
            const int s = i % T, t = (i / T) % T, hd = i / (T * T);
            float dot = 0.f;
            const float* q = big + t * DUAL_MAXW + hd * dh;
            const float* k = big + s * DUAL_MAXW + D + hd * dh;
            for (int d = 0; d < dh; ++d) dot = fmaf(q[d], k[d], dot);
            sc[(hd * TT + t) * TT + s] = s < len ? dot * qscale : -INFINITY;
        }
        __syncthreads();
        for (int i = tid; i < a.heads * T; i += DUAL_THREADS) {
            float* row = sc + (i / T * TT + i % T) * TT;
            float mx = -INFINITY;
            for (int s = 0; s < T; ++s) mx = fmaxf(mx, row[s]);
            float zs = 0.f;
            for (int s = 0; s < T; ++s) { const float e = expf(row[s] - mx); row[s] = e; zs += e; }
            const float inv = 1.f / zs;
            for (int s = 0; s < T; ++s) row[s] *= inv;
        }
        __syncthreads();
        for (int i = tid; i < T * D; i += DUAL_THREADS) {
            const int c = i % D, t = i / D, hd = c / dh;
            {
                float acc = 0.f;
                for (int s = 0; s < T; ++s) acc = fmaf(sc[(hd * TT + t) * TT + s], big[s * DUAL_MAXW + 2 * D + c], acc);
                o[t * D + c] = acc;
            }
        }
        __syncthreads();
        linear_rows<TT>(o, D, D, w, w + D * D, D, h, D, T, false, h, D, part);    // out_proj + residual (in place: an
        w += D * D + D;                                                           // element of h is read and written by one thread)
        // ---- x = x + linear2(gelu(linear1(norm2(x))))
        layernorm_rows(h, y, T, w, w + D);
        w += 2 * D;
        __syncthreads();
        linear_rows<TT>(y, D, D, w, w + (long long)D * a.ff, a.ff, big, DUAL_MAXW, T, true, nullptr, 0, part);
        w += (long long)D * a.ff + a.ff;
        linear_rows<TT>(big, DUAL_MAXW, a.ff, w, w + (long long)a.ff * D, D, h, D, T, false, h, D, part);
        w += (long long)a.ff * D + D;
    }
    // ---- attention pooling: softmax_t(h v / tau) over the valid frames                                     (:30-47)
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int t = wave; t < T; t += DUAL_THREADS / 64) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) s = fmaf(h[t * D + lane + 64 * i], w[lane + 64 * i], s);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) sc[t] = t < len ? s * a.inv_tau : -3.402823466e38f;        // masked_fill(finfo.min)
        }
    }
    __syncthreads();
    if (tid < D) {
        float mx = -INFINITY;
        for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
        float zs = 0.f, acc = 0.f;
        for (int t = 0; t < T; ++t) { const float e = expf(sc[t] - mx); zs += e; acc = fmaf(e, h[t * D + tid], acc); }
        a.z[(long long)clip * a.z_ld + br * D + tid] = acc / zs;
    }
}

// head: LayerNorm(n) -> Linear(n, hidden) -> GELU -> Linear(hidden, 1) [-> sigmoid];  flat weights: gamma [n], beta [n],
// W1^T [n][hidden], b1 [hidden], w2 [hidden], b2
__global__ __launch_bounds__(256) void dual_head_kernel(const float* z, const float* w, int n, int hidden, float* logits,
                                                        float* scores) {
    extern __shared__ float sm[];
    float* x = sm; float* y = sm + n; float* red = y + n;
    const int tid = threadIdx.x, clip = blockIdx.x;
    float s = 0.f;
    for (int i = tid; i < n; i += 256) { x[i] = z[(long long)clip * n + i]; s += x[i]; }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float mean = red[0] / n;
    __syncthreads();
    float q = 0.f;
    for (int i = tid; i < n; i += 256) { const float d = x[i] - mean; q += d * d; }
    red[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float rstd = rsqrtf(red[0] / n + 1e-5f);
    __syncthreads();
    for (int i = tid; i < n; i += 256) x[i] = (x[i] - mean) * rstd * w[i] + w[n + i];
    __syncthreads();
    const float* W1t = w + 2 * n; const float* b1 = W1t + (long long)n * hidden; const float* w2 = b1 + hidden;
    float part = 0.f;
    for (int j = tid; j < hidden; j += 256) {
        // eight independent partial sums (k mod 8): one dependent chain of n loads + FMAs took 170 us for 16 clips
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        int k = 0;
        for (; k + 8 <= n; k += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fmaf(x[k + u], W1t[(long long)(k + u) * hidden + j], acc[u]);
        for (; k < n; ++k) acc[0] = fmaf(x[k], W1t[(long long)k * hidden + j], acc[0]);
        const float sum = b1[j] + (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
        part = fmaf(gelu_erf(sum), w2[j], part);
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
        const float l = red[0] + w2[hidden];
        logits[clip] = l;
        if (scores) scores[clip] = 1.f / (1.f + expf(-l));
    }
}

// AltFreezingRGBEncoder.forward with from_features (dualrun/model/dual_rgb.py:27-44) + rgb_proj (:70, no bias):
//   zv[b] = sum_t V[b][t] * w[b][t],  w = valid / sum_t max(valid[t], 1e-6)  - the reference clamps EACH element before the
//   sum (:42-43), so the denominator is nvalid + (T - nvalid) * 1e-6  (no mask: the plain mean over tv frames),
//   z[b][0..d) = zv[b] @ Wt   (Wt = rgb_proj.weight^T, [vis][d])
// `lengths` counts the valid frames of a mask over `tmask` frames; V has tv == tmask frames, or tv == 1 (broadcast over
// the mask as torch does: the weights then sum to 1, or to 0 for a clip with no valid frame).  One workgroup per clip.
__global__ __launch_bounds__(256) void masked_mean_proj_kernel(const float* v, int tv, int vis, const int* lengths, int tmask,
                                                               const float* wt, int d, float* z, int z_ld) {
    extern __shared__ float sm[];                       // [vis] pooled vector
    const int tid = threadIdx.x, clip = blockIdx.x;
    int nvalid = lengths ? lengths[clip] : tmask;
    nvalid = nvalid < 0 ? 0 : (nvalid > tmask ? tmask : nvalid);
    const float inv = lengths ? 1.f / ((float)nvalid + (float)(tmask - nvalid) * 1e-6f) : 1.f / (float)tv;
    for (int k = tid; k < vis; k += 256) {
        float s = 0.f;
        if (tv == 1) s = lengths ? v[(long long)clip * vis + k] * ((float)nvalid * inv) : v[(long long)clip * vis + k];
        else {
            const int tend = lengths ? nvalid : tv;
            for (int t = 0; t < tend; ++t) s = fmaf(v[((long long)clip * tv + t) * vis + k], inv, s);
        }
        sm[k] = s;
    }
    __syncthreads();
    for (int j = tid; j < d; j += 256) {
        float acc[8];                                   // eight independent partial sums (k mod 8), as in dual_head_kernel
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        int k = 0;
        for (; k + 8 <= vis; k += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fmaf(sm[k + u], wt[(long long)(k + u) * d + j], acc[u]);
        for (; k < vis; ++k) acc[0] = fmaf(sm[k], wt[(long long)k * d + j], acc[0]);
        z[(long long)clip * z_ld + j] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
}

// GatedMoE.forward (dualrun/rgb/engine_rgb.py:369-384): a 3 -> hidden -> 1 gate on (z_rgb, z_dual, |z_rgb - z_dual|) mixes the
// two temperature-scaled probabilities; returns the fused logit and the gate.  One thread per clip.
// w: t_rgb, t_dual, W1 [hidden][3], b1 [hidden], w2 [hidden], b2
__global__ void gated_moe_kernel(const float* z_rgb, const float* z_dual, const float* w, int hidden, int n, float* z, float* g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = z_rgb[i], b = z_dual[i], d = fabsf(a - b);
    const float* W1 = w + 2; const float* b1 = W1 + 3 * hidden; const float* w2 = b1 + hidden;
    float s = w2[hidden];
    for (int j = 0; j < hidden; ++j) {
        const float hj = fmaf(W1[3 * j + 2], d, fmaf(W1[3 * j + 1], b, fmaf(W1[3 * j], a, b1[j])));
        s = fmaf(w2[j], relu_f(hj), s);
    }
    const float gate = 1.f / (1.f + expf(-s));
    const float pr = 1.f / (1.f + expf(-a / fmaxf(w[0], 1.0f)));
    const float pd = 1.f / (1.f + expf(-b / fmaxf(w[1], 0.1f)));
    const float p = gate * pr + (1.f - gate) * pd;
    z[i] = logf((p + 1e-6f) / (1.f - p + 1e-6f));
    g[i] = gate;
}

__global__ void transpose_f32_kernel(const float* src, int rows, int cols, float* dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(long long)c * rows + r] = src[i];
}

}  // namespace af

extern "C" long long af_dual_branch_weight_floats(int din, int d_model, int depth, int ff) {
    const long long D = d_model;
    return (long long)din * D + D + 2 * D + 3 * 4 * D + D * D + D +
           (long long)depth * (2 * D + 3 * D * D + 3 * D + D * D + D + 2 * D + D * ff + ff + (long long)ff * D + D) + D;
}

extern "C" int af_transpose_f32(const float* src, int rows, int cols, float* dst, void* stream) {
    using namespace af;
    AF_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose: bad argument");
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, rows, cols, dst);
    AF_CHECK_LAUNCH("transpose_f32_kernel");
    return AF_OK;
}

extern "C" int af_dual_branch_encoders(int branches, const float* const* x, const float* const* weights, const int* din,
                                       const int* lengths, const float* pe, int clips, int frames, int d_model, int depth,
                                       int heads, int ff, float pool_tau, float* z, int z_ld, void* stream) {
    using namespace af;
    AF_REQUIRE(x && weights && din && pe && z && clips >= 0, "dual_branch_encoders: null argument");
    AF_REQUIRE(branches >= 1 && branches <= DUAL_MAXB, "dual_branch_encoders: 1..%d branches (got %d)", DUAL_MAXB, branches);
    AF_REQUIRE(d_model == DUAL_D, "dual_branch_encoders: built for d_model = %d (got %d)", DUAL_D, d_model);
    AF_REQUIRE(frames >= 1 && frames <= DUAL_MAXT, "dual_branch_encoders: 1..%d frames per clip (got %d)", DUAL_MAXT, frames);
    AF_REQUIRE(depth >= 0 && heads >= 1 && d_model % heads == 0, "dual_branch_encoders: bad depth / heads");
    AF_REQUIRE(ff >= 4 && ff <= DUAL_MAXW && ff % 4 == 0, "dual_branch_encoders: dim_feedforward 4..%d, a multiple of 4 (got %d)", DUAL_MAXW, ff);
    AF_REQUIRE(z_ld >= branches * d_model, "dual_branch_encoders: z_ld < branches * d_model");
    DualArgs a;
    for (int b = 0; b < branches; ++b) {
        AF_REQUIRE(x[b] && weights[b], "dual_branch_encoders: null branch %d", b);
        AF_REQUIRE(din[b] >= 4 && din[b] <= DUAL_D && din[b] % 4 == 0,
                   "dual_branch_encoders: 4..%d input features, a multiple of 4 (branch %d has %d)", DUAL_D, b, din[b]);
        a.x[b] = x[b]; a.w[b] = weights[b]; a.din[b] = din[b];
    }
    if (clips == 0) return AF_OK;
    a.lengths = lengths; a.pe = pe; a.z = z;
    a.T = frames; a.depth = depth; a.heads = heads; a.ff = ff; a.z_ld = z_ld;
    a.inv_tau = 1.0f / (pool_tau > 1e-3f ? pool_tau : 1e-3f);
    const int tt = frames <= 8 ? 8 : 16;
    const int hmax = heads > 4 ? heads : 4;
    const int lds = (3 * tt * DUAL_D + tt * DUAL_MAXW + 32 * DUAL_D + hmax * tt * tt) * 4;
    AF_REQUIRE(lds <= 160 * 1024, "dual_branch_encoders: %d heads do not fit LDS", heads);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = tt == 8 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&dual_branch_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(&dual_branch_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(AF_ERR_LAUNCH, "dual_branch_encoders: hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (tt == 8) hipLaunchKernelGGL((dual_branch_kernel<8>), dim3(clips, branches), dim3(DUAL_THREADS), lds, s, a);
    else hipLaunchKernelGGL((dual_branch_kernel<16>), dim3(clips, branches), dim3(DUAL_THREADS), lds, s, a);
    AF_CHECK_LAUNCH("dual_branch_kernel");
    return AF_OK;
}

extern "C" int af_mlp_head(const float* z, const float* weights, int clips, int n, int hidden, float* logits, float* scores,
                           void* stream) {
    using namespace af;
    AF_REQUIRE(z && weights && logits && clips >= 0 && n >= 1 && n <= 4096 && hidden >= 1, "mlp_head: bad argument");
    if (clips == 0) return AF_OK;
    hipLaunchKernelGGL(dual_head_kernel, dim3(clips), dim3(256), (2 * n + 256) * 4, (hipStream_t)stream, z, weights, n, hidden,
                       logits, scores);
    AF_CHECK_LAUNCH("dual_head_kernel");
    return AF_OK;
}

extern "C" int af_dual_head(const float* z, const float* weights, int clips, int n, float* logits, void* stream) {
    return af_mlp_head(z, weights, clips, n, n, logits, nullptr, stream);
}

extern "C" int af_masked_mean_proj(const float* v, int clips, int tv, int vis, const int* lengths, int tmask, const float* wt,
                                   int d, float* z, int z_ld, void* stream) {
    using namespace af;
    AF_REQUIRE(v && wt && z && clips >= 0 && tv >= 1 && vis >= 1 && vis <= 16384 && d >= 1 && z_ld >= d, "masked_mean_proj: bad argument");
    AF_REQUIRE(tmask >= 1 && (tv == tmask || tv == 1), "masked_mean_proj: V has %d frames, the mask %d (equal, or 1 to broadcast)", tv, tmask);
    if (clips == 0) return AF_OK;
    hipLaunchKernelGGL(masked_mean_proj_kernel, dim3(clips), dim3(256), vis * 4, (hipStream_t)stream, v, tv, vis, lengths, tmask, wt,
                       d, z, z_ld);
    AF_CHECK_LAUNCH("masked_mean_proj_kernel");
    return AF_OK;
}

extern "C" int af_gated_moe(const float* z_rgb, const float* z_dual, const float* weights, int hidden, int n, float* z, float* gate,
                            void* stream) {
    using namespace af;
    AF_REQUIRE(z_rgb && z_dual && weights && z && gate && hidden >= 1 && n >= 0, "gated_moe: bad argument");
    if (n == 0) return AF_OK;
    hipLaunchKernelGGL(gated_moe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, z_rgb, z_dual, weights, hidden, n, z, gate);
    AF_CHECK_LAUNCH("gated_moe_kernel");
    return AF_OK;
}

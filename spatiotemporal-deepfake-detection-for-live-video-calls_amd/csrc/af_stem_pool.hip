// Stem with the max-pool fused: Conv3d(3->64,[kt,7,7],s[1,2,2]) + BN + ReLU + MaxPool3d([1,3,3],s[1,2,2],p[0,1,1])
// (reference altfreezing/slowfast/models/stem_helper.py:156-178) as ONE launch, 16-bit operands.
//
// A persistent workgroup owns one output frame (n, t) - or, when there are fewer frames than CUs (small batches: the
// live-call case is ONE clip = 32 frames), one band of its pooled rows - and streams the conv rows top to bottom, two
// at a time:
//   * all kt*7 weight slices (64 x 1120, 143 KB in 16-bit) are loaded into LDS once per frame - the un-fused
//     kernel re-fetches them for every 256 positions;
//   * wave w owns the 16-column tile w of BOTH rows of the pair (2 m-tiles x 4 channel tiles): the 3-row
//     vertical max of the pool is then lane-local - previous odd row (kept in registers), even row, odd row;
//   * the vertically reduced row goes through a 14 KB LDS line, the horizontal 3-max + stride 2 is taken there
//     and the pooled row leaves as whole 128-byte pixels.
// The 64x32x112x112 conv output never exists in HBM (B=16: -822 MB written, -822 MB read, one launch less).
// Activation fragments come straight from the padded stem input by aligned 16-byte loads (as in af_stem.hip).
#include "af_common.h"

namespace af {

struct StemPoolArgs {
    const char* in;      // padded input [N][T+4][H+6][W+8][4]
    const char* w;       // packed [kt][7][4 chunks][64][16 B]
    const float* scale;
    const float* shift;
    char* out;           // pooled [N][T][Hp][Wq][64]
    int Tp, Hp, Wp;      // padded input dims
    int kt;
    int To, Ho, Wo;      // conv output dims
    int Hq, Wq;          // pooled dims
    int frames;          // N*To
    int bands, band_rows;// pooled-row bands per frame (work unit = (frame, band)), pooled rows per band
};

template <int DT, int KT>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const StemPoolArgs a) {
    typedef Elem<DT> E;
    typedef typename E::type elem_t;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int KH = 7, NCH = 4, COUT = 64, TN = 4;
    constexpr int PIXB = 8;                                    // bytes per padded input pixel (4 x 16 bit)

    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int wchunks = KT * KH * NCH * COUT;                  // uint4 in the weight image
    uint4* wl = smem;
    elem_t* line = reinterpret_cast<elem_t*>(smem + wchunks);  // [Wo_tiles*16][64] vertically reduced conv row
    const int ncol_tiles = (a.Wo + 15) >> 4;                   // <= 8 (host-checked)
    const bool active = wave < ncol_tiles;

    // weights -> LDS, once
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.w);
    for (int i = tid; i < wchunks; i += 512) wl[i] = wsrc[i];

    f32x4 sc[TN], sf[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + i * 16 + fg * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + i * 16 + fg * 4);
    }
    const long long row_bytes = (long long)a.Wp * PIXB, plane_bytes = row_bytes * a.Hp;
    int wo = wave * 16 + frow;
    if (wo > a.Wo - 1) wo = a.Wo - 1;                          // clamp: loads stay inside the row, result discarded

    for (int unit = blockIdx.x; unit < a.frames * a.bands; unit += gridDim.x) {
        const int frame = unit / a.bands, band = unit - frame * a.bands;
        const int n = frame / a.To, to = frame - n * a.To;
        // pooled rows [j_begin, j_end) of this frame; a band that does not start at the top first runs the row pair above
        // it (not stored) to get the conv row its first pooled row reaches up to
        const int j_begin = band * a.band_rows, j_end = (j_begin + a.band_rows < a.Hq) ? j_begin + a.band_rows : a.Hq;
        const int j_first = j_begin > 0 ? j_begin - 1 : 0;
        // padded coords of tap (0,0,0) for conv output (to, ho, wo): t = to, h = 2*ho, w = 2*wo
        const char* fin = a.in + ((long long)(n * a.Tp + to) * a.Hp) * row_bytes + (long long)(2 * wo) * PIXB + fg * 16;
        f32x4 prev[TN];                                        // conv row 2j-1 (post-ReLU); 0 == -inf after a ReLU
#pragma unroll
        for (int i = 0; i < TN; ++i) prev[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                       // weights visible (first frame) / line free
        // current dt plane's fragments: conv row 2j reads padded input rows 4j .. 4j+6, conv row 2j+1 rows 4j+2 .. 4j+8 -
        // NINE distinct rows for the pair, fetched once (row index clamped to the padded image: only a discarded odd
        // row of an odd-height image ever reaches past it)
        constexpr int NR = KH + 2;
        uint4 bx[NR];
        if (active) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = 4 * j_first + r < a.Hp ? 4 * j_first + r : a.Hp - 1;
                bx[r] = *reinterpret_cast<const uint4*>(fin + (long long)row * row_bytes);
            }
        }

        for (int j = j_first; j < j_end; ++j) {
            f32x4 acc[2][TN];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < TN; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (active) {
                // activation fragments of one dt plane (9 input rows feed the 7 kernel rows of both conv rows: 9 aligned 16-byte loads) are
                // fetched a whole plane ahead of the MFMAs that use them - the last plane of a row pair prefetches
                // the first plane of the next pair; everything is unrolled, so the "copy" is register renaming
#pragma unroll
                for (int dt = 0; dt < KT; ++dt) {
                    uint4 nx[NR];
                    const bool more = dt + 1 < KT || j + 1 < j_end;
                    if (more) {
                        const int jn = dt + 1 < KT ? j : j + 1, dtn = dt + 1 < KT ? dt + 1 : 0;
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const int row = 4 * jn + r < a.Hp ? 4 * jn + r : a.Hp - 1;
                            nx[r] = *reinterpret_cast<const uint4*>(fin + (long long)row * row_bytes + dtn * plane_bytes);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);         // keep the 9 loads AHEAD of this plane's MFMAs (hipcc sinks them otherwise)
#pragma unroll
                    for (int dh = 0; dh < KH; ++dh) {
                        const uint4* wrow = wl + ((dt * KH + dh) * NCH + fg) * COUT + frow;
#pragma unroll
                        for (int i = 0; i < TN; ++i) {
                            const uint4 af = wrow[i * 16];
                            Mma<DT>::run(af, bx[dh], acc[0][i]);
                            Mma<DT>::run(af, bx[dh + 2], acc[1][i]);
                        }
                    }
                    if (more) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) bx[r] = nx[r];
                    }
                }
            }
            // BN + ReLU, vertical 3-max (rows 2j-1, 2j, 2j+1), keep the odd row for the next pair
            const bool odd_ok = 2 * j + 1 < a.Ho;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                f32x4 v0 = acc[0][i] * sc[i] + sf[i], v1 = acc[1][i] * sc[i] + sf[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = relu_f(v0[e]);
                    v1[e] = odd_ok ? relu_f(v1[e]) : 0.f;
                    const float m = max_nan(max_nan(prev[i][e], v0[e]), v1[e]);
                    prev[i][e] = v1[e];
                    v0[e] = m;
                }
                if (active) {                                  // line[col][64 ch]: 4 consecutive channels per lane
                    typedef elem_t e4 __attribute__((ext_vector_type(4)));
                    e4 o;
                    o[0] = E::from_f32(v0[0]); o[1] = E::from_f32(v0[1]); o[2] = E::from_f32(v0[2]); o[3] = E::from_f32(v0[3]);
                    *reinterpret_cast<e4*>(line + (wave * 16 + frow) * COUT + i * 16 + fg * 4) = o;
                }
            }
            __syncthreads();
            // horizontal 3-max, stride 2: pooled col q <- conv cols 2q-1, 2q, 2q+1; 8 channels (16 B) per thread
            for (int idx = tid; idx < (j >= j_begin ? a.Wq * 8 : 0); idx += 512) {
                const int q = idx >> 3, ch = (idx & 7) * 8;
                // post-ReLU 16-bit patterns order like unsigned integers, NaN (either sign) on top: v_pk_max_u16 (as in af_stem3.hip)
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                u16x8 m = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int d = -1; d <= 1; ++d) {
                    const int c = 2 * q + d;
                    if (c >= 0 && c < a.Wo)
                        m = __builtin_elementwise_max(m, *reinterpret_cast<const u16x8*>(line + c * COUT + ch));
                }
                *reinterpret_cast<u16x8*>(a.out + ((((long long)n * a.To + to) * a.Hq + j) * a.Wq + q) * (COUT * 2) + ch * 2) = m;
            }
            __syncthreads();                                   // line may be overwritten by the next row pair
        }
    }
}

template <int DT, int KT>
static int launch_stem_pool_kt(const StemPoolArgs& a, hipStream_t stream) {
    const int g_cus = device_cus();
    const int lds = a.kt * 7 * 4 * 64 * 16 + ((a.Wo + 15) / 16) * 16 * 64 * 2;
    if (lds > 160 * 1024) return set_error(AF_ERR_ARG, "stem_pool: %d bytes of LDS needed", lds);
    AF_SET_MAX_LDS((&stem_pool_kernel<DT, KT>), 160 * 1024, "stem_pool");
    const int units = a.frames * a.bands;
    const int grid = units < g_cus ? units : g_cus;           // one resident workgroup per CU; weights loaded once each
    hipLaunchKernelGGL((stem_pool_kernel<DT, KT>), dim3(grid), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("stem_pool_kernel");
    return AF_OK;
}

template <int DT>
static int launch_stem_pool(const StemPoolArgs& a, hipStream_t stream) {
    switch (a.kt) {
        case 1: return launch_stem_pool_kt<DT, 1>(a, stream);
        case 3: return launch_stem_pool_kt<DT, 3>(a, stream);
        default: return launch_stem_pool_kt<DT, 5>(a, stream);
    }
}

}  // namespace af

extern "C" int af_stem_conv_bn_relu_maxpool(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                            const float* scale, const float* shift, void* out, void* stream) {
    using namespace af;
    AF_REQUIRE(d && stem_in && w_packed && scale && shift && out, "stem_pool: null argument");
    AF_REQUIRE(d->dtype == AF_BF16 || d->dtype == AF_F16, "stem_pool: 16-bit dtypes only (fp32 weights do not fit LDS)");
    AF_REQUIRE(d->cin == 3 && d->cout == 64, "stem_pool: expects 3 -> 64 channels");
    AF_REQUIRE(d->kh == 7 && d->kw == 7 && d->sh == 2 && d->sw == 2 && d->st == 1 && d->ph == 3 && d->pw == 3,
               "stem_pool: expects a [kt,7,7] kernel, stride [1,2,2], pad [kt/2,3,3]");
    AF_REQUIRE(d->kt >= 1 && d->kt <= 2 * AF_STEM_PAD_T + 1 && (d->kt & 1) && d->pt == d->kt / 2, "stem_pool: bad kt/pt");
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0, "stem_pool: bad dims");
    const int to = d->t, ho = (d->h + 6 - 7) / 2 + 1, wo = (d->w + 6 - 7) / 2 + 1;
    AF_REQUIRE(to == d->to && ho == d->ho && wo == d->wo, "stem_pool: conv output dims mismatch");
    AF_REQUIRE(wo <= 128, "stem_pool: conv output width %d > 128", wo);
    AF_REQUIRE(aligned16(stem_in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out),
               "stem_pool: buffers must be 16-byte aligned");
    StemPoolArgs a;
    a.in = (const char*)stem_in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.Tp = d->t + 2 * AF_STEM_PAD_T; a.Hp = d->h + 2 * AF_STEM_PAD_H; a.Wp = d->w + AF_STEM_PAD_W_TOTAL;
    a.kt = d->kt; a.To = to; a.Ho = ho; a.Wo = wo;
    a.Hq = (ho - 1) / 2 + 1; a.Wq = (wo - 1) / 2 + 1;
    a.frames = d->n * to;
    // fewer frames than CUs (batch < 8 clips of 32 frames): cut every frame into bands of pooled rows so that the whole
    // chip works; a band costs one extra (unstored) row pair, so no bands once the frames alone fill the CUs
    {
        const int cus = device_cus();
        int bands = a.frames >= cus ? 1 : cus / a.frames;
        if (bands > a.Hq / 4) bands = a.Hq / 4 > 0 ? a.Hq / 4 : 1;      // at least 4 pooled rows per band
        a.band_rows = (a.Hq + bands - 1) / bands;
        a.bands = (a.Hq + a.band_rows - 1) / a.band_rows;
    }
    a.in += (long long)(AF_STEM_PAD_T - d->pt) * a.Hp * a.Wp * 4 * dtype_size(d->dtype);
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == AF_BF16 ? launch_stem_pool<AF_BF16>(a, s) : launch_stem_pool<AF_F16>(a, s);
}

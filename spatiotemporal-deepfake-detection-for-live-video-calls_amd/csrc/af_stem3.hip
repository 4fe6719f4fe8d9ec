// K-packed fused stem: Conv3d(3->64,[kt,7,7],s[1,2,2]) + BN + ReLU + MaxPool3d([1,3,3],s[1,2,2],p[0,1,1])
// (reference altfreezing/slowfast/models/stem_helper.py:156-178) as ONE launch, 16-bit operands - the successor of
// af_stem_pool.hip for the 3-channel input.
//
// af_stem_pool.hip keeps the input as 4-channel pixels and spends one MFMA K-block (32) per kernel row (dt, dh):
// 8 pixels x 4 channels for 7 x 3 useful taps = 66 % useful MACs, 35 K-blocks for kt = 5.  The stem is MFMA-bound on
// exactly that padding.  Here the input keeps its 3 REAL channels (6 bytes per pixel, "rgb3" layout written by
// af_pack_input_*_rgb3) and K is the flat sequence of 16-byte FRAGMENTS (dt, dh, j), j = 0..2: fragment j of a kernel row
// is elements 8j .. 8j+7 of the row's 24 consecutive (dw, c) values (21 real taps + 3 that meet zero weights), i.e. 16
// contiguous bytes at (row, pixel 2*wo) + 16 j - never straddling an input row.  A K-block is 4 consecutive fragments (one
// per 16-lane group), so lanes of different groups read different (dt, dh) rows: each lane keeps the byte offset of its
// fragment of every block (kt = 5: 105 fragments -> 27 blocks instead of 35: 1.30x fewer MFMAs and LDS weight reads).
// The two conv rows of a pair no longer share loads (their fragments sit in different lane groups): 2 loads per block.
// Everything else is af_stem_pool.hip's design: persistent workgroup per (frame, band of pooled rows), all weight blocks
// resident in LDS, wave = 16 columns x 2 conv rows x 64 channels, lane-local vertical 3-max, horizontal 3-max through an
// LDS line, only the pooled tensor written.
#include "af_common.h"

namespace af {

struct Stem3Args {
    const char* in;      // rgb3 input [N][T+4][H+6] rows of `row_bytes` (pixels of 3 x 16 bit, left halo 3 pixels)
    const char* w;       // packed [NBLK][4 groups][64][16 B]
    const float* scale;
    const float* shift;
    char* out;           // pooled [N][T][Hq][Wq][64]
    int Tp, Hp;          // padded input frames / rows
    int row_bytes;       // input row pitch (multiple of 16)
    int kt;
    int To, Ho, Wo;      // conv output dims
    int Hq, Wq;          // pooled dims
    int frames;          // N*To
    int bands, band_rows;
    int out_ld;          // pooled row stride in channels (>= 64: room for a lateral's channels behind them)
};

// PW: the image is at most 7 column tiles wide (224 x 224: 112 columns), so wave 7 has no MFMA work - it becomes the
// POOLING wave: the horizontal 3-max / stride 2 and the store of row pair j run there while waves 0-6 already multiply
// pair j + 1 (two LDS lines, ONE workgroup barrier per pair instead of two, the pooling off the critical path).
// Round 3: the MFMA loop is software-pipelined by hand.  hipcc had scheduled it as  2 weight reads -> wait -> 2 MFMAs -> wait ->
// 2 MFMAs  (a chain of LDS latencies: 57 % MFMA issue on the two-wave SIMDs), read every fragment offset from an LDS table in
// front of its load (an lgkmcnt(0) stall per group) and drained vmcnt(0) at every row-pair boundary to copy the prefetched
// group into place.  Now: the four weight fragments of block b + 1 are read while block b multiplies (two register sets,
// sched_barrier-pinned; the reads of the next pair's block 0 ride under the last block), fragment offsets live in registers,
// and the activation groups alternate between two static register sets with an EVEN number of groups per pair, so the
// group prefetched for the next pair is already where that pair expects it - no copy, no drain.
template <int DT, int KT, bool PW, int GB>
__global__ __launch_bounds__(512, 2) void stem3_pool_kernel(const Stem3Args a) {
    typedef Elem<DT> E;
    typedef typename E::type elem_t;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int COUT = 64, TN = 4;
    constexpr int NF = KT * 7 * 3;                             // 16-byte fragments of the kernel
    constexpr int NBLK = (NF + 3) / 4;                         // MFMA K-blocks
    constexpr int NGRP = (NBLK + GB - 1) / GB;                 // prefetch groups of GB blocks per row pair
    static_assert(NGRP % 2 == 0 || NGRP == 1, "an even number of groups: the two activation register sets alternate");

    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    constexpr int wchunks = NBLK * 4 * COUT;                   // uint4 in the weight image
    uint4* wl = smem;
    elem_t* line = reinterpret_cast<elem_t*>(smem + wchunks);  // [Wo_tiles*16][64] vertically reduced conv row
    const int ncol_tiles = (a.Wo + 15) >> 4;                   // <= 8 (host-checked)
    const int line_elems = ncol_tiles * 16 * COUT;
    const bool active = wave < ncol_tiles;
    // BN scale / shift in LDS: a global load in the epilogue would be waited for with vmcnt(0), i.e. together with the
    // activation group already prefetched for the next row pair
    float* bn = reinterpret_cast<float*>(line + (PW ? 3 : 1) * line_elems);     // [2][64]

    const uint4* wsrc = reinterpret_cast<const uint4*>(a.w);
    for (int i = tid; i < wchunks; i += 512) wl[i] = wsrc[i];
    if (tid < 2 * COUT) bn[tid] = tid < COUT ? a.scale[tid] : a.shift[tid - COUT];

    const int plane_bytes = a.row_bytes * a.Hp;                // < 2^31 (host-checked)
    int wo = wave * 16 + frow;
    if (wo > a.Wo - 1) wo = a.Wo - 1;                          // clamp: loads stay inside the row, result discarded
    // byte offset of this lane's fragment of every block: fragment f = 4 blk + g = (dt, dh, j) from (frame t, row 4j, pixel
    // 2 wo at 6 bytes per pixel); NBLK registers per lane
    int fo_tab[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        int f = b * 4 + fg;
        if (f > NF - 1) f = NF - 1;                            // beyond the kernel: any valid address (zero weights)
        const int dt = f / 21, r = f - dt * 21, dh = r / 3, jj = r - dh * 3;
        fo_tab[b] = wo * 12 + dt * plane_bytes + dh * a.row_bytes + jj * 16;
    }
    const int pair_bytes = 2 * a.row_bytes;                    // conv row 2j+1 starts two input rows below conv row 2j
    // weight fragments of block b (one per channel tile) from the LDS image
    auto read_a = [&](uint4 (&d)[TN], int b) {
        const uint4* wrow = wl + (b * 4 + fg) * COUT + frow;
#pragma unroll
        for (int i = 0; i < TN; ++i) d[i] = wrow[i * 16];
    };

    // units -> workgroups, XCD-contiguous (placement is a speed matter only): workgroups b and b + 8 share an XCD and its L2,
    // so XCD x takes the x-th eighth of the units - at 16 clips two whole clips, whose 32 frames its 32 CUs convolve at the same
    // time: every input plane then enters that L2 once instead of once per temporal tap on five different XCDs
    const int total_units = a.frames * a.bands;
    const bool xcd_map = (gridDim.x & 7) == 0;
    const int upx = (total_units + 7) >> 3;
    const int u_begin = xcd_map ? (int)(blockIdx.x & 7) * upx + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int u_end = xcd_map ? (((int)(blockIdx.x & 7) + 1) * upx < total_units ? ((int)(blockIdx.x & 7) + 1) * upx : total_units) : total_units;
    const int u_step = xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    for (int unit = u_begin; unit < u_end; unit += u_step) {
        const int frame = unit / a.bands, band = unit - frame * a.bands;
        const int n = frame / a.To, to = frame - n * a.To;
        const int j_begin = band * a.band_rows, j_end = (j_begin + a.band_rows < a.Hq) ? j_begin + a.band_rows : a.Hq;
        const int j_first = j_begin > 0 ? j_begin - 1 : 0;
        const char* fin = a.in + ((long long)(n * a.Tp + to) * a.Hp) * a.row_bytes;        // uniform
        f32x4 prev[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) prev[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                       // weights visible (first unit) / line free

        // fragments of one group of blocks, both conv rows; the next group is fetched while this one multiplies
        uint4 xs[2][GB][2];
        auto load_group = [&](uint4 (&dst)[GB][2], int jrow, int grp) {
            const char* rb0 = fin + (long long)(4 * jrow) * a.row_bytes;                   // uniform
            const char* rb1 = rb0 + pair_bytes;
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int b = grp * GB + k;
                if (b < NBLK) {
                    dst[k][0] = *reinterpret_cast<const uint4*>(rb0 + (unsigned)fo_tab[b]);
#ifdef AF_STEM_NO_ROW1_LOADS                               // timing-only ablation (diagnostic builds): what do the second row's loads cost?
                    dst[k][1] = dst[k][0];
#else
                    dst[k][1] = *reinterpret_cast<const uint4*>(rb1 + (unsigned)fo_tab[b]);
#endif
                }
            }
        };
        uint4 afc[TN], afn[TN];
        if (active) { load_group(xs[0], j_first, 0); read_a(afc, 0); }
        f32x4 acc[2][TN];

        // the K loop of one row pair (accumulators zeroed first)
        auto mfma_pair = [&](int j) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < TN; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
                // the next group (of the next pair after the last one) goes into the other register set
                // (unconditionally - behind a branch hipcc's vmcnt counts for the last blocks assume the shorter path and
                // drain the prefetch; the last pair of a unit re-fetches its own group 0 and drops it)
                if (grp + 1 < NGRP) load_group(xs[(grp + 1) & 1], j, grp + 1);
                else load_group(xs[(grp + 1) & 1], j + 1 < j_end ? j + 1 : j, 0);
                __builtin_amdgcn_sched_barrier(0);             // keep the loads AHEAD of this group's MFMAs
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    const int b = grp * GB + k;
                    if (b < NBLK) {
                        read_a(afn, b + 1 < NBLK ? b + 1 : 0);             // next block's weights under this block's MFMAs
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < TN; ++i) {
                            Mma<DT>::run(afc[i], xs[grp & 1][k][0], acc[0][i]);
                            Mma<DT>::run(afc[i], xs[grp & 1][k][1], acc[1][i]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < TN; ++i) afc[i] = afn[i];
                    }
                }
            }
        };
        // BN + ReLU, vertical 3-max (rows 2j-1, 2j, 2j+1), keep the odd row for the next pair; the reduced row -> `dst` line
        auto epilogue = [&](int j, elem_t* dst) {
            const bool odd_ok = 2 * j + 1 < a.Ho;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + i * 16 + fg * 4);
                const f32x4 sf = *reinterpret_cast<const f32x4*>(bn + COUT + i * 16 + fg * 4);
                f32x4 v0 = acc[0][i] * sc + sf, v1 = acc[1][i] * sc + sf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = relu_f(v0[e]);
                    v1[e] = odd_ok ? relu_f(v1[e]) : 0.f;
                    const float m = max_nan(max_nan(prev[i][e], v0[e]), v1[e]);
                    prev[i][e] = v1[e];
                    v0[e] = m;
                }
                typedef elem_t e4 __attribute__((ext_vector_type(4)));
                e4 o;
                o[0] = E::from_f32(v0[0]); o[1] = E::from_f32(v0[1]); o[2] = E::from_f32(v0[2]); o[3] = E::from_f32(v0[3]);
                *reinterpret_cast<e4*>(dst + (wave * 16 + frow) * COUT + i * 16 + fg * 4) = o;
            }
        };
        // horizontal 3-max, stride 2: pooled col q <- conv cols 2q-1, 2q, 2q+1; 8 channels (16 B) per thread -> output row j
        auto pool_row = [&](int j, const elem_t* pline, int pstart, int pstep) {
            for (int idx = pstart; idx < a.Wq * 8; idx += pstep) {
                const int q = idx >> 3, ch = (idx & 7) * 8;
                // the line holds values that went through ReLU: zero, positive, +inf or NaN.  For those the 16-bit patterns of bf16 and
                // f16 order like unsigned integers with every NaN (either sign) above +inf, so the max is v_pk_max_u16 - two values per
                // instruction, no conversion - and a NaN member makes the result NaN like ATen's max_pool; 0 = the pool's padding
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                u16x8 m = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int d = -1; d <= 1; ++d) {
                    const int c = 2 * q + d;
                    if (c >= 0 && c < a.Wo)
                        m = __builtin_elementwise_max(m, *reinterpret_cast<const u16x8*>(pline + c * COUT + ch));
                }
                *reinterpret_cast<u16x8*>(a.out + ((((long long)n * a.To + to) * a.Hq + j) * a.Wq + q) * (a.out_ld * 2) + ch * 2) = m;
            }
        };

        if (PW) {
            // Pooling-wave form, STAGGERED (round 3).  The two MFMA waves of a SIMD ran in lockstep - both in their K loop, then
            // both in their epilogue (~200 VALU instructions each with the matrix pipe idle), then the barrier.  Waves 4-6 (the
            // second wave of SIMDs 0-2) now defer a pair's epilogue to the start of the next interval: while waves 0-3 multiply
            // pair j, waves 4-6 finish pair j - 1 and then multiply pair j under the epilogue of waves 0-3 - matrix beside vector
            // on every SIMD.  A row pair's line is therefore complete one barrier later (three line buffers): in interval it,
            // early waves write line it % 3, late waves line (it - 1) % 3, and wave 7 pools line (it - 2) % 3.  One barrier per
            // interval for everybody, npairs + 2 intervals per unit.
            const bool late = wave >= 4;
            const int npairs = j_end - j_first;
            for (int it = 0; it < npairs + 2; ++it) {
                const int j = j_first + it;
                if (active) {                                  // (one K-loop call site: two inlined copies cost 68 spilled registers)
                    if (late && it >= 1 && it <= npairs) epilogue(j - 1, line + ((it + 2) % 3) * line_elems);
                    if (it < npairs) mfma_pair(j);
                    if (!late && it < npairs) epilogue(j, line + (it % 3) * line_elems);
                } else if (wave == 7 && it >= 2 && j - 2 >= j_begin) {
                    pool_row(j - 2, line + ((it + 1) % 3) * line_elems, lane, 64);
                }
                __syncthreads();
            }
        } else {
            for (int j = j_first; j < j_end; ++j) {
                if (active) { mfma_pair(j); epilogue(j, line); }
                __syncthreads();
                if (j >= j_begin) pool_row(j, line, tid, 512);
                __syncthreads();                               // line may be overwritten by the next row pair
            }
        }
    }
}

// (64,3,kt,7,7) fp32 -> [NBLK][4][64][8]: fragment f = 4 blk + g = (dt, dh, j); element e of it is tap (dw, c) with
// 3 dw + c = 8 j + e (beyond 20: zero), fragments beyond the kernel are zero
template <int DT>
__global__ void pack_stem3_weight_kernel(const float* __restrict__ w, int cout, int kt, int nblk,
                                         typename Elem<DT>::type* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)nblk * 4 * 64 * 8;
    if (idx >= total) return;
    const int e = (int)(idx & 7); long long r = idx >> 3;
    const int o = (int)(r % 64); r /= 64;
    const int f = (int)r;                                      // 4 blk + g
    float v = 0.f;
    if (f < kt * 21 && o < cout) {
        const int dt = f / 21, q = f - dt * 21, dh = q / 3, jj = q - dh * 3, k = 8 * jj + e;
        if (k < 21) { const int dw = k / 3, c = k - dw * 3; v = w[((((long long)o * 3 + c) * kt + dt) * 7 + dh) * 7 + dw]; }
    }
    out[idx] = Elem<DT>::from_f32(v);
}

// interior of the rgb3 stem input: one thread per pixel, three 16-bit stores
template <int DT, typename Src>
__global__ void pack_input3_kernel(Src src, int n, int t, int h, int w, int row_bytes, char* __restrict__ out) {
    typedef typename Elem<DT>::type elem_t;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)n * t * h * w;
    if (idx >= total) return;
    const int x = (int)(idx % w); long long r = idx / w;
    const int y = (int)(r % h); r /= h;
    const int z = (int)(r % t); const long long b = r / t;
    const long long Tp = t + 2 * AF_STEM_PAD_T, Hp = h + 2 * AF_STEM_PAD_H;
    elem_t* o = reinterpret_cast<elem_t*>(out + ((b * Tp + z + AF_STEM_PAD_T) * Hp + y + AF_STEM_PAD_H) * row_bytes) +
                (x + AF_STEM_PAD_W_LEFT) * 3;
    o[0] = Elem<DT>::from_f32(src(b, 0, z, y, x));
    o[1] = Elem<DT>::from_f32(src(b, 1, z, y, x));
    o[2] = Elem<DT>::from_f32(src(b, 2, z, y, x));
}

struct Src3F32 {
    const float* p; long long sn, sc, st, sh, sw;
    __device__ float operator()(long long b, int c, int z, int y, int x) const { return p[b * sn + c * sc + z * st + y * sh + x * sw]; }
};
struct Src3U8 {
    const uint8_t* p; int t, h, w; float mean[3], stdv[3];
    __device__ float operator()(long long b, int c, int z, int y, int x) const {
        // (float(u8) - mean) / std : the callers' x.sub(mean).div(std) on float32 (af_realtime.py:83)
        return ((float)p[(((b * t + z) * h + y) * w + x) * 3 + c] - mean[c]) / stdv[c];
    }
};

static inline int rgb3_row_bytes(int w) { return ((w + AF_STEM_PAD_W_TOTAL) * 6 + 15) & ~15; }

template <int DT, int KT, bool PW>
static int launch_stem3_pw(const Stem3Args& a, hipStream_t stream) {
    constexpr int NBLK = (KT * 21 + 3) / 4;
    // blocks per prefetch group: an even number of groups per row pair (27 blocks: 6 x 5; 16: 4 x 4; 6: 2 x 3; register budget: 2 x GB x 8 for the two sets)
    constexpr int GB = KT == 5 ? 5 : KT == 3 ? 4 : 3;
    const int lds = NBLK * 4 * 64 * 16 + (PW ? 3 : 1) * ((a.Wo + 15) / 16) * 16 * 64 * 2 + 2 * 64 * 4;
    if (lds > 160 * 1024) return set_error(AF_ERR_ARG, "stem3_pool: %d bytes of LDS needed", lds);
    AF_SET_MAX_LDS((&stem3_pool_kernel<DT, KT, PW, GB>), 160 * 1024, "stem3_pool");
    const int cus = device_cus(), units = a.frames * a.bands;
    hipLaunchKernelGGL((stem3_pool_kernel<DT, KT, PW, GB>), dim3(units < cus ? units : cus), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("stem3_pool_kernel");
    return AF_OK;
}

template <int DT, int KT>
static int launch_stem3_kt(const Stem3Args& a, hipStream_t stream) {
    // a free eighth wave (<= 7 column tiles) takes the pooling off the MFMA waves' critical path
    return (a.Wo + 15) / 16 <= 7 ? launch_stem3_pw<DT, KT, true>(a, stream) : launch_stem3_pw<DT, KT, false>(a, stream);
}

template <typename Src>
static int launch_pack_input3(const Src& src, int n, int t, int h, int w, int dtype, void* out, hipStream_t s) {
    const long long total = (long long)n * t * h * w;
    dim3 g((unsigned)((total + 255) / 256)), b(256);
    const int rb = rgb3_row_bytes(w);
    if (dtype == AF_BF16) hipLaunchKernelGGL((pack_input3_kernel<AF_BF16, Src>), g, b, 0, s, src, n, t, h, w, rb, (char*)out);
    else hipLaunchKernelGGL((pack_input3_kernel<AF_F16, Src>), g, b, 0, s, src, n, t, h, w, rb, (char*)out);
    AF_CHECK_LAUNCH("pack_input3_kernel");
    return AF_OK;
}

}  // namespace af

using namespace af;

extern "C" int64_t af_stem_input_bytes_rgb3(int n, int t, int h, int w, int dtype) {
    if ((dtype != AF_BF16 && dtype != AF_F16) || n <= 0 || t <= 0 || h <= 0 || w <= 0) return AF_ERR_ARG;
    // + 8 rows of slack: the (discarded) odd conv row below an odd-height image reads past the last frame's last row
    return ((int64_t)n * (t + 2 * AF_STEM_PAD_T) * (h + 2 * AF_STEM_PAD_H) + 8) * rgb3_row_bytes(w);
}

extern "C" int af_pack_input_f32_rgb3(const float* x, int n, int t, int h, int w, int64_t stride_n, int64_t stride_c,
                                      int64_t stride_t, int64_t stride_h, int64_t stride_w, int dtype, void* stem_in,
                                      void* stream) {
    AF_REQUIRE(x && stem_in && (dtype == AF_BF16 || dtype == AF_F16) && n > 0 && t > 0 && h > 0 && w > 0, "pack_input_f32_rgb3: bad argument");
    AF_REQUIRE(aligned16(stem_in), "pack_input_f32_rgb3: output must be 16-byte aligned");
    Src3F32 src{x, stride_n, stride_c, stride_t, stride_h, stride_w};
    return launch_pack_input3(src, n, t, h, w, dtype, stem_in, (hipStream_t)stream);
}

extern "C" int af_pack_input_u8_rgb3(const uint8_t* clips, int n, int t, int h, int w, const float mean[3],
                                     const float std_[3], int dtype, void* stem_in, void* stream) {
    AF_REQUIRE(clips && stem_in && mean && std_ && (dtype == AF_BF16 || dtype == AF_F16) && n > 0 && t > 0 && h > 0 && w > 0,
               "pack_input_u8_rgb3: bad argument");
    AF_REQUIRE(aligned16(stem_in), "pack_input_u8_rgb3: output must be 16-byte aligned");
    Src3U8 src;
    src.p = clips; src.t = t; src.h = h; src.w = w;
    for (int i = 0; i < 3; ++i) { src.mean[i] = mean[i]; src.stdv[i] = std_[i]; }
    return launch_pack_input3(src, n, t, h, w, dtype, stem_in, (hipStream_t)stream);
}

extern "C" int64_t af_packed_stem_weight_bytes_rgb3(int kt, int dtype) {
    if ((dtype != AF_BF16 && dtype != AF_F16) || kt <= 0 || kt > 2 * AF_STEM_PAD_T + 1) return AF_ERR_ARG;
    return (int64_t)((kt * 21 + 3) / 4) * 4 * 64 * 16;
}

extern "C" int af_pack_stem_weight_rgb3(const float* w, int cout, int kt, int dtype, void* packed, void* stream) {
    AF_REQUIRE(w && packed && (dtype == AF_BF16 || dtype == AF_F16) && cout > 0 && cout <= 64 && kt > 0 && kt <= 2 * AF_STEM_PAD_T + 1,
               "pack_stem_weight_rgb3: bad argument");
    const int nblk = (kt * 21 + 3) / 4;
    const long long total = (long long)nblk * 4 * 64 * 8;
    dim3 g((unsigned)((total + 255) / 256)), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == AF_BF16) hipLaunchKernelGGL((pack_stem3_weight_kernel<AF_BF16>), g, b, 0, s, w, cout, kt, nblk, (__bf16*)packed);
    else hipLaunchKernelGGL((pack_stem3_weight_kernel<AF_F16>), g, b, 0, s, w, cout, kt, nblk, (_Float16*)packed);
    AF_CHECK_LAUNCH("pack_stem3_weight_kernel");
    return AF_OK;
}

extern "C" int af_stem_conv_bn_relu_maxpool_rgb3_ld(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                                    const float* scale, const float* shift, void* out, int out_ld, void* stream);

extern "C" int af_stem_conv_bn_relu_maxpool_rgb3(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                                 const float* scale, const float* shift, void* out, void* stream) {
    return af_stem_conv_bn_relu_maxpool_rgb3_ld(d, stem_in, w_packed, scale, shift, out, 0, stream);
}

extern "C" int af_stem_conv_bn_relu_maxpool_rgb3_ld(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                                    const float* scale, const float* shift, void* out, int out_ld, void* stream) {
    AF_REQUIRE(d && stem_in && w_packed && scale && shift && out, "stem3_pool: null argument");
    AF_REQUIRE(d->dtype == AF_BF16 || d->dtype == AF_F16, "stem3_pool: 16-bit dtypes only");
    AF_REQUIRE(d->cin == 3 && d->cout == 64, "stem3_pool: expects 3 -> 64 channels");
    AF_REQUIRE(d->kh == 7 && d->kw == 7 && d->sh == 2 && d->sw == 2 && d->st == 1 && d->ph == 3 && d->pw == 3,
               "stem3_pool: expects a [kt,7,7] kernel, stride [1,2,2], pad [kt/2,3,3]");
    AF_REQUIRE(d->kt >= 1 && d->kt <= 2 * AF_STEM_PAD_T + 1 && (d->kt & 1) && d->pt == d->kt / 2, "stem3_pool: bad kt/pt");
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0, "stem3_pool: bad dims");
    const int to = d->t, ho = (d->h + 6 - 7) / 2 + 1, wo = (d->w + 6 - 7) / 2 + 1;
    AF_REQUIRE(to == d->to && ho == d->ho && wo == d->wo, "stem3_pool: conv output dims mismatch");
    AF_REQUIRE(wo <= 128, "stem3_pool: conv output width %d > 128", wo);
    AF_REQUIRE(aligned16(stem_in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out),
               "stem3_pool: buffers must be 16-byte aligned");
    Stem3Args a;
    a.in = (const char*)stem_in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.Tp = d->t + 2 * AF_STEM_PAD_T; a.Hp = d->h + 2 * AF_STEM_PAD_H; a.row_bytes = rgb3_row_bytes(d->w);
    AF_REQUIRE((long long)a.row_bytes * a.Hp * (d->kt + 1) < (1LL << 31), "stem3_pool: frame too large for 32-bit fragment offsets");
    a.kt = d->kt; a.To = to; a.Ho = ho; a.Wo = wo;
    a.Hq = (ho - 1) / 2 + 1; a.Wq = (wo - 1) / 2 + 1;
    a.frames = d->n * to;
    if (out_ld == 0) out_ld = 64;
    AF_REQUIRE(out_ld >= 64 && out_ld % 8 == 0, "stem3_pool: bad out_ld %d", out_ld);
    a.out_ld = out_ld;
    {
        const int cus = device_cus();
        int bands = a.frames >= cus ? 1 : cus / a.frames;
        if (bands > a.Hq / 4) bands = a.Hq / 4 > 0 ? a.Hq / 4 : 1;      // at least 4 pooled rows per band
        a.band_rows = (a.Hq + bands - 1) / bands;
        a.bands = (a.Hq + a.band_rows - 1) / a.band_rows;
    }
    a.in += (long long)(AF_STEM_PAD_T - d->pt) * a.Hp * a.row_bytes;
    hipStream_t s = (hipStream_t)stream;
    switch (d->kt) {
        case 1: return d->dtype == AF_BF16 ? launch_stem3_kt<AF_BF16, 1>(a, s) : launch_stem3_kt<AF_F16, 1>(a, s);
        case 3: return d->dtype == AF_BF16 ? launch_stem3_kt<AF_BF16, 3>(a, s) : launch_stem3_kt<AF_F16, 3>(a, s);
        default: return d->dtype == AF_BF16 ? launch_stem3_kt<AF_BF16, 5>(a, s) : launch_stem3_kt<AF_F16, 5>(a, s);
    }
}

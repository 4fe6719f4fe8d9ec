// Kernels that only the FTCN-TT plugin needs (reference altfreezing/model/classifier/
// i3d_temporal_var_fix_dropout_tt_cfg.py + time_transformer.py); its trunk otherwise runs on the I3D kernels
// (all of its convolutions are Tx1x1 / 1x1x1).
//
//   * temporal stem: Conv3d(3->64, [kt,1,1], stride 1, pad [kt/2,0,0]) + BN + MaxPool3d((1,2,2)) + ReLU in one pass
//     (`temporal_only_conv` :207-288 turns the 5x7x7/stride-2 stem into this; the stem's own 1x3x3/stride-2
//     max-pool follows as a separate af_maxpool3d).  K = kt taps x 4 padded channels <= 32: one MFMA K-block.
//   * the transformer head (time_transformer.py:8-83, 219-281) on 17 tokens x 1024 per clip, in fp32 throughout:
//     token assembly (class token + position embedding), LayerNorm, softmax attention, exact GELU.  The four
//     Linear layers run on the convolution kernel (fp32 MFMA, 1x1x1 over the token rows).
#include "af_common.h"

namespace af {

// ------------------------------------------------------------------------------------------------------
// temporal stem
struct TStemArgs {
    const char* in;      // packed clip [N][T + 2*PAD_T][H + 2*PAD_H][W + PAD_W_TOTAL][4] (af_pack_input_*)
    const char* w;       // packed A fragments [KB][4 tiles][64 lanes][16 B]
    const float* scale;
    const float* shift;
    char* out;           // [N][T][H/2][W/2][64]
    int T, Hp, Wp, Tp;   // frames; padded input dims
    int H, W, Ho, Wo;    // conv image; pooled image
    int tiles_w;         // ceil(Wo / 16)
    long long tiles;     // N * T * Ho * tiles_w
    int t_off;           // first padded frame of tap 0 for output frame 0 (PAD_T - kt/2)
    int kt;
};

// A wave owns tiles of 16 consecutive pooled pixels of one row = 2 x 32 conv positions, as FOUR position tiles: tile
// (dy, dx) holds window member (dy, dx) of each of the 16 pooled pixels, so the 2x2 max is an elementwise max of four
// accumulators (no cross-lane traffic) and every lane stores.  Lane = (fg = k-group, frow = pooled pixel in the tile).
// 16-bit: one K-block of 32 = 8 tap slots x 4 channels, lane group fg holds taps 2fg, 2fg+1.  fp32: K-blocks of
// 16 = 4 tap slots, lane group fg holds tap 4*kb + fg.
template <int DT>
__global__ __launch_bounds__(256) void tstem_kernel(const TStemArgs a) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC, ES = 16 / EPC;
    constexpr int TPC = EPC / 4;                    // taps per 16-byte chunk
    constexpr int KB = DT == AF_F32 ? 2 : 1;        // K-blocks (kt <= 5 taps + padding)
    constexpr int PIXB = 4 * ES;                    // bytes per padded pixel
    const int lane = threadIdx.x & 63, frow = lane & 15, fg = lane >> 4;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);

    uint4 wf[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[kb][i] = reinterpret_cast<const uint4*>(a.w)[(kb * 4 + i) * 64 + lane];
    f32x4 sc[4], sf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + i * 16 + fg * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + i * 16 + fg * 4);
    }
    const long long row_b = (long long)a.Wp * PIXB, plane_b = row_b * a.Hp;
    // 16-bit output: the tile's 16 pooled pixels are 2 KB of CONTIGUOUS NDHWC bytes; the accumulator layout would write
    // them as 8-byte pieces (32 contiguous bytes per pixel and instruction).  They cross a per-wave LDS patch instead
    // ([pixel][8 chunks of 16 B], chunk ^= pixel & 7) and leave as two 1-KB wave stores of whole 128-byte rows.
    __shared__ uint4 patch_all[DT == AF_F32 ? 1 : 4 * 128];
    char* patch = reinterpret_cast<char*>(patch_all) + (DT == AF_F32 ? 0 : (threadIdx.x >> 6) * 2048);

    struct Where { int tw, ph, t, pw; long long n; bool live; const char* px; };
    auto locate = [&](unsigned tile) {                                  // (tiles < 2^31: host-checked)
        Where w;
        w.tw = (int)(tile % (unsigned)a.tiles_w); unsigned q = tile / (unsigned)a.tiles_w;
        w.ph = (int)(q % (unsigned)a.Ho); q /= (unsigned)a.Ho;
        w.t = (int)(q % (unsigned)a.T); w.n = q / (unsigned)a.T;
        w.pw = w.tw * 16 + frow;                                        // pooled column of this lane
        w.live = w.pw < a.Wo;
        if (!w.live) w.pw = a.Wo - 1;                                   // ragged last tile: clamped, never stored
        w.px = a.in + ((w.n * a.Tp + w.t + a.t_off) * plane_b) + (long long)(2 * w.ph + AF_STEM_PAD_H) * row_b +
               (long long)(2 * w.pw + AF_STEM_PAD_W_LEFT) * PIXB;
        return w;
    };
    // 16-bit: the two horizontal window members are adjacent 8-byte pixels - one 16-byte load (8-byte aligned: the left
    // padding is odd) per (row, tap) serves both: 4 loads of 256 contiguous bytes per lane group instead of 8 strided
    // 8-byte ones; lane group fg holds taps 2fg, 2fg+1 (taps >= kt have zero weights and are not read).  The loads of the
    // NEXT tile are issued before this tile's MFMAs and epilogue.
    typedef u32x4 __attribute__((aligned(8))) u32x4_a8;
    auto load_taps = [&](const char* px, uint4 (&l)[2][2]) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                l[dy][tp] = uint4{0u, 0u, 0u, 0u};
                if (fg * 2 + tp < a.kt) l[dy][tp] = __builtin_bit_cast(uint4, *reinterpret_cast<const u32x4_a8*>(px + dy * row_b + (fg * 2 + tp) * plane_b));
            }
    };
    unsigned tile = (unsigned)wave0;
    Where nxt = locate(tile < (unsigned)a.tiles ? tile : 0u);
    uint4 lnext[2][2];
    if (DT != AF_F32 && tile < (unsigned)a.tiles) load_taps(nxt.px, lnext);
    for (; tile < (unsigned)a.tiles; tile += (unsigned)nwaves) {
        const Where cur = nxt;
        const int tw = cur.tw, ph = cur.ph, t = cur.t, pw = cur.pw;
        const long long n = cur.n;
        const bool live = cur.live;
        const char* px = cur.px;
        uint4 l[2][2];
        if (DT != AF_F32) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) { l[dy][0] = lnext[dy][0]; l[dy][1] = lnext[dy][1]; }
            if (tile + (unsigned)nwaves < (unsigned)a.tiles) {
                nxt = locate(tile + (unsigned)nwaves);
                load_taps(nxt.px, lnext);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if (tile + (unsigned)nwaves < (unsigned)a.tiles) {
            nxt = locate(tile + (unsigned)nwaves);
        }
        f32x4 acc[4][4];                                                // [window member][channel tile]
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            // this lane's chunk: TPC consecutive taps starting at tap0; taps >= kt have zero weights and are not read
            const int tap0 = (kb * 4 + fg) * TPC;
            if (DT == AF_F32) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const char* p = px + (m >> 1) * row_b + (m & 1) * PIXB;
                    uint4 b = uint4{0u, 0u, 0u, 0u};
                    if (tap0 < a.kt) b = *reinterpret_cast<const uint4*>(p + tap0 * plane_b);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mma<DT>::run(wf[kb][i], b, acc[m][i]);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint4 &l0 = l[m >> 1][0], &l1 = l[m >> 1][1];
                    const uint4 b = (m & 1) ? uint4{l0.z, l0.w, l1.z, l1.w} : uint4{l0.x, l0.y, l1.x, l1.y};
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mma<DT>::run(wf[kb][i], b, acc[m][i]);
                }
            }
        }
        // BN on each window member, 2x2 max (NaN propagates like ATen's max_pool), ReLU, store 4 channels per tile
        const long long opix = ((n * a.T + t) * a.Ho + ph) * a.Wo + pw;
        const long long opix0 = ((n * a.T + t) * a.Ho + ph) * a.Wo + tw * 16;      // pooled pixel 0 of the tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // max over the window of bn(x) = bn(max x) for a scale >= 0 and bn(min x) for a scale < 0 (x -> x*s + b is
            // monotonic, so the selected member's image IS the maximum image, bit for bit): one FMA instead of four and
            // NaN-propagating max / min (v_maximum3_f32 / v_minimum3_f32) on the raw accumulators: a NaN member makes the result NaN
            // like ATen's max_pool.
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = acc[0][i][e], x1 = acc[1][i][e], x2 = acc[2][i][e], x3 = acc[3][i][e];
                const float mx = max_nan(max_nan(x0, x1), max_nan(x2, x3)), mn = min_nan(min_nan(x0, x1), min_nan(x2, x3));
                v[e] = relu_f((sc[i][e] >= 0.f ? mx : mn) * sc[i][e] + sf[i][e]);
            }
            if (DT == AF_F32) {
                if (live) Vec4<DT>::store(a.out + (opix * 64 + i * 16 + fg * 4) * ES, v);
            } else {
                const int c = i * 2 + (fg >> 1);                        // 16-byte chunk of channels i*16 + fg*4 ..
                Vec4<DT>::store(patch + frow * 128 + ((c ^ (frow & 7)) * 16) + (fg & 1) * 8, v);
            }
        }
        if (DT != AF_F32) {
            __builtin_amdgcn_wave_barrier();                            // same-wave LDS ops complete in order
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int p = it * 8 + (lane >> 3), c = lane & 7;
                const uint4 o = *reinterpret_cast<const uint4*>(patch + p * 128 + ((c ^ (p & 7)) * 16));
                if (tw * 16 + p < a.Wo)
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + ((opix0 + p) * 64 + c * 8) * ES));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The temporal stem WITH the stem's own MaxPool3d([1,3,3],[1,2,2],[0,1,1]) fused behind it (round 3, 16-bit):
//     conv [kt,1,1] + BN -> MaxPool3d((1,2,2)) -> ReLU -> MaxPool3d((1,3,3), s (1,2,2), p (0,1,1))
// A final pixel (qy, qx) is the max over rows 2qy-1 .. 2qy+1, columns 2qx-1 .. 2qx+1 of the half-resolution map, each of
// whose pixels is the max over a 2x2 window of conv outputs: the max over the 6 x 6 conv positions (4qy-2 .. 4qy+3) x
// (4qx-2 .. 4qx+3) (ReLU commutes with max; BN is monotonic per channel, so the raw max / min decides, as in tstem_kernel).
// Windows overlap (stride 4, extent 6): 2.25x the conv positions are computed - the conv is one MFMA K-block per position
// tile, ~60 GFLOP per launch in all - and in exchange the 822-MB half-resolution tensor is neither written nor read back
// (tstem 0.41 ms + maxpool 0.23 ms before).  The pool's -inf padding is handled in the addressing: a window row / column
// pair outside the half-resolution map is CLAMPED onto a valid one of the same window (a duplicate member changes no max).
// Lane = (final pixel frow of a 16-pixel tile, k-group fg = taps 2fg, 2fg+1); per window row three 16-byte loads per
// tap chunk (two adjacent conv pixels each); the next row's loads (the next tile's first row behind the last one: six rows,
// an even count, so the two register sets stay static) are issued before this row's 24 MFMAs.
// Measured (B = 16 bf16): 0.39 + 0.23 ms for the two launches -> 0.53 ms.  The launch is bound by the vector ALU (max3 / min3 / NaN
// flag per member and element, fragment assembly: ~1 300 operations next to 144 MFMAs per tile) AND by the vector-memory address
// rate (36 loads per tile).  Tried and dropped: two 8-byte loads per member straight into the fragment registers (no assembly
// moves, twice the load instructions: 0.63 ms); rows two per trip with static fragment sets + a wave-uniform skip of the min (or
// max) when every BN scale has one sign (0.82 ms: the uniform conditions became branches inside the unrolled element loop).
struct TStemP3Args {
    const char* in;
    const char* w;
    const float* scale;
    const float* shift;
    char* out;           // [N][T][Hq][Wq][64]
    int T, Hp, Wp, Tp;
    int H2, W2;          // half-resolution map (h / 2, w / 2)
    int Hq, Wq;          // final map
    int tiles_w;
    long long tiles;     // N * T * Hq * tiles_w
    int t_off, kt;
};

template <int DT>
__global__ __launch_bounds__(256, 2) void tstem_pool3_kernel(const TStemP3Args a) {
    typedef Elem<DT> E;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int PIXB = 8;                                   // bytes per padded pixel (4 channels)
    const int lane = threadIdx.x & 63, frow = lane & 15, fg = lane >> 4;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    uint4 wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = reinterpret_cast<const uint4*>(a.w)[i * 64 + lane];
    f32x4 sc[4], sf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + i * 16 + fg * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + i * 16 + fg * 4);
    }
    const long long row_b = (long long)a.Wp * PIXB, plane_b = row_b * a.Hp;
    __shared__ uint4 patch_all[4 * 128];
    char* patch = reinterpret_cast<char*>(patch_all) + (threadIdx.x >> 6) * 2048;

    // a tile: wave-uniform (clip, frame, row qy, column tile tw) -> scalar base of the frame's first conv row; per lane the byte
    // offsets of its three clamped column pairs
    struct Where { int tw, qy, t; long long n; const char* base; int coff[3]; };
    auto locate = [&](unsigned tile_u) {
        const unsigned tile = (unsigned)__builtin_amdgcn_readfirstlane((int)tile_u);
        Where w;
        w.tw = (int)(tile % (unsigned)a.tiles_w); unsigned q = tile / (unsigned)a.tiles_w;
        w.qy = (int)(q % (unsigned)a.Hq); q /= (unsigned)a.Hq;
        w.t = (int)(q % (unsigned)a.T); w.n = q / (unsigned)a.T;
        int qx = w.tw * 16 + frow;
        if (qx > a.Wq - 1) qx = a.Wq - 1;                      // ragged last tile: clamped, never stored
        w.base = a.in + ((w.n * a.Tp + w.t + a.t_off) * plane_b) + (long long)AF_STEM_PAD_H * row_b + (long long)AF_STEM_PAD_W_LEFT * PIXB;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                          // half-resolution column 2qx - 1 + j, clamped into the map
            int c = 2 * qx - 1 + j;
            c = c < 0 ? 0 : (c > a.W2 - 1 ? a.W2 - 1 : c);
            w.coff[j] = 2 * c * PIXB + fg * 2 * (int)plane_b;  // conv columns 2c, 2c + 1 (one 16-byte load) of tap 2 fg
        }
        return w;
    };
    typedef u32x4 __attribute__((aligned(8))) u32x4_a8;
    // window row dy of a tile: conv row 4qy - 2 + dy, clamped into the conv rows of the window's valid half-resolution rows
    auto load_row = [&](const char* base, int qy, const int (&coff)[3], int dy, uint4 (&l)[3][2]) {
        const int rlo = 2 * qy - 1 < 0 ? 0 : 2 * qy - 1, rhi = 2 * qy + 1 > a.H2 - 1 ? a.H2 - 1 : 2 * qy + 1;
        int y = 4 * qy - 2 + dy;
        y = y < 2 * rlo ? 2 * rlo : (y > 2 * rhi + 1 ? 2 * rhi + 1 : y);
        const char* rowp = base + (long long)y * row_b;        // uniform
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                l[j][tp] = uint4{0u, 0u, 0u, 0u};
                if (fg * 2 + tp < a.kt) l[j][tp] = __builtin_bit_cast(uint4, *reinterpret_cast<const u32x4_a8*>(rowp + (unsigned)(coff[j] + tp * (int)plane_b)));
            }
    };
    unsigned tile = (unsigned)wave0;
    if (tile >= (unsigned)a.tiles) return;
    Where cur = locate(tile);
    uint4 rowc[3][2], rown[3][2];
    load_row(cur.base, cur.qy, cur.coff, 0, rowc);
    for (; tile < (unsigned)a.tiles; tile += (unsigned)nwaves) {
        float mx[4][4], mn[4][4];                              // NaN-propagating max / min (v_maximum3_f32 / v_minimum3_f32): a NaN member
#pragma unroll                                                 // makes the result NaN, like ATen's max_pool
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { mx[i][e] = -INFINITY; mn[i][e] = INFINITY; }
        const unsigned ntile = tile + (unsigned)nwaves < (unsigned)a.tiles ? tile + (unsigned)nwaves : tile;
        const Where nxt = locate(ntile);
#pragma unroll 1
        for (int dy = 0; dy < 6; ++dy) {
            // the next window row (behind the last one: row 0 of the next tile) is fetched while this one multiplies
            if (dy < 5) load_row(cur.base, cur.qy, cur.coff, dy + 1, rown);
            else load_row(nxt.base, nxt.qy, nxt.coff, 0, rown);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint4 &l0 = rowc[j][0], &l1 = rowc[j][1];
                const uint4 blo = uint4{l0.x, l0.y, l1.x, l1.y}, bhi = uint4{l0.z, l0.w, l1.z, l1.w};
                f32x4 lo[4], hi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = f32x4{0.f, 0.f, 0.f, 0.f}; hi[i] = lo[i];
                    Mma<DT>::run(wf[i], blo, lo[i]);
                    Mma<DT>::run(wf[i], bhi, hi[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xl = lo[i][e], xh = hi[i][e];
                        // (builtins, not inline asm: hipcc does not space an asm statement behind the MFMA whose result it reads)
                        mx[i][e] = max_nan(mx[i][e], max_nan(xl, xh));
                        mn[i][e] = min_nan(mn[i][e], min_nan(xl, xh));
                    }
                __builtin_amdgcn_sched_barrier(0);                 // one pair's 8 accumulators live at a time
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) { rowc[j][0] = rown[j][0]; rowc[j][1] = rown[j][1]; }
        }
        // BN on the deciding member (max for a scale >= 0, min below), NaN like ATen's max_pool, ReLU, 16 pixels x 128 B out
        const long long opix0 = ((cur.n * a.T + cur.t) * a.Hq + cur.qy) * a.Wq + cur.tw * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = relu_f((sc[i][e] >= 0.f ? mx[i][e] : mn[i][e]) * sc[i][e] + sf[i][e]);
            }
            const int c = i * 2 + (fg >> 1);
            Vec4<DT>::store(patch + frow * 128 + ((c ^ (frow & 7)) * 16) + (fg & 1) * 8, v);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int p = it * 8 + (lane >> 3), c = lane & 7;
            const uint4 o = *reinterpret_cast<const uint4*>(patch + p * 128 + ((c ^ (p & 7)) * 16));
            if (cur.tw * 16 + p < a.Wq)
                __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + ((opix0 + p) * 64 + c * 8) * 2));
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
    }
}

__global__ void pack_tstem_weight_kernel(const float* w, int kt, int dtype, char* out) {
    // out[kb][tile][lane][16 B]; lane = fg*16 + frow -> channel tile*16 + frow, chunk = taps tap0.. x 4 channels
    const int epc = dtype == AF_F32 ? 4 : 8, tpc = epc / 4, kbs = dtype == AF_F32 ? 2 : 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // (kb, tile, lane, element)
    if (idx >= kbs * 4 * 64 * epc) return;
    const int e = idx % epc, lane = (idx / epc) % 64, tile = (idx / epc / 64) % 4, kb = idx / epc / 64 / 4;
    const int frow = lane & 15, fg = lane >> 4;
    const int tap = (kb * 4 + fg) * tpc + e / 4, ch = e % 4, co = tile * 16 + frow;
    const float v = (tap < kt && ch < 3) ? w[(co * 3 + ch) * kt + tap] : 0.f;
    if (dtype == AF_F32) reinterpret_cast<float*>(out)[idx] = v;
    else if (dtype == AF_BF16) reinterpret_cast<__bf16*>(out)[idx] = (__bf16)v;
    else reinterpret_cast<_Float16*>(out)[idx] = (_Float16)v;
}

// ------------------------------------------------------------------------------------------------------
// transformer head (fp32)
// tokens[b][0] = cls + pos[0];  tokens[b][1 + t] = pooled[b][t] + pos[1 + t]     (time_transformer.py:270-273)
__global__ void tokens_assemble_kernel(const float* pooled, const float* cls, const float* pos, int n_tok, int dim,
                                       long long total, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % dim); const long long row = i / dim;
    const int t = (int)(row % (n_tok + 1)); const long long b = row / (n_tok + 1);
    const float x = t == 0 ? cls[d] : pooled[(b * n_tok + (t - 1)) * dim + d];
    out[i] = x + pos[(long long)t * dim + d];
}

// nn.LayerNorm(dim, eps 1e-5): one wave per row; two-pass mean / variance in fp32 like ATen
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, long long x_stride, const float* gamma,
                                                        const float* beta, int rows, int dim, float eps, float* y,
                                                        long long y_stride) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * x_stride;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += xr[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / dim;
    float v = 0.f;
    for (int i = lane; i < dim; i += 64) { const float d = xr[i] - mean; v += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = rsqrtf(v / dim + eps);
    float* yr = y + row * y_stride;
    for (int i = lane; i < dim; i += 64) yr[i] = (xr[i] - mean) * rstd * gamma[i] + beta[i];
}

// softmax(q k^T * scale) v for a handful of tokens: one workgroup per (clip, head); qkv rows are [q | k | v], each
// heads * dim_head wide (Attention.forward, time_transformer.py:52-71: 'b n (h d) -> b h n d')
constexpr int ATT_MAX_TOK = 64, ATT_MAX_DH = 128;
__global__ __launch_bounds__(128) void attention_kernel(const float* qkv, int n_tok, int heads, int dh, float scale,
                                                        float* out) {
    extern __shared__ float sm[];                      // q,k,v [n_tok][dh] each, then scores [n_tok][n_tok]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    const int inner = heads * dh;
    float* q = sm; float* k = q + n_tok * dh; float* v = k + n_tok * dh; float* p = v + n_tok * dh;
    for (int i = tid; i < n_tok * dh; i += blockDim.x) {
        const int t = i / dh, d = i % dh;
        const float* row = qkv + ((long long)b * n_tok + t) * 3 * inner + h * dh + d;
        q[i] = row[0]; k[i] = row[inner]; v[i] = row[2 * inner];
    }
    __syncthreads();
    for (int i = tid; i < n_tok * n_tok; i += blockDim.x) {
        const int r = i / n_tok, c = i % n_tok;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += q[r * dh + d] * k[c * dh + d];
        p[i] = s * scale;
    }
    __syncthreads();
    for (int r = tid; r < n_tok; r += blockDim.x) {
        float m = -INFINITY;
        for (int c = 0; c < n_tok; ++c) m = fmaxf(m, p[r * n_tok + c]);
        float z = 0.f;
        for (int c = 0; c < n_tok; ++c) { const float e = expf(p[r * n_tok + c] - m); p[r * n_tok + c] = e; z += e; }
        const float inv = 1.f / z;
        for (int c = 0; c < n_tok; ++c) p[r * n_tok + c] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < n_tok * dh; i += blockDim.x) {
        const int r = i / dh, d = i % dh;
        float s = 0.f;
        for (int c = 0; c < n_tok; ++c) s += p[r * n_tok + c] * v[c * dh + d];
        out[((long long)b * n_tok + r) * inner + h * dh + d] = s;        // 'b h n d -> b n (h d)'
    }
}

// nn.GELU() (exact, erf form), in place
__global__ void gelu_kernel(float* x, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; x[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
}

}  // namespace af

extern "C" int af_pack_tstem_weight(const float* w_oidhw, int cout, int kt, int dtype, void* out, void* stream) {
    using namespace af;
    AF_REQUIRE(w_oidhw && out && dtype_ok(dtype), "pack_tstem_weight: bad argument");
    AF_REQUIRE(cout == 64 && kt >= 1 && kt <= 2 * AF_STEM_PAD_T + 1 && (kt & 1), "pack_tstem_weight: expects 64 x 3 x kt x 1 x 1, kt odd <= %d", 2 * AF_STEM_PAD_T + 1);
    const int total = (dtype == AF_F32 ? 2 : 1) * 4 * 64 * (dtype == AF_F32 ? 4 : 8);
    hipLaunchKernelGGL(pack_tstem_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oidhw, kt, dtype, (char*)out);
    AF_CHECK_LAUNCH("pack_tstem_weight_kernel");
    return AF_OK;
}

extern "C" long long af_packed_tstem_weight_bytes(int dtype) { return (dtype == AF_F32 ? 2 : 1) * 4 * 64 * 16; }

extern "C" int af_tstem_conv_bn_pool_relu(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                          const float* scale, const float* shift, void* out, void* stream) {
    using namespace af;
    AF_REQUIRE(d && stem_in && w_packed && scale && shift && out, "tstem: null argument");
    AF_REQUIRE(dtype_ok(d->dtype), "tstem: bad dtype %d", d->dtype);
    AF_REQUIRE(d->cin == 3 && d->cout == 64 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 &&
                   d->ph == 0 && d->pw == 0, "tstem: expects Conv3d(3->64, [kt,1,1], stride 1)");
    AF_REQUIRE(d->kt >= 1 && d->kt <= 2 * AF_STEM_PAD_T + 1 && (d->kt & 1) && d->pt == d->kt / 2, "tstem: bad kt/pt");
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h >= 2 && d->w >= 2, "tstem: bad dims");
    AF_REQUIRE(d->to == d->t && d->ho == d->h / 2 && d->wo == d->w / 2, "tstem: output dims are (t, h/2, w/2): the 2x2 pool is fused");
    AF_REQUIRE(aligned16(stem_in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out),
               "tstem: buffers must be 16-byte aligned");
    TStemArgs a;
    a.in = (const char*)stem_in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.T = d->t; a.Tp = d->t + 2 * AF_STEM_PAD_T; a.Hp = d->h + 2 * AF_STEM_PAD_H; a.Wp = d->w + AF_STEM_PAD_W_TOTAL;
    a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.tiles_w = (d->wo + 15) / 16;
    a.tiles = (long long)d->n * d->t * d->ho * a.tiles_w;
    a.t_off = AF_STEM_PAD_T - d->pt; a.kt = d->kt;
    AF_REQUIRE(a.tiles < (1LL << 31), "tstem: %lld tiles", a.tiles);
    long long blocks = (a.tiles + 4 * 4 - 1) / (4 * 4);                   // ~4 tiles per wave
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    switch (d->dtype) {
        case AF_F32: hipLaunchKernelGGL((tstem_kernel<AF_F32>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case AF_BF16: hipLaunchKernelGGL((tstem_kernel<AF_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((tstem_kernel<AF_F16>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    }
    AF_CHECK_LAUNCH("tstem_kernel");
    return AF_OK;
}

extern "C" int af_tstem_conv_bn_pool_relu_maxpool(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                                  const float* scale, const float* shift, void* out, void* stream) {
    using namespace af;
    AF_REQUIRE(d && stem_in && w_packed && scale && shift && out, "tstem_pool3: null argument");
    AF_REQUIRE(d->dtype == AF_BF16 || d->dtype == AF_F16, "tstem_pool3: 16-bit dtypes only");
    AF_REQUIRE(d->cin == 3 && d->cout == 64 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 &&
                   d->ph == 0 && d->pw == 0, "tstem_pool3: expects Conv3d(3->64, [kt,1,1], stride 1)");
    AF_REQUIRE(d->kt >= 1 && d->kt <= 2 * AF_STEM_PAD_T + 1 && (d->kt & 1) && d->pt == d->kt / 2, "tstem_pool3: bad kt/pt");
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h >= 2 && d->w >= 2, "tstem_pool3: bad dims");
    AF_REQUIRE(d->to == d->t && d->ho == d->h / 2 && d->wo == d->w / 2, "tstem_pool3: d describes the conv with the 2x2 pool (to, ho, wo = t, h/2, w/2)");
    AF_REQUIRE(aligned16(stem_in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out),
               "tstem_pool3: buffers must be 16-byte aligned");
    TStemP3Args a;
    a.in = (const char*)stem_in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.T = d->t; a.Tp = d->t + 2 * AF_STEM_PAD_T; a.Hp = d->h + 2 * AF_STEM_PAD_H; a.Wp = d->w + AF_STEM_PAD_W_TOTAL;
    a.H2 = d->ho; a.W2 = d->wo; a.Hq = (d->ho - 1) / 2 + 1; a.Wq = (d->wo - 1) / 2 + 1;
    a.tiles_w = (a.Wq + 15) / 16;
    a.tiles = (long long)d->n * d->t * a.Hq * a.tiles_w;
    a.t_off = AF_STEM_PAD_T - d->pt; a.kt = d->kt;
    AF_REQUIRE(a.tiles < (1LL << 31), "tstem_pool3: %lld tiles", a.tiles);
    long long blocks = (a.tiles + 4 * 4 - 1) / (4 * 4);                   // ~4 tiles per wave
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == AF_BF16) hipLaunchKernelGGL((tstem_pool3_kernel<AF_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((tstem_pool3_kernel<AF_F16>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    AF_CHECK_LAUNCH("tstem_pool3_kernel");
    return AF_OK;
}

extern "C" int af_tokens_assemble(const float* pooled, const float* cls_token, const float* pos_embedding, int clips,
                                  int n_tok, int dim, float* out, void* stream) {
    using namespace af;
    AF_REQUIRE(pooled && cls_token && pos_embedding && out && clips >= 0 && n_tok > 0 && dim > 0, "tokens_assemble: bad argument");
    const long long total = (long long)clips * (n_tok + 1) * dim;
    if (total == 0) return AF_OK;
    hipLaunchKernelGGL(tokens_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pooled, cls_token, pos_embedding, n_tok, dim, total, out);
    AF_CHECK_LAUNCH("tokens_assemble_kernel");
    return AF_OK;
}

extern "C" int af_layernorm(const float* x, long long x_row_stride, const float* gamma, const float* beta, int rows,
                            int dim, float eps, float* y, long long y_row_stride, void* stream) {
    using namespace af;
    AF_REQUIRE(x && gamma && beta && y && rows >= 0 && dim > 0, "layernorm: bad argument");
    if (rows == 0) return AF_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, x_row_stride, gamma,
                       beta, rows, dim, eps, y, y_row_stride);
    AF_CHECK_LAUNCH("layernorm_kernel");
    return AF_OK;
}

extern "C" int af_attention(const float* qkv, int clips, int n_tok, int heads, int dim_head, float* out, void* stream) {
    using namespace af;
    AF_REQUIRE(qkv && out && clips >= 0 && heads > 0, "attention: bad argument");
    AF_REQUIRE(n_tok > 0 && n_tok <= ATT_MAX_TOK && dim_head > 0 && dim_head <= ATT_MAX_DH,
               "attention: at most %d tokens x %d per head (got %d x %d)", ATT_MAX_TOK, ATT_MAX_DH, n_tok, dim_head);
    if (clips == 0) return AF_OK;
    const int lds = (3 * n_tok * dim_head + n_tok * n_tok) * 4;
    AF_SET_MAX_LDS(&attention_kernel, (3 * ATT_MAX_TOK * ATT_MAX_DH + ATT_MAX_TOK * ATT_MAX_TOK) * 4, "attention");
    hipLaunchKernelGGL(attention_kernel, dim3(clips * heads), dim3(128), lds, (hipStream_t)stream, qkv, n_tok, heads, dim_head,
                       1.0f / sqrtf((float)dim_head), out);
    AF_CHECK_LAUNCH("attention_kernel");
    return AF_OK;
}

extern "C" int af_gelu(float* x, long long n, void* stream) {
    using namespace af;
    AF_REQUIRE(x && n >= 0, "gelu: bad argument");
    if (n == 0) return AF_OK;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n);
    AF_CHECK_LAUNCH("gelu_kernel");
    return AF_OK;
}

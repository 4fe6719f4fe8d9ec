// 1x1x1 / stride 1 convolution with a SHORT K (64 or 128 input channels in all, optionally split over two inputs: the
// projection shortcut accumulated into the same tile) into 256 x n channels + BN [+ residual] [+ ReLU] [+ max over
// frame pairs]: the `c` convs of the s2 / s3 bottlenecks (reference altfreezing/slowfast/models/resnet_helper.py:
// 304-325, 411-444 and the MaxPool3d after s2, video_model_builder.py pathway0_pool).
//
// These layers are streams: per output position 128-256 B of activations come in, 512-1024 B of residual come in and
// as much goes out, against 16-32 MFMAs per wave.  In the generic implicit GEMM a workgroup lives for one tile: it
// waits a full memory latency for its operands, a second one for the residual inside the epilogue, and re-fetches
// the weights (twice the bytes of the activations it multiplies) from L2.  Here
//   * workgroups are persistent (one per CU) and every wave keeps its 64 output channels x K of WEIGHTS IN REGISTERS;
//   * a tile is 128 positions; its activations arrive by LDS-DMA into a 2-slot ring and its residual rows by plain
//     16-byte loads into registers, both issued ONE TILE AHEAD: while tile q is multiplied, transposed and stored,
//     ~100-150 KB of loads for tile q+1 are in flight per CU, which is what the HBM stream needs (Little's law);
//   * the epilogue is the usual per-wave patch: fp32 accumulators -> LDS -> whole 128-byte row segments with
//     BN, residual, ReLU, the frame-pair max and the one rounding applied on the way out.
// 8 waves = 4 channel groups x 2 position halves, wave tile 64 channels x 64 positions.
//
// Temporal pool (tpool): tile = (clip, frame pair, 64 pixels).  Rows are ordered so that the two frames of a pixel
// are accumulator tiles j and j + 2 of the SAME lane: the pair max needs no exchange at all.
#include "af_common.h"

namespace af {

struct C111Args {
    const char* in;
    const char* in2;
    const char* w;       // packed [Cout][Cin]
    const char* w2;      // packed [Cout][Cin2]
    const float* scale;
    const float* shift;
    const char* res;
    char* out;
    int Cin, Cin2, Cout, out_ld, relu;
    int ncol;            // Cout / 256: channel columns, each its own workgroup stream
    int T, HW, chunks;   // tpool: frames, pixels per frame, 64-pixel chunks per frame
    long long M;         // input positions
    int tiles;           // position tiles
};

// WC = output channels per wave: 64 (4 channel groups x 2 position halves) or, for K = 256 where 64 channels of
// weights would not fit the registers next to two residual sets, 32 (8 channel groups, every wave all 128 positions).
template <int DT, int KS1, int KS2, bool TPOOL, bool RES, int WC>
__global__ __launch_bounds__(512, 1) void conv111_kernel(const C111Args a) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC, ES = 16 / EPC;
    static_assert(EPC == 8, "16-bit storage types only");
    static_assert(WC == 64 || (WC == 32 && !TPOOL), "wave columns of 64 or 32 channels; the pooled tile order assumes 64");
    constexpr int KS = KS1 + KS2, BM = 128, TN = WC / 16;
    constexpr int NWN = 256 / WC, NWM = 8 / NWN, WPOS = BM / NWM, MT = WPOS / 16;   // wave grid; positions, m-tiles per wave
    constexpr int SLAB = BM * 128, STAGE = KS * SLAB;  // bytes: one 64-channel K slab of the tile; one ring slot
    constexpr int PROW = WC + 4;                       // patch row stride in floats (pad: conflict-free b128 writes)
    constexpr int LPR = WC / 8, RPI = 64 / LPR, ITS = 16 / RPI;   // epilogue: lanes per row, rows per instruction, instructions per m-tile

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int wn = wave % NWN, wm = wave / NWN;        // channel group, position part
    float* patch = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + 2 * STAGE) + wave * (16 * PROW);
    // workgroup -> (channel column, tile stream).  Consecutive workgroup ids go round the 8 XCDs, each with its own
    // L2: the columns of one stream (they read the same activation tiles at the same time) are put on ONE XCD, so the
    // tile comes from HBM once and from that L2 ncol - 1 times (measured before: s3 `c` fetched its activations twice).
    int col = blockIdx.x % a.ncol, first = blockIdx.x / a.ncol;
    const int stride = gridDim.x / a.ncol;
    if (gridDim.x % (8 * a.ncol) == 0) {
        const int xcd = blockIdx.x & 7, y = blockIdx.x >> 3;
        col = y % a.ncol;
        first = xcd + 8 * (y / a.ncol);
    }

    // tile row r -> position offset from the tile origin.  plain: r.  tpool: r = (half b, tile j, lane row fr) is
    // pixel b*32 + (j&1)*16 + fr of frame j>>1.
    auto row_pixel = [&](int r) { return (r >> 6) * 32 + ((r >> 4) & 1) * 16 + (r & 15); };
    auto row_off = [&](int r) -> long long { return TPOOL ? (long long)((r >> 5) & 1) * a.HW + row_pixel(r) : r; };

    // ---- weights: 4 channel tiles x (2 KS) 32-wide K chunks, fetched once
    uint4 wreg[TN][2 * KS];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const long long ch = col * 256 + wn * WC + i * 16 + frow;
#pragma unroll
        for (int k = 0; k < 2 * KS1; ++k)
            wreg[i][k] = *reinterpret_cast<const uint4*>(a.w + (ch * a.Cin + k * 32 + fg * 8) * ES);
#pragma unroll
        for (int k = 0; k < 2 * KS2; ++k)
            wreg[i][2 * KS1 + k] = *reinterpret_cast<const uint4*>(a.w2 + (ch * a.Cin2 + k * 32 + fg * 8) * ES);
    }

    // ---- epilogue geometry: lane = (row rr of RPI, 8 channels at cc) of a 16-row patch
    const int rr = lane / LPR, cc = (lane % LPR) * 8;
    const int ch0 = col * 256 + wn * WC + cc;
    // BN scale / shift of this lane's 8 channels: registers (read from LDS per use they were a third of the kernel's LDS
    // traffic; the two-pass accumulators left the room)
    f32x4 sc[2], sf[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        sc[e] = *reinterpret_cast<const f32x4*>(a.scale + ch0 + 4 * e);
        sf[e] = *reinterpret_cast<const f32x4*>(a.shift + ch0 + 4 * e);
    }

    // ---- producer: per-lane source offsets of the 2 DMA pieces (8 rows each) this wave brings in per K slab
    const int drow = lane >> 3, chunk = (lane & 7) ^ drow;
    // (piece 1 is piece 0 shifted by 64 tile rows: a scalar offset, like the K slab)
    const int xrow = wave * 8 + drow;
    const unsigned xoff = (unsigned)(row_off(xrow) * a.Cin * ES + chunk * 16);
    const unsigned x2off = KS2 ? (unsigned)(row_off(xrow) * a.Cin2 * ES + chunk * 16) : 0u;
    const int half_off = (int)(row_off(64) * a.Cin * ES), half2_off = (int)(row_off(64) * a.Cin2 * ES);
    // origin position of a tile, and how many of its rows / pixels exist
    auto tile_origin = [&](int tile, int& live) -> long long {
        if (TPOOL) {
            const int c = tile % a.chunks, nt = tile / a.chunks;            // nt = n * (T/2) + frame pair
            live = a.HW - c * 64;                                          // pixels of the chunk that exist
            return (long long)nt * 2 * a.HW + c * 64;
        }
        const long long o = (long long)tile * BM;
        live = (int)(a.M - o < BM ? a.M - o : BM);
        return o;
    };
    auto row_live = [&](int r, int live) { return TPOOL ? row_pixel(r) < live : r < live; };

    auto issue_tile = [&](int tile, int slot) {
        int live;
        const long long o = tile_origin(tile, live);
        const i32x4 d1 = make_desc(a.in + o * a.Cin * ES);
        const unsigned base = lds0 + slot * STAGE + wave * (8 * 128);
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                blds16(row_live(xrow + 64 * i, live) ? xoff : kOutOfRange, d1, s * 128 + i * half_off, base + s * SLAB + i * (64 * 128));
        if (KS2) {
            const i32x4 d2 = make_desc(a.in2 + o * a.Cin2 * ES);
#pragma unroll
            for (int s = 0; s < KS2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    blds16(row_live(xrow + 64 * i, live) ? x2off : kOutOfRange, d2, s * 128 + i * half2_off, base + (KS1 + s) * SLAB + i * (64 * 128));
        }
    };
    // residual / output rows of a tile in epilogue order: k = (j, it) is tile row wm*WPOS + j*16 + it*RPI + rr, channels
    // ch0 .. ch0+7.  Buffer addressing from the tile's origin: one per-lane offset (row 0 of the lane) plus a per-k
    // scalar, so no 64-bit address lives in a register across the loop; residual rows that do not exist get kOutOfRange
    // (the load returns zeros).
    const int row0 = wm * WPOS + rr;
    const unsigned res_lane = (unsigned)((row_off(row0) * a.Cout + ch0) * ES);
    const unsigned out_lane = (unsigned)(((long long)(TPOOL ? row_pixel(row0) : row0) * a.out_ld + ch0) * ES);
    auto k_row = [&](int k) { return (k / ITS) * 16 + (k % ITS) * RPI; };             // tile row of k relative to row0
    auto res_soff = [&](int k) { return (int)((row_off(k_row(k)) * a.Cout) * ES); };  // (row_off is additive over these bits)
    auto out_soff = [&](int k) { return (int)(((long long)(TPOOL ? row_pixel(k_row(k)) : k_row(k)) * a.out_ld) * ES); };
    auto load_residual = [&](int tile, u32x4 (&r)[8]) {
        int live;
        const long long o = tile_origin(tile, live);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.res + o * a.Cout * ES), (short)0, (int)kOutOfRange, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            r[k] = __builtin_amdgcn_raw_buffer_load_b128(rd, row_live(row0 + k_row(k), live) ? res_lane : kOutOfRange, res_soff(k), 2);
    };

    const int my_tiles = first < a.tiles ? (a.tiles - first + stride - 1) / stride : 0;
    u32x4 rcur[8], rnext[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rnext[k] = u32x4{0u, 0u, 0u, 0u};
    if (my_tiles > 0) {
        issue_tile(first, 0);
        if (RES) load_residual(first, rnext);
    }
    __builtin_amdgcn_sched_barrier(0);

    for (int q = 0; q < my_tiles; ++q) {
        const int tile = first + q * stride, slot = q & 1;
        wait_vmcnt<0>();                                 // tile q's activations and residual rows have landed ...
        __builtin_amdgcn_s_barrier();                    // ... for every wave, and nobody still reads slot (q+1)&1
        if (RES) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                rcur[k] = rnext[k];
                asm volatile("" : "+v"(rcur[k]));        // the copy happens here, behind the wait above
            }
        }
        if (q + 1 < my_tiles) {                          // one tile ahead: DMA the activations, load the residual rows
            issue_tile(tile + stride, slot ^ 1);
            if (RES) load_residual(tile + stride, rnext);
        }
        __builtin_amdgcn_sched_barrier(0);

        int live;
        const long long o = tile_origin(tile, live);
        // first output position of the tile (pooled: (clip, frame pair) * HW + first pixel = half the origin's frame index)
        const long long opos = TPOOL ? (o - o % a.HW) / 2 + o % a.HW : o;
        char* obase = a.out + opos * a.out_ld * ES;
        const uint4* xs = smem + slot * (STAGE / 16) + (wm * WPOS + frow) * 8;

        // the wave's positions go in passes of 2 accumulator tiles (registers: the weights and two residual sets stay
        // live); tpool: a pass is the two frames (tiles h, h + 2) of 16 pixels
#pragma unroll
        for (int h = 0; h < MT / 2; ++h) {
            f32x4 acc[TN][2];
#pragma unroll
            for (int i = 0; i < TN; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int k = 0; k < 2 * KS; ++k) {
                const int c = ((k & 1) * 4 + fg) ^ (frow & 7);
                uint4 bf[2];
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) bf[jl] = xs[(k >> 1) * (SLAB / 16) + (TPOOL ? h + 2 * jl : 2 * h + jl) * 16 * 8 + c];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl) Mma<DT>::run(wreg[i][k], bf[jl], acc[i][jl]);
            }

            // ---- epilogue: 16 positions at a time through the wave's patch
            float keep[ITS][8];                          // tpool: frame 0 of the pixel rows, waiting for frame 1
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                const int j = TPOOL ? h + 2 * jl : 2 * h + jl;
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    *reinterpret_cast<f32x4*>(patch + frow * PROW + i * 16 + fg * 4) = acc[i][jl];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    const int prow = it * RPI + rr, row = wm * WPOS + j * 16 + prow;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(patch + prow * PROW + cc + e);
                        const f32x4 r = t * sc[e >> 2] + sf[e >> 2];
                        v[e] = r[0]; v[e + 1] = r[1]; v[e + 2] = r[2]; v[e + 3] = r[3];
                    }
                    if (RES) {
                        const uint4 rraw = __builtin_bit_cast(uint4, rcur[j * ITS + it]);
                        const typename E::type* re = reinterpret_cast<const typename E::type*>(&rraw);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += E::to_f32(re[e]);
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
                    }
                    if (TPOOL && jl == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) keep[it][e] = v[e];
                        continue;
                    }
                    if (TPOOL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (v[e] > keep[it][e] || v[e] != v[e]) ? v[e] : keep[it][e];   // NaN propagates like ATen's max_pool
                    }
                    uint4 ov;
                    typename E::type* oe = reinterpret_cast<typename E::type*>(&ov);
#pragma unroll
                    for (int e = 0; e < 8; ++e) oe[e] = E::from_f32(v[e]);
                    // (a plain store from a uniform base + 32-bit lane offset.  NOT buffer_store_dwordx4 with an SGPR offset: the
                    // compiler schedules the next VALU write of the data registers right behind it, and on gfx950 the store
                    // then reads the overwritten dword - measured; it knows the hazard only for immediate offsets)
                    if (row_live(row, live))
                        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, ov), reinterpret_cast<u32x4*>(obase + out_soff(j * ITS + it) + out_lane));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

template <int DT, int KS1, int KS2, bool TPOOL, bool RES, int WC = 64>
static int launch111(const C111Args& a, int blocks, hipStream_t stream) {
    const int lds = 2 * (KS1 + KS2) * 128 * 128 + 8 * 16 * (WC + 4) * 4;
    AF_SET_MAX_LDS((&conv111_kernel<DT, KS1, KS2, TPOOL, RES, WC>), lds, "conv111");
    hipLaunchKernelGGL((conv111_kernel<DT, KS1, KS2, TPOOL, RES, WC>), dim3(blocks), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv111_kernel");
    return AF_OK;
}

// position tiles of the layer if it takes this path, 0 otherwise
static long long conv111_tiles(const af_conv_desc* d, const af_conv_desc* d2, int out_ld) {
    if (d->dtype == AF_F32) return 0;
    if (d->kt != 1 || d->kh != 1 || d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt || d->ph || d->pw) return 0;
    if (d->cout % 256 != 0 || (d->tpool != 0 && d->tpool != 1)) return 0;
    if (d2) {
        if (d->cin != 64 || d2->cin != 64 || d->tpool) return 0;
        if (d2->st != 1 || d2->sh != 1 || d2->sw != 1 || d2->t != d->t || d2->h != d->h || d2->w != d->w) return 0;
    } else if (d->cin != 64 && d->cin != 128 && d->cin != 256) return 0;
    if (d->tpool && (d->cin != 64 || d->t % 2 != 0)) return 0;
    const long long hw = (long long)d->h * d->w;
    if ((hw + 128) * 128 * 2 * 2 >= (1LL << 31)) return 0;                 // 32-bit row offsets inside a tile
    const long long tiles = d->tpool ? (long long)d->n * (d->t / 2) * ((hw + 63) / 64) : ((long long)d->n * d->t * hw + 127) / 128;
    if (tiles >= (1LL << 30)) return 0;
    // persistent streams only pay with several tiles per workgroup (one clip of the deep stages stays on the generic path)
    if (tiles * (d->cout / 256) < 4LL * device_cus()) return 0;
    return tiles;
}

bool conv111_applies(const af_conv_desc* d, const af_conv_desc* d2, const void* residual, int out_ld) {
    if (d2 && residual) return false;
    return conv111_tiles(d, d2, out_ld) != 0;
}

int conv111_run(const af_conv_desc* d, const void* in, const void* w_packed, const af_conv_desc* d2, const void* in2,
                const void* w2_packed, const float* scale, const float* shift, const void* residual, void* out, int out_ld,
                hipStream_t stream) {
    C111Args a;
    a.in = (const char*)in; a.in2 = (const char*)in2; a.w = (const char*)w_packed; a.w2 = (const char*)w2_packed;
    a.scale = scale; a.shift = shift; a.res = (const char*)residual; a.out = (char*)out;
    a.Cin = d->cin; a.Cin2 = d2 ? d2->cin : 0; a.Cout = d->cout; a.out_ld = out_ld; a.relu = d->relu;
    a.ncol = d->cout / 256;
    a.T = d->t; a.HW = d->h * d->w; a.chunks = (a.HW + 63) / 64;
    a.M = (long long)d->n * d->t * a.HW;
    a.tiles = (int)conv111_tiles(d, d2, out_ld);
    int streams = device_cus() / a.ncol;
    if (streams > a.tiles) streams = a.tiles;
    const int blocks = streams * a.ncol;
    const bool bf = d->dtype == AF_BF16;
#define AF_C111(K1, K2, TP, RS) (bf ? launch111<AF_BF16, K1, K2, TP, RS>(a, blocks, stream) : launch111<AF_F16, K1, K2, TP, RS>(a, blocks, stream))
    if (d2) return AF_C111(1, 1, false, false);
    if (d->tpool) return residual ? AF_C111(1, 0, true, true) : AF_C111(1, 0, true, false);
    if (d->cin == 64) return residual ? AF_C111(1, 0, false, true) : AF_C111(1, 0, false, false);
    if (d->cin == 256)     // 32-channel wave columns: the weights of 64 channels x 256 would not fit the registers
        return residual ? (bf ? launch111<AF_BF16, 4, 0, false, true, 32>(a, blocks, stream) : launch111<AF_F16, 4, 0, false, true, 32>(a, blocks, stream))
                        : (bf ? launch111<AF_BF16, 4, 0, false, false, 32>(a, blocks, stream) : launch111<AF_F16, 4, 0, false, false, 32>(a, blocks, stream));
    return residual ? AF_C111(2, 0, false, true) : AF_C111(2, 0, false, false);
#undef AF_C111
}

}  // namespace af

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
//   * a tile is 128 positions (K = 64) or 64 (K = 128 / 256); its activations arrive by LDS-DMA into a ring of 2 / 4 slots and
//     its residual rows by 16-byte loads into registers, both issued ONE / THREE TILES AHEAD: while tile q is multiplied,
//     transposed and stored, ~100-150 KB of loads for the next tiles are in flight per CU, which is what the HBM stream
//     needs (Little's law).  Short tiles + a deeper ring keep that amount in flight CONTINUOUSLY (a 128-position tile
//     of K = 256 issued 128 KB at its start, which had landed half-way through its 6 us of MFMA + epilogue: by ablation
//     memory alone 50 us, compute alone 39 us, together 68 us; now 59 us).  All waits are counted s_waitcnt vmcnt: the
//     residual loads are issued through inline asm so that hipcc does not count them (bload16_nt_uncounted, af_common.h);
//   * the epilogue is the usual per-wave patch: fp32 accumulators -> LDS -> whole 128-byte row segments with
//     BN, residual, ReLU, the frame-pair max and the one rounding applied on the way out.
// 8 waves = 4 channel groups x 2 position halves, wave tile 64 channels x 64 positions.
//
// Temporal pool (tpool): tile = (clip, frame pair, 64 pixels).  Rows are ordered so that the two frames of a pixel
// are accumulator tiles j and j + 2 of the SAME lane: the pair max needs no exchange at all.
#include "af_common.h"
#include <stdlib.h>

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

// s_waitcnt vmcnt(nl * NL + ns * NS) for uniform run-time nl in [0, D-1], ns in [0, D] (the immediate is a compile-time constant)
template <int NL, int NS, int D, int A = 0, int B = 0>
__device__ __forceinline__ void wait_counted(int nl, int ns) {
    if constexpr (A <= D - 1 && B <= D) {
        if (nl == A && ns == B) { wait_vmcnt<A * NL + B * NS>(); return; }
        if constexpr (B < D) wait_counted<NL, NS, D, A, B + 1>(nl, ns);
        else wait_counted<NL, NS, D, A + 1, 0>(nl, ns);
    } else {
        wait_vmcnt<0>();
    }
}

// WC = output channels per wave: 64 (4 channel groups x 2 position halves) or, for K = 256 where 64 channels of
// weights would not fit the registers next to two residual sets, 32 (8 channel groups, every wave all 128 positions).
// BM = positions per tile, NSLOT = slots of the activation ring = residual register sets: tile q + NSLOT - 1 is fetched while
// tile q is multiplied (NSLOT = 2: one tile ahead, the original pipeline; deeper for the short tiles of K = 256).
template <int DT, int KS1, int KS2, bool TPOOL, bool RES, int WC, int BM = 128, int NSLOT = 2>
__global__ __launch_bounds__(512, 1) void conv111_kernel(const C111Args a) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC, ES = 16 / EPC;
    static_assert(EPC == 8, "16-bit storage types only");
    static_assert(WC == 64 || (WC == 32 && !TPOOL), "wave columns of 64 or 32 channels; the pooled tile order assumes 64");
    static_assert((BM == 128 || (BM == 64 && !TPOOL)) && NSLOT >= 2 && NSLOT <= 4, "tiles of 128 or 64 positions; the pooled row order assumes 128");
    constexpr int KS = KS1 + KS2, TN = WC / 16, D = NSLOT - 1;
    constexpr int NWN = 256 / WC, NWM = 8 / NWN, WPOS = BM / NWM, MT = WPOS / 16;   // wave grid; positions, m-tiles per wave
    constexpr int SLAB = BM * 128, STAGE = KS * SLAB;  // bytes: one 64-channel K slab of the tile; one ring slot
    constexpr int PROW = WC + 4;                       // patch row stride in floats (pad: conflict-free b128 writes)
    constexpr int LPR = WC / 8, RPI = 64 / LPR, ITS = 16 / RPI;   // epilogue: lanes per row, rows per instruction, instructions per m-tile
    constexpr int NR = MT * ITS;                       // residual loads = output stores of a wave per tile
    constexpr int NPC = BM / 64;                       // DMA pieces of a wave per K slab

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int wn = wave % NWN, wm = wave / NWN;        // channel group, position part
    float* patch = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + NSLOT * STAGE) + wave * (16 * PROW);
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

    // ---- epilogue geometry: lane = (row rr of RPI, 8 channels at cc) of a 16-row patch.
    // REGEPI (round 4, the 32-channel wave columns of K = 256): no patch - the packed halves of the wave's two channel tiles trade
    // places between lane rows (v_permlane16_swap, swap_pair16), so lane (position frow, group fg) holds 8 consecutive channels:
    // row rr = frow, channels (fg & 1) * 16 + (fg >> 1) * 8 .. + 7 - the same 64-byte row segments per wave instruction as the
    // patch form wrote, without four LDS round trips per tile on an in-order wave; the residual rows arrive in that layout and are
    // traded back into tile order for the fp32 sum.  Measured: a tie with the patch form (59.1 against 58.5 us for s4's c conv): the
    // launch is bound by its 64-byte memory segments (4.6 TB/s with the compute switched off), not by the transposition.
    constexpr bool REGEPI = WC == 32;
    const int rr = REGEPI ? frow : lane / LPR, cc = REGEPI ? (fg & 1) * 16 + (fg >> 1) * 8 : (lane % LPR) * 8;
    const int ch0 = col * 256 + wn * WC + cc;
    // BN scale / shift: of this lane's 8 channels in row order, or (REGEPI) of its 4 channels of tile 0 and of tile 1: registers
    // (read from LDS per use they were a third of the kernel's LDS traffic; the two-pass accumulators left the room)
    f32x4 sc[2], sf[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int chp = REGEPI ? col * 256 + wn * WC + e * 16 + fg * 4 : ch0 + 4 * e;
        sc[e] = *reinterpret_cast<const f32x4*>(a.scale + chp);
        sf[e] = *reinterpret_cast<const f32x4*>(a.shift + chp);
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
            for (int i = 0; i < NPC; ++i)
                blds16(row_live(xrow + 64 * i, live) ? xoff : kOutOfRange, d1, s * 128 + i * half_off, base + s * SLAB + i * (64 * 128));
        if (KS2) {
            const i32x4 d2 = make_desc(a.in2 + o * a.Cin2 * ES);
#pragma unroll
            for (int s = 0; s < KS2; ++s)
#pragma unroll
                for (int i = 0; i < NPC; ++i)
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
    auto load_residual = [&](int tile, u32x4 (&r)[NR]) {
        int live;
        const long long o = tile_origin(tile, live);
        const i32x4 rd = make_desc(a.res + o * a.Cout * ES);
#pragma unroll
        for (int k = 0; k < NR; ++k)
            r[k] = bload16_nt_uncounted(row_live(row0 + k_row(k), live) ? res_lane : kOutOfRange, rd, res_soff(k));
    };

    const int my_tiles = first < a.tiles ? (a.tiles - first + stride - 1) / stride : 0;
    // tile q: activations in ring slot q % NSLOT, residual rows in register set q % NSLOT (the tile loop is unrolled NSLOT
    // times: static slots and sets, nothing is copied)
    u32x4 rs[NSLOT][NR];
#pragma unroll
    for (int u = 0; u < NSLOT; ++u)
#pragma unroll
        for (int k = 0; k < NR; ++k) rs[u][k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < my_tiles) {
            issue_tile(first + j * stride, j);
            if (RES) load_residual(first + j * stride, rs[j]);
        }
    __builtin_amdgcn_sched_barrier(0);
    // vector-memory operations of a wave per tile: loads (DMA pieces + residual rows) and stores.  Behind the loads of tile q
    // the queue holds the loads of tiles q+1 .. q+D-1 and the stores of up to D earlier tiles, all issued by every wave in full
    // (only the layer's last tile can be ragged, and its stores are nobody's predecessor): counted waits leave them in flight.
    constexpr int NL = KS * NPC + (RES ? NR : 0), NS = TPOOL ? NR / 2 : NR;
    static_assert((D - 1) * NL + D * NS <= 63, "vmcnt range");

    for (int q0 = 0; q0 < my_tiles; q0 += NSLOT) {
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
        const int q = q0 + u;
        if (q >= my_tiles) break;
        const int tile = first + q * stride, slot = u;
        u32x4 (&rcur)[NR] = rs[u];
        // tile q's activations and residual rows have landed ...
        if (D == 1) wait_vmcnt<0>();
        else {
            const int nl = my_tiles - 1 - q < D - 1 ? my_tiles - 1 - q : D - 1, ns = q < D ? q : D;
            wait_counted<NL, NS, D>(nl, ns);
        }
        __builtin_amdgcn_s_barrier();                    // ... for every wave, and nobody still reads slot (q - 1) % NSLOT
        if (RES) {
#pragma unroll
            for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(rcur[k]));   // hipcc's own wait for the set sits here, behind ours
        }
        if (q + D < my_tiles) {                          // D tiles ahead: DMA the activations, load the residual rows
            issue_tile(tile + D * stride, (u + D) % NSLOT);
            if (RES) load_residual(tile + D * stride, rs[(u + D) % NSLOT]);
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

            if constexpr (REGEPI) {
                // ---- epilogue in registers: BN in tile order, + the residual traded back into tile order, ReLU, the one rounding,
                // then the packed halves of the two tiles trade lane rows: 16 bytes = 8 consecutive channels per lane
                const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) {
                    const int j = 2 * h + jl, row = wm * WPOS + j * 16 + rr;
                    f32x4 v0 = acc[0][jl] * sc[0] + sf[0], v1 = acc[1][jl] * sc[1] + sf[1];
                    if (RES) {
                        const u32x4 x = rcur[j];
                        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(x[0], x[2], false, false);
                        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(x[1], x[3], false, false);
                        const u32x2 ra = u32x2{s0[0], s1[0]}, rb = u32x2{s0[1], s1[1]};
                        v0 += Vec4<DT>::load(&ra); v1 += Vec4<DT>::load(&rb);
                    }
                    v0[0] = max_nan(v0[0], lo); v0[1] = max_nan(v0[1], lo); v0[2] = max_nan(v0[2], lo); v0[3] = max_nan(v0[3], lo);
                    v1[0] = max_nan(v1[0], lo); v1[1] = max_nan(v1[1], lo); v1[2] = max_nan(v1[2], lo); v1[3] = max_nan(v1[3], lo);
                    u32x2 p0, p1;
                    Vec4<DT>::store(&p0, v0); Vec4<DT>::store(&p1, v1);
                    const u32x4 o = swap_pair16(p0, p1);
                    if (row_live(row, live))
                        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(obase + out_soff(j) + out_lane));
                }
                continue;
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
                        for (int e = 0; e < 8; ++e) v[e] = max_nan(keep[it][e], v[e]);   // NaN propagates like ATen's max_pool
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
}

template <int DT, int KS1, int KS2, bool TPOOL, bool RES, int WC = 64, int BM = 128, int NSLOT = 2>
static int launch111(const C111Args& a, int blocks, hipStream_t stream) {
    const int lds = NSLOT * (KS1 + KS2) * BM * 128 + 8 * 16 * (WC + 4) * 4;
    AF_SET_MAX_LDS((&conv111_kernel<DT, KS1, KS2, TPOOL, RES, WC, BM, NSLOT>), lds, "conv111");
    hipLaunchKernelGGL((conv111_kernel<DT, KS1, KS2, TPOOL, RES, WC, BM, NSLOT>), dim3(blocks), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv111_kernel");
    return AF_OK;
}

// positions per tile: 64 for K = 128 / 256 (16- / 32-KB stages, four of them in the ring), 128 for K = 64
static int conv111_bm(const af_conv_desc* d) { return d->cin >= 128 ? 64 : 128; }

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
    const int bm = conv111_bm(d);
    const long long tiles = d->tpool ? (long long)d->n * (d->t / 2) * ((hw + 63) / 64) : ((long long)d->n * d->t * hw + bm - 1) / bm;
    if (tiles >= (1LL << 30)) return 0;
    // persistent streams only pay with several tiles per workgroup (one clip of the deep stages stays on the generic path)
    if (tiles * bm / 128 * (d->cout / 256) < 4LL * device_cus()) return 0;                // (counted in 128-position tiles)
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
    // K = 256 (s4 `c`).  Round 4, late: 64-channel wave columns - 128-byte row segments for the residual rows and the output instead of
    // 64 - with a TWO-slot ring.  128 weight registers next to the three residual sets of the 4-slot ring spilled 11 registers (and a
    // scratch reload's vmcnt(0) is poison in this loop), which is why rounds 2-4 ran this layer on 32-channel columns; with ONE residual
    // set (one tile ahead) it is 242 registers, and one tile of look-ahead on full lines beats three tiles of it on half lines:
    // 58.6 -> 49.9 us (tools/exp_c111_wc64.py, interleaved on one box).  AF_C111_WC64=0: the 32-channel form, for A/B runs.
    // (A three-slot ring with the BN parameters moved to LDS - 256 registers, no spills - measured the same as two slots: 49.0 us.)
    if (d->cin == 256) {
        const char* ewc = getenv("AF_C111_WC64");
        if (!(ewc && atoi(ewc) == 0))
            return residual ? (bf ? launch111<AF_BF16, 4, 0, false, true, 64, 64, 2>(a, blocks, stream) : launch111<AF_F16, 4, 0, false, true, 64, 64, 2>(a, blocks, stream))
                            : (bf ? launch111<AF_BF16, 4, 0, false, false, 64, 64, 2>(a, blocks, stream) : launch111<AF_F16, 4, 0, false, false, 64, 64, 2>(a, blocks, stream));
        return residual ? (bf ? launch111<AF_BF16, 4, 0, false, true, 32, 64, 4>(a, blocks, stream) : launch111<AF_F16, 4, 0, false, true, 32, 64, 4>(a, blocks, stream))
                        : (bf ? launch111<AF_BF16, 4, 0, false, false, 32, 64, 4>(a, blocks, stream) : launch111<AF_F16, 4, 0, false, false, 32, 64, 4>(a, blocks, stream));
    }
    // K = 128: tiles of 64 positions as well (16-KB stages), three tiles ahead
    return residual ? (bf ? launch111<AF_BF16, 2, 0, false, true, 64, 64, 4>(a, blocks, stream) : launch111<AF_F16, 2, 0, false, true, 64, 64, 4>(a, blocks, stream))
                    : (bf ? launch111<AF_BF16, 2, 0, false, false, 64, 64, 4>(a, blocks, stream) : launch111<AF_F16, 2, 0, false, false, 64, 64, 4>(a, blocks, stream));
#undef AF_C111
}

}  // namespace af

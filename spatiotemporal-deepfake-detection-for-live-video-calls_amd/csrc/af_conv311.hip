// 3x1x1 / stride 1 / pad (1,0,0) convolution into 64 or 128 channels + BN + ReLU: the temporal `a` conv of the
// s2 / s3 bottlenecks (reference altfreezing/slowfast/models/resnet_helper.py:267-281).
//
// The generic implicit GEMM brings every activation row into LDS once PER TAP.  For a temporal kernel the three
// taps of an output position are the SAME spatial position in frames t-1, t, t+1 - rows that other outputs of the
// same workgroup need anyway if the tile is cut along time.  This kernel therefore tiles M as
//     tile = (clip n, P consecutive spatial positions, ALL T frames),   P = BM / T   (8 or 16)
// and keeps, per 64-channel K slab, ONE activation image in LDS: (T + 2) x P rows (row = (t + 1) * P + p; the two
// extra frames are the zero padding, fetched as out-of-range lanes).  Tap dt of output row r is LDS row r + dt * P:
// a constant row shift that is a multiple of 8, so the XOR swizzle (chunk ^ (row & 7)) is unchanged and every B
// fragment stays one conflict-free ds_read_b128.  L2 -> LDS traffic for activations drops 3x (the layer was bound
// by exactly that), and what remains is close to the HBM stream of the layer.
//
// A stage = one K slab: 3 x BN weight rows (tap-major) + the activation image = 58-68 KB; 2-slot ring.  The
// workgroup is persistent: the stage stream runs on across tile boundaries, so the first slab of the next tile
// is in flight while the epilogue of this one (BN + ReLU, transposed through a per-wave patch in the output type,
// 16-byte row stores) runs.  8 waves, each 64 channels x 32 rows: BN = 64 -> 256-row tiles, BN = 128 -> 128-row tiles.
#include "af_common.h"

namespace af {

struct C311Args {
    const char* in;
    const char* w;       // packed [BN][3][CinP]
    const float* scale;
    const float* shift;
    char* out;
    int T, HW, Cin, kpt;         // kpt = K slabs per tap (Cin / slab elements)
    int P, chunks;               // spatial positions per tile, tiles per clip
    int tiles;                   // N * chunks
    int relu, out_ld;
};

template <int DT, int BN>
__global__ __launch_bounds__(512, 2) void conv311_kernel(const C311Args a) {
    typedef Elem<DT> E;
    typedef typename E::type OT;
    constexpr int EPC = E::EPC, ES = 16 / EPC;
    constexpr int WN = BN / 64, BM = 256 / WN, TN = 4, TM = 2;
    constexpr int WROWS = 3 * BN;                      // weight rows per stage: (dt, channel)
    constexpr int WPIECES = WROWS / 8 / 8;             // DMA pieces per wave: 3 or 6
    constexpr int XPW = (BM + 2 * 16) / 8 / 8 + 1;     // activation pieces per wave (rows: BM + 2 P, P <= 16)
    constexpr int PROW = 64 + EPC;                     // patch row stride (elements; keeps rows 16-byte aligned)

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int P = a.P, XR = BM + 2 * P;                // activation rows per stage
    const int XP = XR >> 3;                            // activation pieces per stage
    const int stage_bytes = (WROWS + XR) * 128;
    OT* patch = reinterpret_cast<OT*>(reinterpret_cast<char*>(smem) + 2 * stage_bytes) + wave * (16 * PROW);
    const int wn = wave % WN, wm = wave / WN;          // channel half (BN = 128), 32-row group

    // ---- producer state: per-lane offsets that never change
    const int drow = lane >> 3, chunk = (lane & 7) ^ drow;       // row inside a piece; source chunk (row & 7 == drow)
    const long long Kw = 3LL * a.kpt * 8 * EPC;                  // weight row length (elements)
    const i32x4 wdesc = make_desc(a.w);
    unsigned woff[WPIECES];
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
        const int row = (wave + 8 * i) * 8 + drow, dt = row / BN, ch = row % BN;
        woff[i] = (unsigned)((ch * Kw + (long long)dt * a.kpt * 8 * EPC) * ES + chunk * 16);
    }
    // activation row (t, p) of the tile: offset from the tile origin (clip n, frame 0, position hw0)
    unsigned xoff[XPW];
    int xp[XPW];                                                 // p of the row (ragged last chunk: masked per tile)
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int row = (wave + 8 * i) * 8 + drow;
        const int t = row / P - 1, p = row % P;
        xp[i] = p;
        xoff[i] = (t >= 0 && t < a.T && row < XR) ? (unsigned)((((long long)t * a.HW + p) * a.Cin) * ES + chunk * 16) : kOutOfRange;
    }
    const long long clip_bytes = (long long)a.T * a.HW * a.Cin * ES;

    // stage stream: q-th stage of this workgroup = (its (q / kpt)-th tile, slab q % kpt)
    const int my_tiles = (a.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * a.kpt;
    int p_tile = blockIdx.x, p_kc = 0;                           // producer cursor
    auto issue_stage = [&](int slot) {
        const int n = p_tile / a.chunks, hw0 = (p_tile % a.chunks) * P;
        const i32x4 xdesc = make_desc(a.in + n * clip_bytes + (long long)hw0 * a.Cin * ES);
        const unsigned base = lds0 + slot * stage_bytes + wave * (8 * 128);
        const int soff = p_kc * 128;
#pragma unroll
        for (int i = 0; i < WPIECES; ++i) blds16(woff[i], wdesc, soff, base + i * (64 * 128));
#pragma unroll
        for (int i = 0; i < XPW; ++i)
            if (wave + 8 * i < XP)
                blds16(hw0 + xp[i] < a.HW ? xoff[i] : kOutOfRange, xdesc, soff, base + WROWS * 128 + i * (64 * 128));
        if (++p_kc == a.kpt) { p_kc = 0; p_tile += gridDim.x; }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 sc[TN], sf[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + wn * 64 + i * 16 + fg * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + wn * 64 + i * 16 + fg * 4);
    }

    if (total > 0) issue_stage(0);
    int c_tile = blockIdx.x, c_kc = 0;                           // consumer cursor
    for (int q = 0; q < total; ++q) {
        const int slot = q & 1;
        wait_vmcnt<0>();                                         // stage q (the only one in flight) has landed ...
        __builtin_amdgcn_s_barrier();                            // ... for everyone, and slot (q+1)&1 is no longer read
        if (q + 1 < total) issue_stage(slot ^ 1);
        const uint4* ws = smem + slot * (stage_bytes / 16) + (wn * 64 + frow) * 8;
        const uint4* xs = smem + slot * (stage_bytes / 16) + WROWS * 8 + (wm * 32 + frow) * 8;
        // the six (tap, k-half) steps of the stage, software-pipelined by hand: the fragments of step s + 1 are read while step
        // s multiplies (hipcc had left the reads right in front of their MFMAs: a chain of LDS latencies)
        uint4 af[2][TN], bf[2][TM];
        auto read_step = [&](uint4 (&fa)[TN], uint4 (&fb)[TM], int step) {
            const int dt = step >> 1, c = ((step & 1) * 4 + fg) ^ (frow & 7);
#pragma unroll
            for (int i = 0; i < TN; ++i) fa[i] = ws[(dt * BN + i * 16) * 8 + c];
#pragma unroll
            for (int j = 0; j < TM; ++j) fb[j] = xs[(dt * P + j * 16) * 8 + c];
        };
        read_step(af[0], bf[0], 0);
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            if (step + 1 < 6) read_step(af[(step + 1) & 1], bf[(step + 1) & 1], step + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<DT>::run(af[step & 1][i], bf[step & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++c_kc < a.kpt) continue;

        // ---- tile finished: BN + ReLU + the one rounding, transpose through the wave's patch, 16-byte row stores
        const int n = c_tile / a.chunks, hw0 = (c_tile % a.chunks) * P;
        constexpr int LPR = 64 / EPC, RPI = 64 / LPR;            // lanes per output row; rows per wave-instruction
        const int rr = lane / LPR, cc = (lane % LPR) * EPC;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                f32x4 v = acc[i][j] * sc[i] + sf[i];
                if (a.relu) { v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]); }
                Vec4<DT>::store(reinterpret_cast<char*>(patch + frow * PROW + i * 16 + fg * 4), v);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int row = it * RPI + rr;
                const int r = wm * 32 + j * 16 + row;            // tile row = t * P + p
                const int t = r / P, p = r % P;
                const u32x4 o = *reinterpret_cast<const u32x4*>(patch + row * PROW + cc);
                if (hw0 + p < a.HW) {
                    const long long pos = ((long long)n * a.T + t) * a.HW + hw0 + p;
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.out + (pos * a.out_ld + wn * 64 + cc) * ES));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        c_kc = 0; c_tile += gridDim.x;
    }
}

template <int DT, int BN>
static int launch311(const C311Args& a, int blocks, hipStream_t stream) {
    constexpr int BM = 256 / (BN / 64), EPC = Elem<DT>::EPC;
    const int lds = 2 * (3 * BN + BM + 2 * a.P) * 128 + 8 * 16 * (64 + EPC) * (16 / EPC);
    AF_SET_MAX_LDS((&conv311_kernel<DT, BN>), 160 * 1024, "conv311");
    hipLaunchKernelGGL((conv311_kernel<DT, BN>), dim3(blocks), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv311_kernel");
    return AF_OK;
}

// tile height for a layer, 0 if the layer does not take the time-tiled path
static int conv311_tile_rows(const af_conv_desc* d) {
    if (d->tpool) return 0;
    if (d->kt != 3 || d->kh != 1 || d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1) return 0;
    if (d->pt != 1 || d->ph != 0 || d->pw != 0) return 0;
    const int bke = d->dtype == AF_F32 ? 32 : 64;
    if (d->cin % bke != 0) return 0;
    int bm = 0;
    if (d->cout == 64 && (d->t == 16 || d->t == 32)) bm = 256;                        // P = 16 or 8
    else if (d->cout == 128 && d->t == 16 && d->dtype != AF_F32) bm = 128;            // P = 8 (fp32: ring + patch exceed LDS)
    if (!bm) return 0;
    if ((long long)d->t * d->h * d->w * d->cin * dtype_size(d->dtype) >= (1LL << 31)) return 0;
    const int p = bm / d->t;
    if ((long long)d->n * ((d->h * d->w + p - 1) / p) >= (1LL << 31)) return 0;
    return bm;
}

// true iff this layer takes the time-tiled path (also used by af_conv_variant)
bool conv311_applies(const af_conv_desc* d, const void* residual, int out_ld) {
    return !residual && conv311_tile_rows(d) != 0;
}

int conv311_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                void* out, int out_ld, hipStream_t stream) {
    C311Args a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.T = d->t; a.HW = d->h * d->w; a.Cin = d->cin; a.kpt = d->cin / (d->dtype == AF_F32 ? 32 : 64);
    a.P = conv311_tile_rows(d) / d->t; a.chunks = (a.HW + a.P - 1) / a.P; a.tiles = d->n * a.chunks;
    a.relu = d->relu; a.out_ld = out_ld;
    const int cus = device_cus();                        // persistent grid: one workgroup per CU
    const int blocks = a.tiles < cus ? a.tiles : cus;
    if (d->cout == 128) return d->dtype == AF_BF16 ? launch311<AF_BF16, 128>(a, blocks, stream) : launch311<AF_F16, 128>(a, blocks, stream);
    switch (d->dtype) {
        case AF_F32: return launch311<AF_F32, 64>(a, blocks, stream);
        case AF_BF16: return launch311<AF_BF16, 64>(a, blocks, stream);
        default: return launch311<AF_F16, 64>(a, blocks, stream);
    }
}

}  // namespace af

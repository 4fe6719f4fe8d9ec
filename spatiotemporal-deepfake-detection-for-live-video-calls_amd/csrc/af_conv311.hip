// 3x1x1 / stride 1 / pad (1,0,0) convolution into 64 channels + BN + ReLU: the temporal `a` conv of the s2
// bottlenecks (reference altfreezing/slowfast/models/resnet_helper.py:267-281; 64->64 in res0, 256->64 after it).
//
// The generic implicit GEMM brings every activation row into LDS once PER TAP.  For a temporal kernel the three
// taps of an output position are the SAME spatial position in frames t-1, t, t+1 - rows that other outputs of the
// same workgroup need anyway if the tile is cut along time.  This kernel therefore tiles M as
//     tile = (clip n, P consecutive spatial positions, ALL T frames),   P = 256 / T   (8 for T = 32, 16 for T = 16)
// and keeps, per 64-channel K slab, ONE activation image in LDS: (T + 2) x P rows (row = (t + 1) * P + p; the two
// extra frames are the zero padding, fetched as out-of-range lanes).  Tap dt of output row r is LDS row r + dt * P:
// a constant row shift that is a multiple of 8, so the XOR swizzle (chunk ^ (row & 7)) is unchanged and every B
// fragment stays one conflict-free ds_read_b128.  L2 -> LDS traffic for activations drops 3x (the layer was bound
// by exactly that), and what remains is close to the HBM stream of the layer.
//
// A stage = one K slab: 3 x 64 weight rows (tap-major) + the activation image = 58-60 KB; 2-slot ring.  The
// workgroup is persistent: the stage stream runs on across tile boundaries, so the first slab of the next tile
// is in flight while the epilogue of this one (BN + ReLU through a per-wave fp32 patch, 16-byte row stores) runs.
// 8 waves, each 64 channels x 32 rows of the tile.
#include "af_common.h"

namespace af {

struct C311Args {
    const char* in;
    const char* w;       // packed [64][3][CinP]
    const float* scale;
    const float* shift;
    char* out;
    int T, HW, Cin, kpt;         // kpt = K slabs per tap (Cin / slab elements)
    int P, chunks;               // spatial positions per tile, tiles per clip
    int tiles;                   // N * chunks
    int relu, out_ld;
};

template <int DT>
__global__ __launch_bounds__(512, 2) void conv311_c64_kernel(const C311Args a) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC, ES = 16 / EPC;
    constexpr int BN = 64, BM = 256, TN = 4, TM = 2;
    constexpr int WROWS = 3 * BN;                      // weight rows per stage: (dt, channel)
    constexpr int WPIECES = WROWS / 8 / 8;             // = 3 DMA pieces per wave
    constexpr int XPW = 5;                             // activation pieces per wave (covers up to 320 rows)
    constexpr int PROW = BN + 4;                       // patch row stride (floats)

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int P = a.P, XR = BM + 2 * P;                // activation rows per stage
    const int XP = XR >> 3;                            // activation pieces per stage
    const int stage_bytes = (WROWS + XR) * 128;
    float* patch = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + 2 * stage_bytes) + wave * (16 * PROW);

    // ---- producer state: per-lane offsets that never change
    const int drow = lane >> 3, chunk = (lane & 7) ^ drow;       // row inside a piece; source chunk (row & 7 == drow)
    const long long Kw = 3LL * a.kpt * 8 * EPC;                  // weight row length (elements)
    const i32x4 wdesc = make_desc(a.w);
    unsigned woff[WPIECES];
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
        const int row = (wave + 8 * i) * 8 + drow, dt = row >> 6, ch = row & 63;
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
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + i * 16 + fg * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + i * 16 + fg * 4);
    }

    if (total > 0) issue_stage(0);
    int c_tile = blockIdx.x, c_kc = 0;                           // consumer cursor
    for (int q = 0; q < total; ++q) {
        const int slot = q & 1;
        wait_vmcnt<0>();                                         // stage q (the only one in flight) has landed ...
        __builtin_amdgcn_s_barrier();                            // ... for everyone, and slot (q+1)&1 is no longer read
        if (q + 1 < total) issue_stage(slot ^ 1);
        const uint4* ws = smem + slot * (stage_bytes / 16) + frow * 8;
        const uint4* xs = smem + slot * (stage_bytes / 16) + WROWS * 8 + (wave * 32 + frow) * 8;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = (kk * 4 + fg) ^ (frow & 7);
                uint4 af[TN], bf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) af[i] = ws[(dt * 64 + i * 16) * 8 + c];
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[j] = xs[(dt * P + j * 16) * 8 + c];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<DT>::run(af[i], bf[j], acc[i][j]);
            }
        if (++c_kc < a.kpt) continue;

        // ---- tile finished: BN + ReLU, transpose through the wave's patch, 16-byte row stores
        const int n = c_tile / a.chunks, hw0 = (c_tile % a.chunks) * P;
        constexpr int LPR = BN / EPC, RPI = 64 / LPR;            // lanes per output row; rows per wave-instruction
        const int rr = lane / LPR, cc = (lane % LPR) * EPC;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                *reinterpret_cast<f32x4*>(patch + frow * PROW + i * 16 + fg * 4) = acc[i][j] * sc[i] + sf[i];
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int row = it * RPI + rr;
                const int r = wave * 32 + j * 16 + row;          // tile row = t * P + p
                const int t = r / P, p = r % P;
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(patch + row * PROW + cc + e);
                    v[e] = x[0]; v[e + 1] = x[1]; v[e + 2] = x[2]; v[e + 3] = x[3];
                }
                if (hw0 + p < a.HW) {
                    uint4 o;
                    typename E::type* oe = reinterpret_cast<typename E::type*>(&o);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) oe[e] = E::from_f32(a.relu ? fmaxf(v[e], 0.f) : v[e]);
                    const long long pos = ((long long)n * a.T + t) * a.HW + hw0 + p;
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + (pos * a.out_ld + cc) * ES));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        c_kc = 0; c_tile += gridDim.x;
    }
}

template <int DT>
static int launch311(const C311Args& a, int blocks, hipStream_t stream) {
    const int lds = 2 * (3 * 64 + 256 + 2 * a.P) * 128 + 8 * 16 * (64 + 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv311_c64_kernel<DT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return set_error(AF_ERR_LAUNCH, "conv311: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv311_c64_kernel<DT>), dim3(blocks), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv311_c64_kernel");
    return AF_OK;
}

// true iff this layer takes the time-tiled path (also used by af_conv_variant)
bool conv311_applies(const af_conv_desc* d, const void* residual, int out_ld) {
    if (residual || d->tpool) return false;
    if (d->kt != 3 || d->kh != 1 || d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1) return false;
    if (d->pt != 1 || d->ph != 0 || d->pw != 0 || d->cout != 64) return false;
    const int bke = d->dtype == AF_F32 ? 32 : 64;
    if (d->cin % bke != 0) return false;
    if (d->t != 16 && d->t != 32) return false;                      // P = 256 / T in {16, 8}: the 2-slot ring fits LDS
    if ((long long)d->t * d->h * d->w * d->cin * dtype_size(d->dtype) >= (1LL << 31)) return false;
    if ((long long)d->n * ((d->h * d->w + 256 / d->t - 1) / (256 / d->t)) >= (1LL << 31)) return false;
    return true;
}

int conv311_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                void* out, int out_ld, hipStream_t stream) {
    C311Args a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.T = d->t; a.HW = d->h * d->w; a.Cin = d->cin; a.kpt = d->cin / (d->dtype == AF_F32 ? 32 : 64);
    a.P = 256 / d->t; a.chunks = (a.HW + a.P - 1) / a.P; a.tiles = d->n * a.chunks;
    a.relu = d->relu; a.out_ld = out_ld;
    static int cus = 0;                                  // persistent grid: one workgroup per CU
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    const int blocks = a.tiles < cus ? a.tiles : cus;
    switch (d->dtype) {
        case AF_F32: return launch311<AF_F32>(a, blocks, stream);
        case AF_BF16: return launch311<AF_BF16>(a, blocks, stream);
        default: return launch311<AF_F16>(a, blocks, stream);
    }
}

}  // namespace af

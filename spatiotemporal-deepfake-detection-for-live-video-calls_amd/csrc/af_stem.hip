// Stem: Conv3d(3->64, [kt,7,7], stride [1,2,2], pad [kt/2,3,3], bias=False) + BN + ReLU
// (reference altfreezing/slowfast/models/stem_helper.py:156-177; i3d: kt = 5).
//
// Cin = 3 is hostile to a K-contiguous implicit GEMM, so the prologue kernels (af_pack.hip) write the
// clip as [N][T+4][H+6][W+8][4]: channel padded 3->4 and a ZERO HALO (2 | 3 | 3 left) around T/H/W.
// With that layout
//   * one (dt,dh) kernel row = 8 pixels x 4 channels = 32 K-values that are CONTIGUOUS in HBM
//     (kw padded 7->8, the 8th tap and the 4th channel have zero weights),
//   * the left halo (3) cancels the conv's -3 offset: the run for output column wo starts at padded
//     pixel 2*wo, i.e. at a 16-byte-aligned address in every dtype,
//   * no bounds checks at all: every tap of every output position reads real memory.
// K = kt*7*32 (1120 for i3d, 66 % useful MACs) - the price of Cin=3 on an MFMA whose K is 32.
//
// A workgroup computes 256 output positions x all 64 channels.  Weights (A operand) for one dt slice
// sit in LDS as [dh][chunk][64 ch][16 B] (conflict-free ds_read_b128); activation
// fragments (B operand) are fetched straight from global memory / L2 as one aligned 16-byte load per
// lane per MFMA-K-step - each input byte is reused by ~12 positions x kt through L1/L2.
#include "af_common.h"

namespace af {

struct StemArgs {
    const char* in;      // padded input
    const char* w;       // packed [kt][kh][NCH][cout][16 B]
    const float* scale;
    const float* shift;
    char* out;
    int Tp, Hp, Wp;      // padded dims
    int kt, kh;
    int To, Ho, Wo;
    int cout;            // real output channels (<= 16 * TN; the weight image is padded to 16 * TN rows)
    long long M;
};

// NCH = 16-byte chunks per (dt,dh) K-row = 32 elements / EPC : fp32 8, 16-bit 4
// TN = channel tiles of 16: 4 for the 64-channel stems, 1 for SlowFast's 8-channel Fast-pathway stem
template <int DT, int TN>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemArgs a) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC;
    constexpr int ES = 16 / EPC;
    constexpr int NCH = 32 / EPC;
    constexpr int KK = NCH / 4;             // fragment reads per K-row (fp32 2, 16-bit 1)
    constexpr int COUT = 16 * TN, TM = 4;
    constexpr int PIXB = 4 * ES;            // bytes per padded pixel

    extern __shared__ uint4 wlds[];         // [kh][NCH][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const long long m0 = (long long)blockIdx.x * 256 + wave * 64;

    // per-lane activation base (bytes) for each of the wave's TM position tiles
    long long xoff[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        long long m = m0 + j * 16 + frow;
        if (m >= a.M) m = a.M - 1;                       // clamp: compute garbage-free, never stored
        int wo = (int)(m % a.Wo); long long t1 = m / a.Wo;
        int ho = (int)(t1 % a.Ho); long long t2 = t1 / a.Ho;
        int to = (int)(t2 % a.To); long long n = t2 / a.To;
        // padded coords of tap (dt=0,dh=0,dw=0): t = to, h = 2*ho, w = 2*wo
        xoff[j] = ((((n * a.Tp + to) * a.Hp + 2 * ho) * a.Wp) + 2 * wo) * PIXB + fg * 16;
    }
    const long long row_bytes = (long long)a.Wp * PIXB;
    const long long plane_bytes = row_bytes * a.Hp;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int slice_chunks = a.kh * NCH * COUT;          // uint4 per dt slice
    for (int dt = 0; dt < a.kt; ++dt) {
        __syncthreads();                                 // previous slice fully consumed
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.w) + (long long)dt * slice_chunks;
        for (int i = tid; i < slice_chunks; i += 256) wlds[i] = wsrc[i];
        __syncthreads();
        for (int dh = 0; dh < a.kh; ++dh) {
            const long long tap_off = dt * plane_bytes + dh * row_bytes;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                uint4 af[TN], bf[TM];
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    bf[j] = *reinterpret_cast<const uint4*>(a.in + xoff[j] + tap_off + kk * 64);
                const uint4* wl = wlds + ((dh * NCH) + kk * 4 + fg) * COUT + frow;
#pragma unroll
                for (int i = 0; i < TN; ++i) af[i] = wl[i * 16];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<DT>::run(af[i], bf[j], acc[i][j]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int ch = i * 16 + fg * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        const f32x4 sf = *reinterpret_cast<const f32x4*>(a.shift + ch);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const long long m = m0 + j * 16 + frow;
            if (m < a.M && ch < a.cout) {
                f32x4 v = acc[i][j] * sc + sf;
                v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]);
                Vec4<DT>::store(a.out + (m * a.cout + ch) * ES, v);
            }
        }
    }
}

// Narrow stems (<= 16 output channels: SlowFast's 8-channel Fast pathway), 16-bit.  With one channel tile every MFMA
// needs its own 16-byte activation fragment from L1 and the kernel above is bound by exactly that (one fragment load per
// MFMA; L1 delivers 64 B/clk per CU).  Output rows ho and ho+1 share 5 of their 7 input rows, so here a wave owns 16
// columns x R CONSECUTIVE OUTPUT ROWS: per dt plane it loads the 2R+5 input rows once and feeds 7R MFMAs from them
// (R = 8: 21 loads for 56 MFMAs instead of 56), the 7 weight fragments of the plane coming from an LDS image of all
// kt slices that is loaded once per workgroup.
// KT > 0 (compile-time temporal taps): the 21 input rows of plane dt + 1 are fetched while plane dt multiplies (two register
// sets; left in one loop hipcc issued every plane's loads right in front of its MFMAs: the full global latency per plane).
template <int DT, int R, int KT>
__global__ __launch_bounds__(256) void stem_rows_kernel(const StemArgs a, int tiles_w, int hblocks, long long units) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC, ES = 16 / EPC, NCH = 4, PIXB = 4 * ES, NR = 2 * R + 5;
    static_assert(EPC == 8, "16-bit types only (one 16-byte chunk per lane and K-row)");
    extern __shared__ uint4 wlds[];                      // [kt][kh][NCH][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int wchunks = a.kt * a.kh * NCH * 16;
    for (int i = tid; i < wchunks; i += 256) wlds[i] = reinterpret_cast<const uint4*>(a.w)[i];
    __syncthreads();
    const long long unit = (long long)blockIdx.x * 4 + wave;
    if (unit >= units) return;
    const int wt = (int)(unit % tiles_w); long long q = unit / tiles_w;
    const int hb = (int)(q % hblocks); q /= hblocks;
    const int to = (int)(q % a.To); const long long n = q / a.To;
    const int ho0 = hb * R;
    int wo = wt * 16 + frow;
    const bool col_ok = wo < a.Wo;
    if (!col_ok) wo = a.Wo - 1;                          // clamp: loads stay inside the row, result never stored
    const long long row_bytes = (long long)a.Wp * PIXB, plane_bytes = row_bytes * a.Hp;
    // wave-uniform row base (scalar registers) + one per-lane 32-bit offset: no 64-bit address VALU per load
    const char* ubase = a.in + (((n * a.Tp + to) * a.Hp + 2 * ho0) * a.Wp) * PIXB;
    const unsigned loff = (unsigned)(2 * wo * PIXB + fg * 16);
    const int rmax = a.Hp - 1 - 2 * ho0;                 // last input row that exists below the block's first one

    f32x4 acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_plane = [&](uint4 (&bx)[NR], int dt) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            bx[r] = *reinterpret_cast<const uint4*>(ubase + dt * plane_bytes + (long long)(r < rmax ? r : rmax) * row_bytes + loff);
    };
    auto mul_plane = [&](const uint4 (&bx)[NR], int dt) {
        uint4 af[7];
#pragma unroll
        for (int dh = 0; dh < 7; ++dh) af[dh] = wlds[((dt * 7 + dh) * NCH + fg) * 16 + frow];
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int dh = 0; dh < 7; ++dh) Mma<DT>::run(af[dh], bx[2 * j + dh], acc[j]);
    };
    if (KT > 0) {
        uint4 b0[NR], b1[NR];
        load_plane(b0, 0);
#pragma unroll
        for (int dt = 0; dt < KT; ++dt) {
            if (dt + 1 < KT) load_plane((dt & 1) ? b0 : b1, dt + 1);
            __builtin_amdgcn_sched_barrier(0);
            mul_plane((dt & 1) ? b1 : b0, dt);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int dt = 0; dt < a.kt; ++dt) {
            uint4 bx[NR];
            load_plane(bx, dt);
            mul_plane(bx, dt);
        }
    }
    const int ch = fg * 4;
    if (!col_ok || ch >= a.cout) return;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch), sf = *reinterpret_cast<const f32x4*>(a.shift + ch);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (ho0 + j >= a.Ho) break;
        const long long m = ((n * a.To + to) * a.Ho + ho0 + j) * a.Wo + wo;
        f32x4 v = acc[j] * sc + sf;
        v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]);
        Vec4<DT>::store(a.out + (m * a.cout + ch) * ES, v);
    }
}

template <int DT>
static int launch_stem_rows(const StemArgs& a, hipStream_t stream) {
    constexpr int R = 8;
    const int tiles_w = (a.Wo + 15) / 16, hblocks = (a.Ho + R - 1) / R;
    const long long n_to = a.M / ((long long)a.Ho * a.Wo), units = n_to * hblocks * tiles_w, blocks = (units + 3) / 4;
    if (blocks > 0x7fffffffLL) return set_error(AF_ERR_ARG, "stem: grid too large");
    const int lds = a.kt * a.kh * 4 * 16 * 16;
    if (a.kt == 5) hipLaunchKernelGGL((stem_rows_kernel<DT, R, 5>), dim3((unsigned)blocks), dim3(256), lds, stream, a, tiles_w, hblocks, units);
    else hipLaunchKernelGGL((stem_rows_kernel<DT, R, 0>), dim3((unsigned)blocks), dim3(256), lds, stream, a, tiles_w, hblocks, units);
    AF_CHECK_LAUNCH("stem_rows_kernel");
    return AF_OK;
}

template <int DT, int TN>
static int launch_stem_tn(const StemArgs& a, hipStream_t stream) {
    constexpr int NCH = 32 / Elem<DT>::EPC;
    const int lds = a.kh * NCH * (16 * TN) * 16;
    const long long blocks = (a.M + 255) / 256;
    if (blocks > 0x7fffffffLL) return set_error(AF_ERR_ARG, "stem: grid too large");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_kernel<DT, TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(AF_ERR_LAUNCH, "stem: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((stem_kernel<DT, TN>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    AF_CHECK_LAUNCH("stem_kernel");
    return AF_OK;
}

template <int DT>
static int launch_stem(const StemArgs& a, hipStream_t stream) {
    if (a.cout <= 16 && a.kh == 7) {
        if (DT == AF_BF16) return launch_stem_rows<AF_BF16>(a, stream);
        if (DT == AF_F16) return launch_stem_rows<AF_F16>(a, stream);
    }
    return a.cout <= 16 ? launch_stem_tn<DT, 1>(a, stream) : launch_stem_tn<DT, 4>(a, stream);
}

}  // namespace af

extern "C" int af_stem_conv_bn_relu(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                    const float* scale, const float* shift, void* out, void* stream) {
    using namespace af;
    AF_REQUIRE(d && stem_in && w_packed && scale && shift && out, "stem: null argument");
    AF_REQUIRE(dtype_ok(d->dtype), "stem: bad dtype %d", d->dtype);
    AF_REQUIRE(d->cin == 3 && (d->cout == 64 || (d->cout > 0 && d->cout <= 16 && d->cout % 4 == 0)),
               "stem: expects 3 -> 64 (or <= 16) channels (got %d -> %d)", d->cin, d->cout);
    AF_REQUIRE(d->kh == 7 && d->kw == 7 && d->sh == 2 && d->sw == 2 && d->st == 1 && d->ph == 3 && d->pw == 3,
               "stem: expects a [kt,7,7] kernel, stride [1,2,2], pad [kt/2,3,3]");
    AF_REQUIRE(d->kt >= 1 && d->kt <= 2 * AF_STEM_PAD_T + 1 && (d->kt & 1) && d->pt == d->kt / 2, "stem: bad kt/pt");
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0, "stem: bad dims");
    const int to = d->t, ho = (d->h + 6 - 7) / 2 + 1, wo = (d->w + 6 - 7) / 2 + 1;
    AF_REQUIRE(to == d->to && ho == d->ho && wo == d->wo, "stem: output dims mismatch");
    AF_REQUIRE(aligned16(stem_in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out),
               "stem: buffers must be 16-byte aligned");
    StemArgs a;
    a.in = (const char*)stem_in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.Tp = d->t + 2 * AF_STEM_PAD_T; a.Hp = d->h + 2 * AF_STEM_PAD_H; a.Wp = d->w + AF_STEM_PAD_W_TOTAL;
    a.kt = d->kt; a.kh = d->kh; a.To = to; a.Ho = ho; a.Wo = wo; a.cout = d->cout;
    a.M = (long long)d->n * to * ho * wo;
    // temporal halo is AF_STEM_PAD_T; a kernel with kt < 5 starts (PAD_T - pt) planes into it
    a.in += (long long)(AF_STEM_PAD_T - d->pt) * a.Hp * a.Wp * 4 * dtype_size(d->dtype);
    hipStream_t s = (hipStream_t)stream;
    switch (d->dtype) {
        case AF_F32: return launch_stem<AF_F32>(a, s);
        case AF_BF16: return launch_stem<AF_BF16>(a, s);
        default: return launch_stem<AF_F16>(a, s);
    }
}

// Implicit-GEMM Conv3d + BatchNorm(scale/shift) [+ residual] [+ ReLU] for NDHWC activations.
//
// Replaces, per launch, nn.Conv3d(bias=False) -> nn.BatchNorm3d(eval) [-> add] [-> nn.ReLU] of the
// reference bottleneck / res block (altfreezing/slowfast/models/resnet_helper.py:267-325, 411-444).
//
// GEMM view:  D[cout][pos] = sum_k  Wp[cout][k] * X[pos][k],   k = (tap, cin),  pos = (n,to,ho,wo)
//   * weights are the MFMA "A" operand (rows = output channels) and im2col'ed activations the "B"
//     operand (cols = output positions): the 16x16 accumulator then holds 4 CONSECUTIVE CHANNELS
//     of one position per lane, so the epilogue reads scale/shift/residual and writes NDHWC with
//     16-byte (fp32) / 8-byte (16-bit) vector accesses and no transpose.
//   * both operands are K-contiguous in memory ([cout][tap][cin] weights, NDHWC activations), so a
//     K-step of one tile row is one contiguous 128-byte run = 8 x 16-byte chunks.
//   * LDS tiles are [row][8 chunks] with chunk ^= (row & 7): conflict-free ds_write_b128 from the
//     staging pass and conflict-free ds_read_b128 for the MFMA fragments (bank math in DESIGN.md).
//   * zero padding is done by predicating the global loads (no halo in HBM).
//   * pipeline: register-staged double buffering, one barrier per K-step: global loads of step
//     s+1 are issued before the MFMAs of step s and written to the other LDS buffer after them.
#include "af_common.h"

namespace af {

struct ConvArgs {
    const char* in;
    const char* w;
    const float* scale;
    const float* shift;
    const char* res;
    char* out;
    int T, H, W, Cin, Cout;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int To, Ho, Wo;
    int relu, out_ld;
    long long M;        // N*To*Ho*Wo
    int tiles_n;        // Cout / BN
    int kpt;            // K-steps per tap = Cin / BK
};

template <int DT, int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC;            // elements per 16-byte chunk
    constexpr int ES = 16 / EPC;           // bytes per element
    constexpr int BK = 8 * EPC;            // K elements per step (128 bytes per tile row)
    constexpr int WTN = BN / WN, WTM = BM / WM;
    constexpr int TN = WTN / 16, TM = WTM / 16;
    constexpr int RW = BN / 32, RX = BM / 32;   // tile rows staged per thread
    static_assert(WN * WM == 4, "4 waves per workgroup");
    static_assert(TN >= 1 && TM >= 1, "tile too small");

    extern __shared__ uint4 smem[];
    // buffer b: W tile at smem + b*STAGE, X tile at smem + b*STAGE + BN*8
    constexpr int STAGE = (BN + BM) * 8;

    // ---- workgroup -> tile, XCD-contiguous (bijective remap; placement is a speed matter only)
    const int bid = blockIdx.x, nb = gridDim.x;
    const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = swz % a.tiles_n;
    const int tile_m = swz / a.tiles_n;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int lchunk = tid & 7, lrow = tid >> 3;

    // ---- per-thread staging rows
    const long long m0 = (long long)tile_m * BM;
    long long xbase[RX];
    int ti0[RX], hi0[RX], wi0[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        long long m = m0 + lrow + 32 * i;
        if (m < a.M) {
            int wo = (int)(m % a.Wo); long long t1 = m / a.Wo;
            int ho = (int)(t1 % a.Ho); long long t2 = t1 / a.Ho;
            int to = (int)(t2 % a.To); long long n = t2 / a.To;
            ti0[i] = to * a.st - a.pt; hi0[i] = ho * a.sh - a.ph; wi0[i] = wo * a.sw - a.pw;
            xbase[i] = (((n * a.T + ti0[i]) * a.H + hi0[i]) * a.W + wi0[i]) * a.Cin + lchunk * EPC;
        } else {
            ti0[i] = -(1 << 20); hi0[i] = 0; wi0[i] = 0; xbase[i] = 0;
        }
    }
    const int taps = a.kt * a.kh * a.kw;
    const int Kw = taps * a.Cin;               // weight row length (elements)
    int wbase[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) wbase[i] = (tile_n * BN + lrow + 32 * i) * Kw + lchunk * EPC;

    uint4 xr[RX], wr[RW];
    int dt = 0, dh = 0, dw = 0, kc = 0, tap = 0;   // position of the NEXT step to load

    auto load_step = [&]() {
        const long long tapoff = ((long long)(dt * a.H + dh) * a.W + dw) * a.Cin + kc * BK;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            bool ok = (unsigned)(ti0[i] + dt) < (unsigned)a.T && (unsigned)(hi0[i] + dh) < (unsigned)a.H &&
                      (unsigned)(wi0[i] + dw) < (unsigned)a.W;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(a.in + (xbase[i] + tapoff) * ES);
            xr[i] = v;
        }
        const int woff = tap * a.Cin + kc * BK;
#pragma unroll
        for (int i = 0; i < RW; ++i) wr[i] = *reinterpret_cast<const uint4*>(a.w + (long long)(wbase[i] + woff) * ES);
        // advance (kc fastest, then dw, dh, dt)
        if (++kc == a.kpt) {
            kc = 0; ++tap;
            if (++dw == a.kw) { dw = 0; if (++dh == a.kh) { dh = 0; ++dt; } }
        }
    };
    auto store_lds = [&](int buf) {
        uint4* ws = smem + buf * STAGE;
        uint4* xs = ws + BN * 8;
        const int sw_ = lchunk ^ (lrow & 7);
#pragma unroll
        for (int i = 0; i < RW; ++i) ws[(lrow + 32 * i) * 8 + sw_] = wr[i];
#pragma unroll
        for (int i = 0; i < RX; ++i) xs[(lrow + 32 * i) * 8 + sw_] = xr[i];
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int S = taps * a.kpt;
    load_step();
    store_lds(0);
    __syncthreads();

    const int frow = lane & 15, fg = lane >> 4;
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) load_step();
        const uint4* ws = smem + buf * STAGE + (wn * WTN + frow) * 8;
        const uint4* xs = smem + buf * STAGE + BN * 8 + (wm * WTM + frow) * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = (kk * 4 + fg) ^ (frow & 7);
            uint4 af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) af[i] = ws[i * 16 * 8 + c];
#pragma unroll
            for (int j = 0; j < TM; ++j) bf[j] = xs[j * 16 * 8 + c];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<DT>::run(af[i], bf[j], acc[i][j]);
        }
        if (s + 1 < S) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: BN scale/shift (+ residual) (+ ReLU), 4 consecutive channels per lane
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int ch = tile_n * BN + wn * WTN + i * 16 + fg * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        const f32x4 sf = *reinterpret_cast<const f32x4*>(a.shift + ch);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const long long m = m0 + wm * WTM + j * 16 + frow;
            if (m < a.M) {
                f32x4 v = acc[i][j] * sc + sf;
                if (a.res) v += Vec4<DT>::load(a.res + (m * a.Cout + ch) * ES);
                if (a.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                Vec4<DT>::store(a.out + (m * a.out_ld + ch) * ES, v);
            }
        }
    }
}

template <int DT, int BN, int BM, int WN, int WM>
static int launch(const ConvArgs& a, hipStream_t stream) {
    const long long tiles_m = (a.M + BM - 1) / BM;
    const long long blocks = tiles_m * a.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return set_error(AF_ERR_ARG, "conv: grid of %lld workgroups", blocks);
    constexpr int lds = 2 * (BN + BM) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<DT, BN, BM, WN, WM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return set_error(AF_ERR_LAUNCH, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<DT, BN, BM, WN, WM>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    AF_CHECK_LAUNCH("conv_igemm_kernel");
    return AF_OK;
}

// tile variant chosen for a layer (also reported to the caller: af_conv_variant)
enum { VAR_128x128 = 0, VAR_64x128 = 1 };
static const char* const kVariantNames[] = {"conv_igemm<BN=128,BM=128>", "conv_igemm<BN=64,BM=128>"};

static int pick_variant(int cout) { return cout % 128 == 0 ? VAR_128x128 : VAR_64x128; }

template <int DT>
static int dispatch(ConvArgs& a, hipStream_t stream) {
    constexpr int BK = 8 * Elem<DT>::EPC;
    a.kpt = a.Cin / BK;
    if (pick_variant(a.Cout) == VAR_128x128) {
        a.tiles_n = a.Cout / 128;
        return launch<DT, 128, 128, 2, 2>(a, stream);
    }
    a.tiles_n = a.Cout / 64;
    return launch<DT, 64, 128, 2, 2>(a, stream);
}

}  // namespace af

extern "C" int af_conv_variant(const af_conv_desc* d) {
    AF_REQUIRE(d && d->cout > 0, "conv_variant: bad descriptor");
    return af::pick_variant(d->cout);
}

extern "C" const char* af_conv_variant_name(int variant) {
    return (variant >= 0 && variant < 2) ? af::kVariantNames[variant] : "?";
}

extern "C" int af_conv3d_bn_act(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale,
                                const float* shift, const void* residual, void* out, int out_ld, void* stream) {
    using namespace af;
    AF_REQUIRE(d && in && w_packed && scale && shift && out, "conv: null argument");
    AF_REQUIRE(dtype_ok(d->dtype), "conv: bad dtype %d", d->dtype);
    AF_REQUIRE(d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "conv: bad dims");
    AF_REQUIRE(d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0, "conv: bad kernel/stride");
    AF_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0, "conv: negative padding");
    const int to = (d->t + 2 * d->pt - d->kt) / d->st + 1, ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1,
              wo = (d->w + 2 * d->pw - d->kw) / d->sw + 1;
    AF_REQUIRE(to == d->to && ho == d->ho && wo == d->wo && to > 0 && ho > 0 && wo > 0,
               "conv: output dims (%d,%d,%d) do not match the descriptor (%d,%d,%d)", to, ho, wo, d->to, d->ho, d->wo);
    const int bk = d->dtype == AF_F32 ? 32 : 64;
    AF_REQUIRE(d->cin % bk == 0, "conv: cin=%d must be a multiple of %d for this dtype", d->cin, bk);
    AF_REQUIRE(d->cout % 64 == 0, "conv: cout=%d must be a multiple of 64", d->cout);
    if (out_ld == 0) out_ld = d->cout;
    AF_REQUIRE(out_ld >= d->cout && out_ld % 4 == 0, "conv: bad out_ld %d", out_ld);
    AF_REQUIRE(aligned16(in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out) &&
                   aligned16(residual), "conv: buffers must be 16-byte aligned");
    AF_REQUIRE((long long)d->cout * d->kt * d->kh * d->kw * d->cin < (1LL << 31), "conv: weight too large");

    ConvArgs a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift;
    a.res = (const char*)residual; a.out = (char*)out;
    a.T = d->t; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout;
    a.kt = d->kt; a.kh = d->kh; a.kw = d->kw; a.st = d->st; a.sh = d->sh; a.sw = d->sw;
    a.pt = d->pt; a.ph = d->ph; a.pw = d->pw; a.To = to; a.Ho = ho; a.Wo = wo;
    a.relu = d->relu; a.out_ld = out_ld;
    a.M = (long long)d->n * to * ho * wo;
    hipStream_t s = (hipStream_t)stream;
    switch (d->dtype) {
        case AF_F32: return dispatch<AF_F32>(a, s);
        case AF_BF16: return dispatch<AF_BF16>(a, s);
        default: return dispatch<AF_F16>(a, s);
    }
}

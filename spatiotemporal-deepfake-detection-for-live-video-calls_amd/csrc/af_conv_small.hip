// Conv3d + BN [+ residual] [+ ReLU] for NARROW layers: at most 16 output channels and at most 32 16-byte K chunks
// (taps x Cin / 8, plus an optional 1x1x1 second segment) - the Fast pathway of SlowFast (8 / 16 / 32 channels,
// reference altfreezing/slowfast/models/video_model_builder.py:146-387 with beta_inv = 8) and its laterals.
//
// The generic implicit GEMM pads such a layer to 64 x 64 channels and stages 128-byte rows of which 16-64 bytes are
// real: 64x too many MACs and, worse, a full LDS pipeline for a layer whose whole working set is a few bytes per
// position (s2 Fast `b` conv: 0.31 ms for 51 MB of traffic).  Here the layer is a direct-gather MFMA:
//   * K is the list of 16-byte chunks (tap, 8 input channels), 4 chunks per v_mfma_f32_16x16x32: lane (position frow,
//     k-group fg) fetches ITS chunk of ITS position straight from the NDHWC tensor with one 16-byte buffer load
//     (neighbouring positions = neighbouring addresses; padding taps = out-of-range offsets = zeros); no LDS at all;
//   * the weights (at most 16 x 256) sit in registers as A fragments for the life of the wave, read once from the
//     generic kernel's packed layout ([cout_pad64][tap][cin_pad64]: no second packer);
//   * a wave owns 64 positions (4 MFMA tiles) per step, all their gathers in flight together; the epilogue applies
//     scale / shift (+ residual) (+ ReLU) per lane - 4 consecutive channels - and stores 8 bytes.
#include "af_common.h"

namespace af {

constexpr int SMALL_MAXKS = 8;          // MFMA K-steps: 32 chunks = 256 K values

struct SmallArgs {
    const char* in; const char* w; const char* in2; const char* w2;
    const float* scale; const float* shift; const char* res; char* out;
    int T, H, W, Cin, kt, kh, kw, st, sh, sw, pt, ph, pw, To, Ho, Wo, Cout;
    int CinP;               // padded channels per tap in the packed weight rows
    long long Kw;           // packed weight row length (elements)
    int T2, H2, W2, Cin2, st2, sh2, sw2, Cin2P;
    int nch1, nch2;         // 16-byte chunks of the two K segments
    int relu, out_ld;
    long long M;
};

template <int DT, int TN>
__global__ __launch_bounds__(256) void conv_small_kernel(const SmallArgs a) {
    typedef Elem<DT> E;
    static_assert(E::EPC == 8, "16-bit operands");
    constexpr int TM = 4;
    const int lane = threadIdx.x & 63, frow = lane & 15, fg = lane >> 4;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const int nch = a.nch1 + a.nch2, ksteps = (nch + 3) >> 2;
    const int cpt = a.Cin >> 3;                                  // chunks per tap

    // ---- weights -> registers (A operand: lane = (channel row, k-group))
    uint4 wf[SMALL_MAXKS][TN];
#pragma unroll
    for (int ks = 0; ks < SMALL_MAXKS; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int q = ks * 4 + fg, co = i * 16 + frow;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (q < a.nch1) v = *reinterpret_cast<const uint4*>(a.w + (co * a.Kw + (long long)(q / cpt) * a.CinP + (q % cpt) * 8) * 2);
            else if (q < nch) v = *reinterpret_cast<const uint4*>(a.w2 + ((long long)co * a.Cin2P + (q - a.nch1) * 8) * 2);
            wf[ks][i] = v;
        }
    f32x4 sc[TN], sf[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + i * 16 + fg * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + i * 16 + fg * 4);
    }
    // this lane's chunk of every K-step: which tensor, which tap, which 8 channels
    int c_dt[SMALL_MAXKS], c_dh[SMALL_MAXKS], c_dw[SMALL_MAXKS], c_co[SMALL_MAXKS];      // c_dt < 0: second segment; -2: none
#pragma unroll
    for (int ks = 0; ks < SMALL_MAXKS; ++ks) {
        const int q = ks * 4 + fg;
        if (q < a.nch1) {
            const int tap = q / cpt;
            c_co[ks] = (q % cpt) * 8;
            c_dw[ks] = tap % a.kw; c_dh[ks] = (tap / a.kw) % a.kh; c_dt[ks] = tap / (a.kw * a.kh);
        } else {
            c_dt[ks] = q < nch ? -1 : -2; c_dh[ks] = c_dw[ks] = 0; c_co[ks] = (q - a.nch1) * 8;
        }
    }
    const __amdgpu_buffer_rsrc_t d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in), (short)0, (int)kOutOfRange, 0x00020000);
    const __amdgpu_buffer_rsrc_t d2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in2 ? a.in2 : a.in), (short)0, (int)kOutOfRange, 0x00020000);

    const long long groups = (a.M + 16 * TM - 1) / (16 * TM);
    for (long long g = wave0; g < groups; g += nwaves) {
        f32x4 acc[TN][TM];
        uint4 bf[SMALL_MAXKS][TM];
        long long mpos[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            long long m = g * (16 * TM) + j * 16 + frow;
            mpos[j] = m;
            const bool live = m < a.M;
            if (!live) m = a.M - 1;
            const int wo = (int)(m % a.Wo); long long t1 = m / a.Wo;
            const int ho = (int)(t1 % a.Ho); long long t2 = t1 / a.Ho;
            const int to = (int)(t2 % a.To); const long long n = t2 / a.To;
#pragma unroll
            for (int ks = 0; ks < SMALL_MAXKS; ++ks) {
                bf[ks][j] = uint4{0u, 0u, 0u, 0u};
                if (ks < ksteps) {
                    unsigned off = kOutOfRange;
                    if (c_dt[ks] >= 0) {
                        const int ti = to * a.st - a.pt + c_dt[ks], hi = ho * a.sh - a.ph + c_dh[ks], wi = wo * a.sw - a.pw + c_dw[ks];
                        if (live && (unsigned)ti < (unsigned)a.T && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
                            off = (unsigned)(((((n * a.T + ti) * a.H + hi) * a.W + wi) * a.Cin + c_co[ks]) * 2);
                        bf[ks][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(d1, off, 0, 0));
                    } else if (c_dt[ks] == -1) {
                        if (live)
                            off = (unsigned)(((((n * a.T2 + to * a.st2) * a.H2 + ho * a.sh2) * a.W2 + wo * a.sw2) * a.Cin2 + c_co[ks]) * 2);
                        bf[ks][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(d2, off, 0, 0));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < SMALL_MAXKS; ++ks)
            if (ks < ksteps) {
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int i = 0; i < TN; ++i) Mma<DT>::run(wf[ks][i], bf[ks][j], acc[i][j]);
            }
        // ---- epilogue: lane = 4 consecutive channels of one position
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const long long m = mpos[j];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int ch = i * 16 + fg * 4;
                if (m < a.M && ch < a.Cout) {
                    f32x4 v = acc[i][j] * sc[i] + sf[i];
                    if (a.res) v += Vec4<DT>::load(a.res + (m * a.Cout + ch) * 2);
                    if (a.relu) { v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]); }
                    Vec4<DT>::store(a.out + (m * a.out_ld + ch) * 2, v);
                }
            }
        }
    }
}

// true iff this layer takes the narrow-layer path (also used by af_conv_variant)
bool conv_small_applies(const af_conv_desc* d, const af_conv_desc* d2, const void* residual, int out_ld) {
    if (d->dtype == AF_F32 || d->tpool) return false;
    // (measured: at 32 output channels - the Fast pathway's `c` convs - the generic 64-wide tile with its 16-byte row
    // stores is faster than this kernel's 8-byte per-lane stores)
    if (d->cout > 16 || d->cout % 4 != 0 || d->cin % 8 != 0 || d->cin > 64) return false;
    const int nch1 = d->kt * d->kh * d->kw * (d->cin / 8);
    int nch2 = 0;
    if (d2) {
        if (d2->cin % 8 != 0) return false;
        nch2 = d2->cin / 8;
        if ((long long)d2->n * d2->t * d2->h * d2->w * d2->cin * 2 >= (1LL << 31)) return false;
    }
    if (nch1 + nch2 > 4 * SMALL_MAXKS) return false;
    if ((long long)d->n * d->t * d->h * d->w * d->cin * 2 >= (1LL << 31)) return false;       // 32-bit buffer offsets
    return true;
}

int conv_small_run(const af_conv_desc* d, const void* in, const void* w_packed, const af_conv_desc* d2, const void* in2,
                   const void* w2_packed, const float* scale, const float* shift, const void* residual, void* out, int out_ld,
                   hipStream_t stream) {
    SmallArgs a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.in2 = (const char*)in2; a.w2 = (const char*)w2_packed;
    a.scale = scale; a.shift = shift; a.res = (const char*)residual; a.out = (char*)out;
    a.T = d->t; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
    a.st = d->st; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
    a.To = d->to; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
    a.CinP = (d->cin + 63) / 64 * 64;
    a.Kw = (long long)d->kt * d->kh * d->kw * a.CinP;
    a.T2 = a.H2 = a.W2 = 1; a.Cin2 = 8; a.st2 = a.sh2 = a.sw2 = 1; a.Cin2P = 64;
    a.nch1 = d->kt * d->kh * d->kw * (d->cin / 8); a.nch2 = 0;
    if (d2) {
        a.T2 = d2->t; a.H2 = d2->h; a.W2 = d2->w; a.Cin2 = d2->cin; a.st2 = d2->st; a.sh2 = d2->sh; a.sw2 = d2->sw;
        a.Cin2P = (d2->cin + 63) / 64 * 64; a.nch2 = d2->cin / 8;
    }
    a.relu = d->relu; a.out_ld = out_ld;
    a.M = (long long)d->n * d->to * d->ho * d->wo;
    long long blocks = (a.M + 255) / 256;                       // one 64-position step per wave, then grid-stride
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (d->dtype == AF_BF16) hipLaunchKernelGGL((conv_small_kernel<AF_BF16, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_small_kernel<AF_F16, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    AF_CHECK_LAUNCH("conv_small_kernel");
    return AF_OK;
}

}  // namespace af

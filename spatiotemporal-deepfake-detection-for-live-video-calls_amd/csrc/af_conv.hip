// Implicit-GEMM Conv3d + BatchNorm(scale/shift) [+ residual] [+ ReLU] for NDHWC activations.
//
// Replaces, per launch, nn.Conv3d(bias=False) -> nn.BatchNorm3d(eval) [-> add] [-> nn.ReLU] of the
// reference bottleneck / res block (altfreezing/slowfast/models/resnet_helper.py:267-325, 411-444).
//
// GEMM view:  D[cout][pos] = sum_k  Wp[cout][k] * X[pos][k],   k = (tap, cin),  pos = (n,to,ho,wo)
//   * weights are the MFMA "A" operand (rows = output channels) and im2col'ed activations the "B"
//     operand (cols = output positions): the 16x16 accumulator then holds 4 CONSECUTIVE CHANNELS
//     of one position per lane, so BN scale/shift are per-lane vectors; a per-wave LDS patch then
//     turns the sub-tile into whole NDHWC rows (16 B per lane, full 128-byte lines).
//   * both operands are K-contiguous in memory ([cout][tap][cin] weights, NDHWC activations), so a
//     K-step of one tile row is one contiguous 128-byte run = 8 x 16-byte chunks.
//   * LDS tiles are [row][8 chunks] with chunk ^= (row & 7): conflict-free ds_write_b128 from the
//     staging pass and conflict-free ds_read_b128 for the MFMA fragments (bank math in DESIGN.md).
//   * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, no VGPR staging and
//     no ds_write): one wave-instruction fills 8 tile rows; the LDS image is lane-linear, so the
//     swizzle is applied on the SOURCE side (lane (row, slot) fetches chunk slot ^ (row & 7)).
//     Addressing is buffer-style: a per-workgroup descriptor, a per-lane 32-bit row offset that never
//     changes, and the K-step's (tap, channel-slab) offset in an SGPR - no address VALU in the K loop.
//   * zero padding: an out-of-bounds tap gets a per-lane offset beyond the descriptor's range and the
//     hardware returns zeros (per-row validity bit per tap, computed once); no halo in HBM, no branch.
//   * pipeline: 3-slot LDS ring, K-steps s+1 and s+2 in flight while step s is multiplied; one raw
//     s_barrier per K-step; explicit counted s_waitcnt vmcnt (hipcc does not see the DMA loads).
#include "af_common.h"
#include <stdlib.h>

namespace af {

// Diagnostic build (-DAF_STAMPS, tools/stamps_lib.sh; never the shipped library): shader-clock stamps around the phases of a tile
#ifdef AF_STAMPS
#define AF_STAMP_DECL unsigned long long stamp_v[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define AF_DBG(bit) (a.dbg & (bit))
#define AF_STAMP(slot) stamp_v[slot] = (slot) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime()
#define AF_STAMP_FLUSH do { if (a.stamps && lane < 8) a.stamps[((long long)bid * 8 + wave) * 8 + lane] = \
    lane == 0 ? stamp_v[0] : lane == 1 ? stamp_v[1] : lane == 2 ? stamp_v[2] : lane == 3 ? stamp_v[3] : lane == 4 ? stamp_v[4] : lane == 5 ? stamp_v[5] : lane == 6 ? stamp_v[6] : stamp_v[7]; } while (0)
#else
#define AF_STAMP_DECL do {} while (0)
#define AF_DBG(bit) false
#define AF_STAMP(slot) do {} while (0)
#define AF_STAMP_FLUSH do {} while (0)
#endif

struct ConvArgs {
    const char* in;
    const char* w;
    const float* scale;
    const float* shift;
    const char* res;
    char* out;
    int T, H, W, Cin, Cout;
    int CinP, CoutP;    // Cin rounded up to the K-step (zero-padded weight columns), Cout rounded up to 64 (zero rows)
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int To, Ho, Wo;
    int relu, out_ld;
    int ring;           // LDS ring slots used by the lean K loop: 3, or 2 (mid-K HBM-bound layers: 2 workgroups per CU)
    int tpool;          // 1: fuse MaxPool3d([2,1,1],[2,1,1]) over output frame pairs into the epilogue;
                        // 2: fuse MaxPool3d([1,2,2],[1,2,2]) over 2x2 output pixels (FTCN-TT's replaced strides)
    long long M;        // N*To*Ho*Wo
    int tiles_n;        // Cout / BN
    int kpt;            // K-steps per tap = Cin / BK
    // optional second K segment: a 1x1x1 (strided) conv over another input that lands on the same output
    // positions - the projection shortcut of a res block, accumulated into the same tile (kpt2 == 0: none)
    const char* in2;
    const char* w2;
    int T2, H2, W2, Cin2, Cin2P, st2, sh2, sw2, kpt2;
    // split-K (small batches: a long-K layer with a handful of tiles): workgroup (tile, blockIdx.y) multiplies K-steps
    // [y * S / ksplit, (y + 1) * S / ksplit) and leaves raw fp32 partial sums in ws[y][M][Cout]; splitk_finish_kernel adds them
    int ksplit;
    // row index -> (n, to, ho, wo): division by the three output extents as multiply-high + shift (exact for 32-bit numerators;
    // Granlund-Montgomery: l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, q = (t + ((n - t) >> 1)) >> (l - 1), t = mulhi(m, n))
    unsigned div_w_m, div_w_l, div_h_m, div_h_l, div_t_m, div_t_l;
    float* ws;
    long long ws_bytes;  // caller's workspace (ws == nullptr or too small: no split)
#ifdef AF_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/stamps_lib.sh): [workgroup][wave][8] shader-clock / wall-clock stamps
    int dbg;                      // timing-only ablations of the MFMA-bound K loops (AF_G_DBG): 1 no vmcnt wait, 2 no barrier, 4 no DMA
#endif
};

// NW = WN*WM*KS = 8 waves (512 threads); wave (wn, wm) owns a (BN/WN) x (BM/WM) sub-tile of 16x16
// MFMA tiles.
// KS = 2 splits each stage's K between wave groups 0-3 / 4-7 (used for Cout = 64: every wave then owns a
// 64x64 sub-tile, halving LDS fragment traffic per MFMA; the two partial sums meet in LDS after the loop).
// DUAL compiles in the second K segment (projection shortcut accumulated into the same tile).
// BMR < BM: the tile covers BMR output positions (frame-aligned tile counts: 224 rows = 224 tiles for the 50 176 positions
// of a 16-clip s4 / s5 tensor instead of 196 on 256 CUs); its LDS image keeps BM rows, rows BMR.. are fetched as zeros.
// One tile (bid of nb) of the layer.
template <int DT, int BN, int BM, int WN, int WM, int KS, int NSTAGE, int MINW, bool DUAL, bool SPLITK, int BMR>
__device__ __forceinline__ void conv_igemm_tile(const ConvArgs& a, const int bid, const int nb) {
    typedef Elem<DT> E;
    constexpr int EPC = E::EPC;            // elements per 16-byte chunk
    constexpr int ES = 16 / EPC;           // bytes per element
    constexpr int WTN = BN / WN, WTM = BMR / WM;
    constexpr int TN = WTN / 16, TM = WTM / 16;
    constexpr int NW = WN * WM * KS;                 // waves per workgroup
    constexpr int GR = NW * 8;                       // tile rows one pass of the workgroup stages (8 lanes per row)
    constexpr int RW = BN / GR, RX = BM / GR;        // tile rows (= LDS-DMA instructions) per thread per stage
    constexpr int PER_WAVE = RW + RX;                // LDS-DMA instructions a wave issues per stage
    constexpr int STAGE_BYTES = (BN + BM) * 128;
    static_assert(NW == 8 && (KS == 1 || KS == 2) && BN % GR == 0 && BM % GR == 0, "8 waves per workgroup");
    static_assert(BMR <= BM && BMR % (WM * 16) == 0, "real tile rows");
    constexpr bool LEAN = MINW >= 4;
    typedef Mma<DT> MMA;
    static_assert(NSTAGE == 3 || (NSTAGE == 2 && !LEAN && KS == 1), "ring depth of the pipelined loop: 3 or 2 slots");

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    AF_STAMP_DECL;
    AF_STAMP(4);                                       // (diagnostic build: entry; stamp 0 sits behind the address set-up)

    // ---- workgroup -> tile, XCD-contiguous (bijective remap; placement is a speed matter only)
    const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = swz % a.tiles_n;
    const int tile_m = swz / a.tiles_n;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgroup = wave / (WN * WM), wsub = wave % (WN * WM);
    const int wn = wsub % WN, wm = wsub / WN;
    const int lrow = tid >> 3;                         // 0..GR-1: row inside a GR-row group
    const int chunk = (tid & 7) ^ (lrow & 7);          // source chunk that lands in LDS slot (tid & 7)

    // ---- per-thread staging rows: a 32-bit offset from the workgroup's origin + a validity bit per tap.
    // The origin is the (padding-shifted) first input element of tile row 0; every other row of the tile lies at a
    // non-negative offset from it (row offsets grow with (n, to, ho, wo), also in the pool-fused row order).
    const int taps = a.kt * a.kh * a.kw;
    const long long m0 = (long long)tile_m * BMR;
    auto row_offsets = [&](long long m, long long& off1, long long& off2, unsigned& mask) {
        // tile row -> output position.  With the temporal pool fused, rows are ordered (n, to/2, ho, wo, to%2):
        // the two frames a pool window spans are ADJACENT rows of the tile (the epilogue takes their max)
        // (spatial 2x2 pool fused: rows are ordered (n, to, ho/2, wo/2, dy, dx) - a window = 4 adjacent rows)
        // (round 4, late: 32-bit multiply-high divisions.  The three 64-bit div / mod pairs per row - five rows per thread - were the bulk
        //  of ~11 k cycles every workgroup spent in front of its first DMA: in-kernel stamps put a tile of s3's projection block at
        //  25.7 k cycles and the 7-tile launch at 115 us = 37 k per tile; persistent workgroups alone changed nothing, so it was not the
        //  dispatch.  M < 2^31 is checked on the host.)
        const unsigned mq = (unsigned)(a.tpool == 2 ? (m >> 2) : a.tpool ? (m >> 1) : m);
        const unsigned wdiv = (unsigned)(a.tpool == 2 ? (a.Wo >> 1) : a.Wo), hdiv = (unsigned)(a.tpool == 2 ? (a.Ho >> 1) : a.Ho);
        auto fdiv = [](unsigned x, unsigned mm, unsigned l) { const unsigned t = __umulhi(mm, x); return l == 0 ? x : (t + ((x - t) >> 1)) >> (l - 1); };
        const unsigned t1 = fdiv(mq, a.div_w_m, a.div_w_l);
        int wo = (int)(mq - t1 * wdiv);
        const unsigned t2 = fdiv(t1, a.div_h_m, a.div_h_l);
        int ho = (int)(t1 - t2 * hdiv);
        const unsigned tdiv = (unsigned)(a.tpool == 1 ? (a.To >> 1) : a.To);
        const unsigned nq = fdiv(t2, a.div_t_m, a.div_t_l);
        int to = (int)(t2 - nq * tdiv); const long long n = nq;
        if (a.tpool == 1) to = 2 * to + (int)(m & 1);
        if (a.tpool == 2) { ho = 2 * ho + (int)((m >> 1) & 1); wo = 2 * wo + (int)(m & 1); }
        const int ti0 = to * a.st - a.pt, hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
        off1 = ((((n * a.T + ti0) * a.H + hi0) * a.W + wi0) * a.Cin) * ES;
        off2 = DUAL ? ((((n * a.T2 + to * a.st2) * a.H2 + ho * a.sh2) * a.W2 + wo * a.sw2) * a.Cin2) * ES : 0;
        // bit (dt * kh + dh) * kw + dw: tap in bounds - built per axis (kt + kh + kw steps, not kt * kh * kw: this runs five times per
        // thread in front of a workgroup's first DMA)
        unsigned mw = 0, mhw = 0, mthw = 0;
        for (int dw = 0; dw < a.kw; ++dw) mw |= ((unsigned)(wi0 + dw) < (unsigned)a.W ? 1u : 0u) << dw;
        for (int dh = 0; dh < a.kh; ++dh) mhw |= ((unsigned)(hi0 + dh) < (unsigned)a.H ? mw : 0u) << (dh * a.kw);
        for (int dt = 0; dt < a.kt; ++dt) mthw |= ((unsigned)(ti0 + dt) < (unsigned)a.T ? mhw : 0u) << (dt * a.kh * a.kw);
        mask = (1u << 31) | mthw;
    };
    // (a 1x1x1 / stride-1 / unpadded / unpooled one-input layer - most `a` and `c` convs - reads row m at m * Cin: no (n, t, h, w) at all)
    const bool flat_rows = !DUAL && taps == 1 && a.tpool == 0 && a.st == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.ph == 0 && a.pw == 0;
    long long org1, org2; unsigned org_mask;
    if (flat_rows) { org1 = m0 * a.Cin * ES; org2 = 0; }
    else row_offsets(m0, org1, org2, org_mask);          // uniform: tile row 0 always exists
    const i32x4 xdesc = make_desc(a.in + org1);
    const i32x4 x2desc = make_desc((DUAL ? a.in2 : a.in) + org2);
    unsigned xoff[RX], x2off[RX];
    unsigned xmask[RX];                                 // bit t: tap t in bounds; bit 31: row < M
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const long long m = m0 + lrow + GR * i;
        xmask[i] = 0; xoff[i] = kOutOfRange; x2off[i] = kOutOfRange;
        if (m < a.M && (BMR == BM || lrow + GR * i < BMR)) {
            if (flat_rows) {
                xmask[i] = (1u << 31) | 1u;
                xoff[i] = (unsigned)((lrow + GR * i) * a.Cin * ES) + chunk * 16;
            } else {
                long long o1, o2;
                row_offsets(m, o1, o2, xmask[i]);
                xoff[i] = (unsigned)(o1 - org1) + chunk * 16;   // < 2 GiB (host-checked span)
                x2off[i] = (unsigned)(o2 - org2) + chunk * 16;
            }
        }
    }
    const long long Kw = (long long)taps * a.CinP;     // weight row length (elements, zero-padded per tap)
    // channel tails (Cin not a multiple of the K-step): this lane's 16-byte chunk of the LAST slab of a tap is
    // real only below Cin; beyond it the lane fetches zeros (the weights there are zero columns as well)
    constexpr int BKE = 8 * EPC;
    const bool clast = chunk * EPC < a.Cin - (a.kpt - 1) * BKE;
    const bool clast2 = DUAL && chunk * EPC < a.Cin2 - (a.kpt2 - 1) * BKE;
    const i32x4 wdesc = make_desc(a.w + (long long)tile_n * BN * Kw * ES);
    const i32x4 w2desc = make_desc(a.w2 + (long long)tile_n * BN * a.Cin2P * ES);
    unsigned woff[RW], w2off[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        woff[i] = (unsigned)(((lrow + GR * i) * Kw + chunk * EPC) * ES);          // < 2 GiB (host-checked weight size)
        w2off[i] = (unsigned)(((long long)(lrow + GR * i) * a.Cin2P + chunk * EPC) * ES);
    }

    // ---- LDS-DMA producer: stage `st` <- K-step (tap, kc), kc fastest: the 64-channel slabs of one tap are
    // consecutive 128-byte pieces of the same NDHWC rows, so a row's channels are fetched in back-to-back steps
    // (DRAM/L2 friendly); measured better than taps-inner, which scattered the 3x1x1 layers' reads over frames.
    // A stage is PER_WAVE pieces per wave (weights rows first, then activation rows); pieces can be issued one
    // at a time so that the main loop can tuck them between MFMA groups.
    int dt = 0, dh = 0, dw = 0, kc = 0, tap = 0;
    auto issue_piece = [&](int st, int g) {
        const unsigned base = lds0 + st * STAGE_BYTES + wave * (8 * 128) + g * (GR * 128);
        if (DUAL && tap >= taps) {                               // second segment (projection shortcut)
            const int off2 = kc * 128;
            if (g < RW) blds16(w2off[g], w2desc, off2, base);
            else blds16(((xmask[g - RW] >> 31) && (kc + 1 < a.kpt2 || clast2)) ? x2off[g - RW] : kOutOfRange, x2desc, off2, base);
        } else if (g < RW) {
            blds16(woff[g], wdesc, tap * a.CinP * ES + kc * 128, base);
        } else {
            const int xsoff = ((dt * a.H + dh) * a.W + dw) * a.Cin * ES + kc * 128;   // < 2^31 (host-checked)
            const bool ok = ((xmask[g - RW] >> tap) & 1u) && (kc + 1 < a.kpt || clast);
            blds16(ok ? xoff[g - RW] : kOutOfRange, xdesc, xsoff, base);
        }
    };
    auto advance = [&]() {
        ++kc;
        if (DUAL && tap >= taps) return;
        if (kc == a.kpt) {
            kc = 0; ++tap;
            if (++dw == a.kw) { dw = 0; if (++dh == a.kh) { dh = 0; ++dt; } }
        }
    };
    auto issue_stage = [&](int st) {
#pragma unroll
        for (int g = 0; g < PER_WAVE; ++g) issue_piece(st, g);
        advance();
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int S_all = taps * a.kpt + (DUAL ? a.kpt2 : 0);
    // SPLITK instantiations only: this workgroup's K range (>= 1 step: the host keeps ksplit <= S_all / 8)
    const int s_lo = SPLITK ? (int)((long long)blockIdx.y * S_all / a.ksplit) : 0;
    const int S = (SPLITK ? (int)((long long)(blockIdx.y + 1) * S_all / a.ksplit) : S_all) - s_lo;
    if (SPLITK) for (int i = 0; i < s_lo; ++i) advance();
    AF_STAMP(0); AF_STAMP(6);
    issue_stage(0);
    if (S > 1 && !(LEAN && a.ring == 2)) issue_stage(1);

    AF_STAMP(1);
    const int frow = lane & 15, fg = lane >> 4;
    constexpr int NKK = 2 / KS;                                   // k-halves of a stage this wave multiplies
    const int kk0 = KS == 2 ? kgroup : 0;
    const int wrow = (wn * WTN + frow) * 8, xrow = BN * 8 + (wm * WTM + frow) * 8;
    if constexpr (LEAN || NKK == 1 || TM < 4) {
        // HBM-bound short-K variants (and the K-split layout): smallest register footprint, one barrier per
        // K-step, all DMA pieces of stage s+2 issued right after it.
        auto multiply = [&](int st_) {
            const uint4* ws = smem + st_ * (STAGE_BYTES / 16) + wrow;
            const uint4* xs = smem + st_ * (STAGE_BYTES / 16) + xrow;
#pragma unroll
            for (int q = 0; q < NKK; ++q) {
                const int c = ((kk0 + q) * 4 + fg) ^ (frow & 7);
                uint4 af[TN], bf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) af[i] = ws[i * 16 * 8 + c];
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[j] = xs[j * 16 * 8 + c];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) MMA::run(af[i], bf[j], acc[i][j]);
            }
        };
        if (LEAN && a.ring == 2) {
            // two slots, one K-step of look-ahead: half the LDS, so two workgroups share a CU and cover each
            // other's load / epilogue phases (4..8-step layers that stream a wide output)
            for (int s = 0; s < S; ++s) {
                wait_vmcnt<0>();                                  // stage s (the only one in flight) has landed
                __builtin_amdgcn_s_barrier();                     // ... for everyone; slot (s+1)&1 is no longer read
                if (s + 1 < S) issue_stage((s + 1) & 1);
                multiply(s & 1);
            }
        } else {
            int st = 0;
            for (int s = 0; s < S; ++s) {
                // stage s has landed for this wave once only the younger stage (s+1) is still in flight ...
                if (s + 1 < S) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>();
                // ... and for every wave after the barrier, which also retires all reads of slot (s-1)%3
                __builtin_amdgcn_s_barrier();
                if (s + 2 < S) issue_stage(st == 0 ? 2 : st - 1);    // slot (s+2)%3 == (s-1)%3 is free now
                multiply(st);
                st = (st == 2) ? 0 : st + 1;
            }
        }
#ifdef AF_NO_ASM_IGEMM
    } else if constexpr (false) {
#else
    } else if constexpr (DT != AF_F32 && !DUAL && !(LEAN || NKK == 1 || TM < 4)) {
#endif
        // ---- MFMA-bound variants, 16-bit operands, one input (round 4; the projection blocks' second K segment doubles the
        // descriptors and offsets the DMA issue selects from - 106 scalar registers and spills next to the DMA - and keeps the builtin loop): the K loop as asm statements, issued in program order - MFMAs
        // accumulating in place, LDS fragment reads hipcc does not count, LDS-DMA, the kernel's own waits (af_common.h) - on the
        // structure conv133g measured its way to (in-kernel stamps): software-pipelined by k-halves; the NR = TN + TM
        // fragment reads of the NEXT half ride on the first NR MFMAs of a group, one ds_read_b128 per MFMA, so no read burst stands
        // between a barrier and the matrix pipe; the DMA pieces of stage s + NSTAGE follow, spread over the rest of the second
        // group; waves 4-7 (the SIMD partners of 0-3) take the step's one barrier H1 MFMAs into their first group instead of behind
        // it.  One instruction stream for both kinds of wave and for the last step (separate paths would meet in TN * TM
        // accumulator phis).  What hipcc had made of the builtin form of this loop ran at 0.79 of the MFMA rate with every
        // memory operation and barrier switched off (AF_G_DBG = 7 of the stamp build), 0.67 with them.
        constexpr int NTH = TN * TM, NR = TN + TM;
        constexpr int H1 = 3 * NTH / 4;
        static_assert(NR <= H1 && NR + PER_WAVE <= NTH, "reads in front of a LATE wave's barrier; a DMA piece or a read per MFMA");
        u32x4 a0[TN], b0[TM], a1[TN], b1[TM];
        const unsigned wb[2] = {lds0 + (unsigned)((wn * WTN + frow) * 128 + (((0 + fg) ^ (frow & 7)) << 4)),
                                lds0 + (unsigned)((wn * WTN + frow) * 128 + (((4 + fg) ^ (frow & 7)) << 4))};
        const unsigned xb[2] = {lds0 + (unsigned)((BN + wm * WTM + frow) * 128 + (((0 + fg) ^ (frow & 7)) << 4)),
                                lds0 + (unsigned)((BN + wm * WTM + frow) * 128 + (((4 + fg) ^ (frow & 7)) << 4))};
        auto frag_read = [&](u32x4 (&af)[TN], u32x4 (&bf)[TM], unsigned wsb, unsigned xsb, auto idx) {
            constexpr int r = decltype(idx)::value;
            if constexpr (r < TN) af[r] = lds_read16_uncounted<r * (16 * 128)>(wsb);
            else if constexpr (r < NR) bf[r - TN] = lds_read16_uncounted<(r - TN) * (16 * 128)>(xsb);
        };
        auto pin_half = [&](u32x4 (&af)[TN], u32x4 (&bf)[TM]) {
#pragma unroll
            for (int i = 0; i < TN; ++i) pin_frag(af[i]);
#pragma unroll
            for (int j = 0; j < TM; ++j) pin_frag(bf[j]);
        };
        // MFMAs [T0, T1) of a group on (af, bf); with READS, read r of the next fragment set is issued behind MFMA r
        auto mma_group = [&](auto t0c, auto t1c, auto readsc, const u32x4 (&af)[TN], const u32x4 (&bf)[TM], u32x4 (&naf)[TN], u32x4 (&nbf)[TM],
                             unsigned wsb, unsigned xsb) {
            constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
            static_for<T1 - T0>([&](auto tt) {
                constexpr int t = T0 + decltype(tt)::value;
                MmaAsm<DT>::run(af[t / TM], bf[t % TM], acc[t / TM][t % TM]);
                if constexpr (decltype(readsc)::value && t < NR) frag_read(naf, nbf, wsb, xsb, IC<t>{});
            });
        };
        // prologue: (the generic prologue above issued stages 0 and 1) stage 2 of a three-slot ring, then stage 0 landed - vmcnt counts
        // in issue order: all but the younger stages
        if (NSTAGE == 3 && S > 2) issue_stage(2);
        if (S >= NSTAGE) wait_vmcnt<(NSTAGE - 1) * PER_WAVE>(); else if (S == 2) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        static_for<NR>([&](auto r) { frag_read(a0, b0, wb[0], xb[0], r); });
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc_live(acc[i][j]);
        mfma_operands_settled();
        const bool late = wave >= 4;
        int stoff = 0;                                             // ring slot of stage s (bytes)
        // (No LDS read is left pending across the loop's back edge - the wait for the look-ahead reads closes the body - and the
        //  loop is neither unrolled nor peeled: hipcc reconciles the register assignment of a peeled first iteration with the
        //  steady loop by copying every live register, fragments whose data has not landed included.  tests/test_host_cpu.py
        //  lints the generated code for both.)
        wait_lgkmcnt<0>();
        pin_half(a0, b0);
#pragma nounroll
        for (int s = 0; s < S; ++s) {
            const bool laststep = s + 1 == S;
            const int stnext = stoff == (NSTAGE - 1) * STAGE_BYTES ? 0 : stoff + STAGE_BYTES;
            // first group: MFMA(a0, b0) [stage s, k-half 0] while the fragments of k-half 1 come in
            // what must have landed in front of the barrier: stage s + 1.  Two slots: the only stage in flight.  Three slots: the
            // pieces of stage s + 2 (or their empty stand-ins) were issued behind it in the previous step and stay in flight; at
            // K-step 0 the younger operations are the prologue's (stage 2 if the layer has one).
            auto wait_stage = [&]() {
                if (AF_DBG(1)) return;
                if (NSTAGE == 2 || (s == 0 && S <= 2)) wait_vmcnt<0>(); else wait_vmcnt<PER_WAVE>();
            };
            mma_group(IC<0>{}, IC<H1>{}, IC<1>{}, a0, b0, a1, b1, wb[1] + stoff, xb[1] + stoff);
            if (late && !laststep) {
                wait_lgkmcnt<0>();
                wait_stage();
                if (!AF_DBG(2)) __builtin_amdgcn_s_barrier();
            }
            mma_group(IC<H1>{}, IC<NTH>{}, IC<0>{}, a0, b0, a1, b1, 0u, 0u);
            wait_lgkmcnt<0>();
            if (!late && !laststep) {
                wait_stage();
                if (!AF_DBG(2)) __builtin_amdgcn_s_barrier();
            }
            pin_half(a1, b1);
            // second group: MFMA(a1, b1) [stage s, k-half 1]; behind its MFMAs, one each, the DMA pieces of stage s + NSTAGE into the slot
            // this step has just left and the fragment reads of stage s + 1, k-half 0 (group2_slot: piece, read, read, piece, ...;
            // behind the last step: unused reads of valid LDS).
            // (The stage's scalar offsets are computed once, not per piece; past the last refill the pieces are still issued, through
            //  an empty descriptor - every lane out of range: zeros into a slot nobody reads again - so the step has no branch per
            //  piece and every step puts PER_WAVE operations on the vmcnt queue.)
            const bool refill = s + NSTAGE < S && !AF_DBG(4);
            const int wsoff = __builtin_amdgcn_readfirstlane(tap * a.CinP * ES + kc * 128);
            const int xsoff = __builtin_amdgcn_readfirstlane(((dt * a.H + dh) * a.W + dw) * a.Cin * ES + kc * 128);
            i32x4 wd = wdesc, xd = xdesc;
            if (!refill) { wd[2] = 0; xd[2] = 0; }
            const bool lastk = kc + 1 < a.kpt || clast;            // this lane's chunk of the slab is real
            const unsigned tapbit = (unsigned)tap;
            // (the last step has no barrier between one wave's DMA issue and another wave's reads of this step's slot: its empty
            //  pieces go to the NEXT slot, which nobody reads or refills any more)
            const unsigned dst0 = lds0 + (laststep ? stnext : stoff) + wave * (8 * 128);
            const unsigned wsb0 = wb[0] + stnext, xsb0 = xb[0] + stnext;
            static_for<NTH>([&](auto tc) {
                constexpr int t = tc, slot = group2_slot(t, PER_WAVE, NR);
                MmaAsm<DT>::run(a1[t / TM], b1[t % TM], acc[t / TM][t % TM]);
                if constexpr (slot != kNoSlot && slot < 0) frag_read(a0, b0, wsb0, xsb0, IC<-1 - slot>{});
                else if constexpr (slot != kNoSlot) {
                    constexpr int g = slot;
                    if constexpr (g < RW) blds16_m0(woff[g], wd, wsoff, __builtin_amdgcn_readfirstlane(dst0 + g * (GR * 128)));
                    else {
                        const bool ok = ((xmask[g - RW] >> tapbit) & 1u) && lastk;
                        blds16_m0(ok ? xoff[g - RW] : kOutOfRange, xd, xsoff, __builtin_amdgcn_readfirstlane(dst0 + g * (GR * 128)));
                    }
                }
            });
            if (refill) advance();
            stoff = stnext;
            wait_lgkmcnt<0>();                                     // a0 / b0 of the next step (read under this group) are back
            pin_half(a0, b0);
        }
        wait_vmcnt<0>();                                           // (the empty stand-in pieces of the last steps write zeros into the ring: the epilogue reuses it)
        mfma_drain();                                              // the accumulators are read by ordinary vector code from here on
    } else {
        // MFMA-bound variants, fp32 (four exact-fp32 MFMAs per chunk) and the two-input projection blocks: software-pipelined by k-HALVES.  The fragments of a half are read from LDS while
        // the MFMAs of the previous half run, so the matrix pipe never waits for the LDS fill that follows a
        // barrier; the barrier (+ the counted vmcnt that makes stage s+1 visible) sits between the two MFMA
        // groups of a stage, and the LDS-DMA pieces of the look-ahead stage are tucked between the MFMAs of the
        // second group.
        constexpr int NTH = TN * TM;                              // tile products per k-half
        constexpr int MPG = (NTH + PER_WAVE - 1) / PER_WAVE;      // products between two DMA pieces
        uint4 a0[TN], b0[TM], a1[TN], b1[TM];
        auto read_half = [&](uint4 (&af)[TN], uint4 (&bf)[TM], int st_, int kk) {
            const int c = (kk * 4 + fg) ^ (frow & 7);
            const uint4* ws = smem + st_ * (STAGE_BYTES / 16) + wrow;
            const uint4* xs = smem + st_ * (STAGE_BYTES / 16) + xrow;
#pragma unroll
            for (int i = 0; i < TN; ++i) af[i] = ws[i * 16 * 8 + c];
#pragma unroll
            for (int j = 0; j < TM; ++j) bf[j] = xs[j * 16 * 8 + c];
        };
        int st = 0;
        if (NSTAGE == 2) {
            // two slots (the 256x256 tile: 64 KB per stage); stage s lives in slot s & 1.  Once the mid-step barrier
            // of stage s has passed, every fragment of stage s sits in registers, so its slot is refilled with stage
            // s+2 during the SECOND MFMA group; that stage is needed at the mid-step barrier of s+1.
            if (S > 1) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>();         // prologue: stage 0 landed
            __builtin_amdgcn_s_barrier();
            read_half(a0, b0, 0, 0);
            // all stages but the last (the last one is peeled: no barrier / look-ahead read in it, and - just as
            // important - no control-flow merge in the steady-state body, which would make hipcc's LDS wait counts
            // conservative and stall the second MFMA group on the look-ahead reads)
            for (int s = 0; s + 1 < S; ++s) {
                const bool refill = s + 2 < S;
                const int st1 = st ^ 1;
                read_half(a1, b1, st, 1);
                __builtin_amdgcn_sched_barrier(0);
                // ---- first half: MFMA(a0,b0) runs under the LDS reads just issued
#pragma unroll
                for (int t = 0; t < NTH; ++t) MMA::run(a0[t / TM], b0[t % TM], acc[t / TM][t % TM]);
                __builtin_amdgcn_sched_barrier(0);
                // this wave's reads of stage s are back and its pieces of stage s+1 (the only one in flight) have
                // landed; after the barrier that holds for every wave
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!AF_DBG(1)) wait_vmcnt<0>();
                if (!AF_DBG(2)) __builtin_amdgcn_s_barrier();
                read_half(a0, b0, st1, 0);
                // ---- second half: MFMA(a1,b1) with the DMA pieces of stage s+2 in the gaps; sched_barrier pins the interleave
#pragma unroll
                for (int g = 0; g < PER_WAVE; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (refill && !AF_DBG(4)) issue_piece(st, g);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = g * MPG; t < (g + 1) * MPG && t < NTH; ++t) MMA::run(a1[t / TM], b1[t % TM], acc[t / TM][t % TM]);
                }
#pragma unroll
                for (int t = PER_WAVE * MPG; t < NTH; ++t) MMA::run(a1[t / TM], b1[t % TM], acc[t / TM][t % TM]);
                if (refill) advance();
                __builtin_amdgcn_sched_barrier(0);
                st = st1;
            }
        } else {
            wait_vmcnt<0>();                                          // prologue: stage 0 (and 1) landed
            __builtin_amdgcn_s_barrier();
            read_half(a0, b0, 0, 0);
            // all stages but the last (the last one is peeled: no barrier / look-ahead read in it, and - just as
            // important - no control-flow merge in the steady-state body, which would make hipcc's LDS wait counts
            // conservative and stall the second MFMA group on the look-ahead reads)
            for (int s = 0; s + 1 < S; ++s) {
                const bool refill = s + 2 < S;                        // slot (s+2)%3 == (s-1)%3: its last reads were
                const int nst = st == 0 ? 2 : st - 1;                 // retired before the previous mid-step barrier
                const int st1 = st == 2 ? 0 : st + 1;
                read_half(a1, b1, st, 1);
                // ---- first half: MFMA(a0,b0) with the DMA pieces of stage s+2 in the gaps
#pragma unroll
                for (int g = 0; g < PER_WAVE; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (refill && !AF_DBG(4)) issue_piece(nst, g);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = g * MPG; t < (g + 1) * MPG && t < NTH; ++t) MMA::run(a0[t / TM], b0[t % TM], acc[t / TM][t % TM]);
                }
#pragma unroll
                for (int t = PER_WAVE * MPG; t < NTH; ++t) MMA::run(a0[t / TM], b0[t % TM], acc[t / TM][t % TM]);
                if (refill) advance();
                __builtin_amdgcn_sched_barrier(0);
                // this wave's reads of stage s are back (they were issued a whole MFMA group ago) and its pieces of
                // stage s+1 have landed; after the barrier that holds for every wave
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!AF_DBG(1)) { if (refill) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>(); }
                if (!AF_DBG(2)) __builtin_amdgcn_s_barrier();
                read_half(a0, b0, st1, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- second half: MFMA(a1,b1) runs under the LDS reads just issued
#pragma unroll
                for (int t = 0; t < NTH; ++t) MMA::run(a1[t / TM], b1[t % TM], acc[t / TM][t % TM]);
                __builtin_amdgcn_sched_barrier(0);
                st = st1;
            }
        }
        read_half(a1, b1, st, 1);                                 // last stage
#pragma unroll
        for (int t = 0; t < NTH; ++t) MMA::run(a0[t / TM], b0[t % TM], acc[t / TM][t % TM]);
#pragma unroll
        for (int t = 0; t < NTH; ++t) MMA::run(a1[t / TM], b1[t % TM], acc[t / TM][t % TM]);
    }

    // ---- epilogue.  The MFMA accumulator holds 4 consecutive channels of one position per lane: fine for
    // the per-channel BN scale/shift, too narrow for HBM (8-byte pieces of a line).  Each wave therefore
    // transposes its sub-tile through a private fp32 LDS patch ([position][WTN + 4 pad] floats; the ring is
    // dead by now) and then streams whole rows: 16 bytes per lane, full 128-byte lines per row for the
    // output store AND the residual load.  Residual add and ReLU happen in fp32 before the one rounding.
    AF_STAMP(2);
    __builtin_amdgcn_s_barrier();                      // every wave has finished reading the ring
    int patch_off = 0;                                 // floats
    if (KS == 2) {
        // K-split partners: waves 4-7 park their partial sums in LDS (lane-linear, conflict-free), waves 0-3 add them
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        if (kgroup == 1) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) red[(wsub * TN * TM + i * TM + j) * 64 + lane] = acc[i][j];
        }
        __builtin_amdgcn_s_barrier();
        if (kgroup == 1) return;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] += red[(wsub * TN * TM + i * TM + j) * 64 + lane];
        patch_off = WN * WM * TN * TM * 64 * 4;        // patches live above the reduction buffer
    }
    // ---- 16-bit layers without a residual, a fused pool or a K split (every `a` / `b` conv and the projection blocks): BN + ReLU + the
    // one rounding happen in the accumulator layout and the transposition patch holds 16-bit values - half the LDS traffic of the
    // fp32 patch below, no conversions behind it (conv133g's epilogue: 7 k against 10.7 k cycles per 28-tile wave, in-kernel stamps).
    // The residual layers keep the fp32 patch: their sum is formed in fp32 on whole rows before the one rounding; so do the small
    // tiles that share a CU three or more at a time (<= 80 registers: no room for the BN parameters of a whole sub-tile).
    if constexpr (DT != AF_F32 && !SPLITK && KS == 1 && !LEAN) {
        if (a.tpool == 0 && !a.res) {
            typedef typename E::type OT;
            constexpr int PROW16 = WTN + 8;            // patch row stride in elements
            OT* patch16 = reinterpret_cast<OT*>(smem) + wsub * (16 * PROW16);
            constexpr int LPR16 = WTN / 8, RPI16 = 64 / LPR16;
            const int rr16 = lane / LPR16, cc16 = (lane % LPR16) * 8;
            const int chb = tile_n * BN + wn * WTN;
            f32x4 sc[TN], sf[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                sc[i] = *reinterpret_cast<const f32x4*>(a.scale + chb + i * 16 + fg * 4);
                sf[i] = *reinterpret_cast<const f32x4*>(a.shift + chb + i * 16 + fg * 4);
            }
            const float lo = a.relu ? 0.f : -__builtin_inff();     // max(v, lo): ReLU or nothing, NaN kept either way, no branch
#pragma unroll
            for (int j = 0; j < TM; ++j) {
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    f32x4 v = acc[i][j] * sc[i] + sf[i];
                    v[0] = max_nan(v[0], lo); v[1] = max_nan(v[1], lo); v[2] = max_nan(v[2], lo); v[3] = max_nan(v[3], lo);
                    Vec4<DT>::store(reinterpret_cast<char*>(patch16 + frow * PROW16 + i * 16 + fg * 4), v);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 16 / RPI16; ++it) {
                    const int row = it * RPI16 + rr16;
                    const long long m = m0 + wm * WTM + j * 16 + row;
                    const u32x4 o = *reinterpret_cast<const u32x4*>(patch16 + row * PROW16 + cc16);
                    if (m < a.M && (BMR == BM || wm * WTM + j * 16 + row < BMR) && chb + cc16 < a.Cout)
                        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.out + (m * a.out_ld + chb + cc16) * ES));
                }
                __builtin_amdgcn_wave_barrier();
            }
            AF_STAMP(3); AF_STAMP(7);
            AF_STAMP_FLUSH;
            return;
        }
    }
    constexpr int PROW = WTN + 4;                      // patch row stride in floats (pad: conflict-free b128 writes)
    constexpr int HALVES = TM % 4 == 0 && TM >= 8 ? 4 : TM % 2 == 0 ? 2 : TM;   // the patch holds a slice of the sub-tile at a time (LDS footprint)
    constexpr int TMH = TM / HALVES, PROWS = TMH * 16;
    float* patch = reinterpret_cast<float*>(smem) + patch_off + wsub * (PROWS * PROW);
    constexpr int LPR = WTN / EPC;                     // lanes per output row (16 bytes each)
    constexpr int RPI = 64 / LPR;                      // rows per wave-instruction
    const int rr = lane / LPR, cc = (lane % LPR) * EPC;
    const int ch0 = tile_n * BN + wn * WTN + cc;
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int chl = i * 16 + fg * 4;           // channel inside the wave's sub-tile
            const int ch = tile_n * BN + wn * WTN + chl;
            const f32x4 sc = SPLITK ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(a.scale + ch);   // partial sums leave raw
            const f32x4 sf = SPLITK ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.shift + ch);
#pragma unroll
            for (int j = 0; j < TMH; ++j)
                *reinterpret_cast<f32x4*>(patch + (j * 16 + frow) * PROW + chl) = acc[i][hf * TMH + j] * sc + sf;
        }
        __builtin_amdgcn_wave_barrier();               // same-wave LDS ops complete in order; keep the order
        if (a.tpool == 2) {
            // rows 4r .. 4r+3 = the 2x2 window of one pooled pixel: ReLU each, then the max -> pooled row (no residual)
            constexpr int QR = PROWS / 4;               // pooled rows in the patch (may be fewer than a wave-instruction covers)
#pragma unroll
            for (int it = 0; it < (QR + RPI - 1) / RPI; ++it) {
                const bool live = it * RPI + rr < QR;
                const int prow = live ? it * RPI + rr : 0;
                const long long m = m0 + wm * WTM + hf * PROWS + 4 * prow;      // first row of the window
                float v[EPC];
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4)
#pragma unroll
                    for (int e = 0; e < EPC; e += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(patch + (4 * prow + p4) * PROW + cc + e);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x = a.relu ? relu_f(t[q]) : t[q];
                            v[e + q] = p4 == 0 ? x : max_nan(v[e + q], x);   // NaN propagates like ATen's max_pool
                        }
                    }
                if (live && m < a.M && ch0 < a.Cout) {
                    uint4 o;
                    typename E::type* oe = reinterpret_cast<typename E::type*>(&o);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) oe[e] = E::from_f32(v[e]);
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + ((m >> 2) * a.out_ld + ch0) * ES));
                }
            }
        } else if (a.tpool) {
            // rows (2r, 2r+1) = frames (2j, 2j+1) of one pixel: residual add + ReLU each, then the max -> pooled row
#pragma unroll
            for (int it = 0; it < PROWS / (2 * RPI); ++it) {
                const int prow = it * RPI + rr;
                const long long m = m0 + wm * WTM + hf * PROWS + 2 * prow;      // even row of the pair
                float v[2][EPC];
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                    for (int e = 0; e < EPC; e += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(patch + (2 * prow + p2) * PROW + cc + e);
                        v[p2][e] = t[0]; v[p2][e + 1] = t[1]; v[p2][e + 2] = t[2]; v[p2][e + 3] = t[3];
                    }
                if (m < a.M && ch0 < a.Cout) {
                    if (a.res) {
                        const long long mq = m >> 1, hw = (long long)a.Ho * a.Wo;
                        const long long pix = mq % hw, nj = mq / hw;               // nj = n * (To/2) + j
                        const long long lin0 = (nj * 2) * hw + pix;                // (n, 2j, ho, wo) in NDHWC order
#pragma unroll
                        for (int p2 = 0; p2 < 2; ++p2) {
                            const uint4 rraw = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.res + ((lin0 + p2 * hw) * a.Cout + ch0) * ES)));
                            const typename E::type* re = reinterpret_cast<const typename E::type*>(&rraw);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) v[p2][e] += E::to_f32(re[e]);
                        }
                    }
                    uint4 o;
                    typename E::type* oe = reinterpret_cast<typename E::type*>(&o);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float x0 = v[0][e], x1 = v[1][e];
                        if (a.relu) { x0 = relu_f(x0); x1 = relu_f(x1); }
                        oe[e] = E::from_f32(max_nan(x0, x1));      // NaN propagates like ATen's max_pool
                    }
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + ((m >> 1) * a.out_ld + ch0) * ES));
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < PROWS / RPI; ++it) {
                const int row = it * RPI + rr;
                const long long m = m0 + wm * WTM + hf * PROWS + row;
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(patch + row * PROW + cc + e);
                    v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
                }
                if (SPLITK) {                           // fp32 partial sums of this K range -> workspace
                    if (m < a.M && ch0 < a.Cout) {
                        float* wp = a.ws + ((long long)blockIdx.y * a.M + m) * a.Cout + ch0;
#pragma unroll
                        for (int e = 0; e < EPC; e += 4)
                            *reinterpret_cast<f32x4*>(wp + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
                    }
                } else if (m < a.M && ch0 < a.Cout) {   // channel groups beyond Cout are padding
                    if (a.res) {
                        const uint4 rraw = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.res + (m * a.Cout + ch0) * ES)));
                        const typename E::type* re = reinterpret_cast<const typename E::type*>(&rraw);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[e] += E::to_f32(re[e]);
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[e] = relu_f(v[e]);
                    }
                    uint4 o;
                    typename E::type* oe = reinterpret_cast<typename E::type*>(&o);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) oe[e] = E::from_f32(v[e]);
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + (m * a.out_ld + ch0) * ES));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    AF_STAMP(3); AF_STAMP(7);
    AF_STAMP_FLUSH;
}

// (Round 4, late: PERSISTENT workgroups - a workgroup walking tiles blockIdx.x, blockIdx.x + gridDim.x, ... - were built on this body and
// dropped.  Inlined into a tile loop hipcc keeps the body's tile-invariant values alive across it: 106 scalar registers everywhere,
// 5-41 spilled vector registers in three tile shapes; as a noinline call the tile's arguments arrive in vector registers and the
// re-read argument block lands in scratch.  On the two shapes that compile clean (256 x 224 / 256 on 4 x 2 waves, 128 x 256) the layers
// ran 1-4 % SLOWER than one workgroup per tile: the 5 us per round they were meant to remove were not the dispatch but the 64-bit
// divisions of `row_offsets`, see there.)
template <int DT, int BN, int BM, int WN, int WM, int KS, int NSTAGE, int MINW, bool DUAL, bool SPLITK, int BMR>
__global__ __launch_bounds__(WN * WM * KS * 64, MINW) void conv_igemm_kernel(const ConvArgs a) {
    conv_igemm_tile<DT, BN, BM, WN, WM, KS, NSTAGE, MINW, DUAL, SPLITK, BMR>(a, blockIdx.x, gridDim.x);
}

template <int DT, int BN, int BM, int WN, int WM, int KS, int MINW, bool DUAL, int NSTAGE = 3, bool SPLITK = false, int BMR = BM>
static int launch(const ConvArgs& a, hipStream_t stream) {
    const long long tiles_m = (a.M + BMR - 1) / BMR;
    const long long blocks = tiles_m * a.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return set_error(AF_ERR_ARG, "conv: grid of %lld workgroups", blocks);
    // LDS actually needed: the ring slots this layer's K loop touches, or the epilogue patches
    constexpr int TMv = BMR / WM / 16;
    constexpr int red_bytes = KS == 2 ? WN * WM * (BN / WN / 16) * TMv * 64 * 16 : 0;
    constexpr int patch_bytes = red_bytes + WN * WM * ((TMv % 4 == 0 && TMv >= 8 ? TMv / 4 : TMv % 2 == 0 ? TMv / 2 : 1) * 16) * (BN / WN + 4) * 4;
    const int S = a.kt * a.kh * a.kw * a.kpt + a.kpt2;
    const int slots = (MINW >= 4 && a.ring == 2) ? 2 : NSTAGE;
    const int ring_bytes = (S < slots ? S : slots) * (BN + BM) * 128;
    const int lds = ring_bytes > patch_bytes ? ring_bytes : patch_bytes;
    AF_SET_MAX_LDS((&conv_igemm_kernel<DT, BN, BM, WN, WM, KS, NSTAGE, MINW, DUAL, SPLITK, BMR>),
                   NSTAGE * (BN + BM) * 128 > patch_bytes ? NSTAGE * (BN + BM) * 128 : patch_bytes, "conv");
    hipLaunchKernelGGL((conv_igemm_kernel<DT, BN, BM, WN, WM, KS, NSTAGE, MINW, DUAL, SPLITK, BMR>), dim3((unsigned)blocks, SPLITK ? a.ksplit : 1), dim3(WN * WM * KS * 64), lds, stream, a);
    AF_CHECK_LAUNCH("conv_igemm_kernel");
    return AF_OK;
}

// tile variant chosen for a layer (also reported to the caller: af_conv_variant).  Long-K layers are
// MFMA-bound: biggest tile.  Short-K layers (<= 3 K-steps) are HBM-bound streams of input, residual and
// output: half-height tiles with a trimmed ring so several workgroups share a CU and overlap each
// other's load / store phases.
enum { VAR_128x256 = 0, VAR_64x256 = 1, VAR_128x128 = 2, VAR_64x128 = 3, VAR_C133 = 4, VAR_128x128_R2 = 5, VAR_256x256 = 6,
       VAR_128x512 = 7, VAR_C311 = 8, VAR_SMALL = 9, VAR_C111 = 10, VAR_C133G = 11, VAR_256x224 = 12, VAR_C311G = 13, VAR_COUNT = 14 };
static const char* const kVariantNames[] = {"conv_igemm<BN=128,BM=256>", "conv_igemm<BN=64,BM=256>",
                                            "conv_igemm<BN=128,BM=128>", "conv_igemm<BN=64,BM=128>",
                                            "conv133_c64<weights in registers>", "conv_igemm<BN=128,BM=128>",
                                            "conv_igemm<BN=256,BM=256>", "conv_igemm<BN=128,BM=512>",
                                            "conv311_c64<time-tiled, taps share one LDS image>",
                                            "conv_small<direct-gather MFMA, narrow layers>",
                                            "conv111<persistent stream, weights in registers>",
                                            "conv133g<frame-resident halo patch, 9 taps share it>",
                                            "conv_igemm<BN=256,BM=224>",
                                            "conv311g<clip-resident (T + 2) x P patch, 3 taps share it>"};

static int pick_variant(int cout, int cin, int taps, int dtype, long long M, int cin2 = 0, int pooled = 0) {
    // AF_FORCE_VAR=<variant id>: tile sweeps of tools/exp_variants.py (experiments only; a tile whose channel count does not
    // divide the layer's is refused)
    if (const char* fv = getenv("AF_FORCE_VAR")) {
        const int v = atoi(fv);
        const int bn = (v == VAR_256x256 || v == VAR_256x224) ? 256 : (v == VAR_128x256 || v == VAR_128x512 || v == VAR_128x128 || v == VAR_128x128_R2) ? 128 : 64;
        const bool generic = v == VAR_128x256 || v == VAR_64x256 || v == VAR_128x128 || v == VAR_64x128 || v == VAR_128x128_R2 ||
                             v == VAR_256x256 || v == VAR_128x512 || v == VAR_256x224;
        if (generic && cout % bn == 0 && !(pooled && v == VAR_256x224)) return v;
    }
    const int ksteps = (taps * cin + cin2) / (dtype == AF_F32 ? 32 : 64);
    const bool wide = cout % 128 == 0, short_k = ksteps <= 3;
    // (with the 2x2 pool fused only a quarter of the output is written: such a layer is MFMA-bound from 4 K-steps on)
    if (cout % 256 == 0 && ksteps >= (pooled == 2 ? 4 : cin2 ? 6 : 9)) {
        // 256x256 tiles (128x64 per wave) move a third less L2 -> LDS traffic per MAC, which is what bounds the
        // 128x256 tile; a layer is as slow as its last round of workgroups, so compare whole rounds on the 256 CUs
        // (measured: one 256x256 tile takes 1.63x the time of a 128x256 tile).
        // 256x224 (64x112 per wave): the positions of the deep stages are multiples of 49 - 224-row tiles are 224 / 448 / 896
        // workgroups where 256-row tiles are 196 / 392 / 784 (77 % of the CUs in the last round); a 224-row tile costs ~0.89 of
        // a 256-row one.  AF_IGEMM_224 = 0 / 2 forces never / always (experiments).
        const char* e224 = getenv("AF_IGEMM_224");
        const int mode224 = e224 ? atoi(e224) : 1;
        const long long tm = (M + 255) / 256, big = tm * (cout / 256), v0 = tm * (cout / 128);
        const long long b224 = (M + 223) / 224 * (cout / 256);
        const double c256 = (double)((big + 255) / 256), c224 = 0.89 * (double)((b224 + 255) / 256);
        const bool take224 = pooled == 0 && mode224 != 0 && (mode224 == 2 || (M % 49 == 0 && c224 < c256 - 0.02));
        if (big >= 128 && (take224 ? c224 : c256) < 0.615 * (double)((v0 + 255) / 256) + 0.05) return take224 ? VAR_256x224 : VAR_256x256;
    }
    if (cout == 128 && !cin2 && taps >= 9) {
        // the same 128x64-per-wave layout turned on its side for 128-channel layers: 128x512 tiles (1x3x3 layers;
        // the 3x1x1 / 1x1x1 ones stream their activations from HBM and measured no faster)
        const long long big = (M + 511) / 512, v0 = (M + 255) / 256;
        if (big >= 128 && (double)((big + 255) / 256) < 0.615 * (double)((v0 + 255) / 256) + 0.05) return VAR_128x512;
    }
    // a handful of rows (the FTCN-TT head: 17 tokens per clip): the smallest tile, so that at least the channel
    // dimension spreads over the CUs
    if ((M + 255) / 256 * (cout / (wide ? 128 : 64)) < 64) return VAR_64x128;
    // wide-output streams of 4..8 K-steps, and the 512 -> 128 `a` convs of s3 (8 K-steps; measured 60 -> 50 us against the
    // 128x256 tile): two workgroups per CU cover each other's load / store phases
    if (wide && !short_k && ksteps <= 8 && (cout >= 512 || cout == 128)) return VAR_128x128_R2;
    return wide ? (short_k ? VAR_128x128 : VAR_128x256) : (short_k ? VAR_64x128 : VAR_64x256);
}

// split-K epilogue: out[m][c] = act((sum_y ws[y][m][c]) * scale[c] + shift[c] (+ res[m][c])); 4 channels per thread
template <int DT>
__global__ void splitk_finish_kernel(const float* ws, int ksplit, long long M, int Cout, const float* scale, const float* shift,
                                     const char* res, int relu, char* out, int out_ld) {
    constexpr int ES = 16 / Elem<DT>::EPC;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, per_row = Cout / 4;
    if (i >= M * per_row) return;
    const long long m = i / per_row;
    const int c = (int)(i % per_row) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int y = 0; y < ksplit; ++y) v += *reinterpret_cast<const f32x4*>(ws + ((long long)y * M + m) * Cout + c);
    v = v * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
    if (res) v += Vec4<DT>::load(res + (m * Cout + c) * ES);
    if (relu) { v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]); }
    Vec4<DT>::store(out + (m * out_ld + c) * ES, v);
}

// M-rows of a variant's tile
static int variant_bm(int v) { return v == VAR_128x512 ? 512 : v == VAR_256x224 ? 224 : (v == VAR_128x256 || v == VAR_64x256 || v == VAR_256x256) ? 256 : 128; }
static int variant_bn(int v) {
    return (v == VAR_256x256 || v == VAR_256x224) ? 256 : (v == VAR_128x256 || v == VAR_128x512 || v == VAR_128x128 || v == VAR_128x128_R2) ? 128 : 64;
}
// split K when the tiles alone leave most CUs idle (one clip, the deep stages): up to 8 K ranges of >= 8 K-steps each whose
// fp32 partial sums meet in the CALLER's workspace (ksplit * M * Cout floats); 1 = no split
static int plan_ksplit(int v, long long M, int coutp, int cout, int s_all, int tpool, bool dual) {
    const long long blocks = (M + variant_bm(v) - 1) / variant_bm(v) * (coutp / variant_bn(v));
    if (tpool || dual || blocks >= 128 || s_all < 16 || cout % 4 != 0) return 1;
    int ks = (int)(256 / blocks);
    if (ks > 8) ks = 8;
    if (ks > s_all / 8) ks = s_all / 8;
    return ks >= 2 ? ks : 1;
}

template <int DT>
static int dispatch(ConvArgs& a, hipStream_t stream) {
    constexpr int BK = 8 * Elem<DT>::EPC;
    a.kpt = (a.Cin + BK - 1) / BK;
    a.CinP = a.kpt * BK;
    a.kpt2 = a.in2 ? (a.Cin2 + BK - 1) / BK : 0;
    a.Cin2P = a.kpt2 * BK;
    a.CoutP = (a.Cout + 63) / 64 * 64;
    const int v = pick_variant(a.CoutP, a.CinP, a.kt * a.kh * a.kw, DT, a.M, a.in2 ? a.Cin2P : 0, a.tpool);
    a.tiles_n = a.CoutP / variant_bn(v);
    a.ring = v == VAR_128x128_R2 ? 2 : 3;
    // split K (small batches) only into a workspace the caller handed over and that is large enough
    a.ksplit = plan_ksplit(v, a.M, a.CoutP, a.Cout, a.kt * a.kh * a.kw * a.kpt + a.kpt2, a.tpool, a.in2 != nullptr);
    if (a.ksplit > 1 && (!a.ws || a.ws_bytes < (long long)a.ksplit * a.M * a.Cout * (long long)sizeof(float))) a.ksplit = 1;
    int rc;
    if (a.in2) {                                 // projection blocks (64-wide tiles: SlowFast's Fast pathway)
        switch (v) {
            case VAR_256x256: rc = launch<DT, 256, 256, 2, 4, 1, 2, true, 2>(a, stream); break;
            case VAR_256x224: rc = launch<DT, 256, 256, 4, 2, 1, 2, true, 2, false, 224>(a, stream); break;
            case VAR_128x256: rc = launch<DT, 128, 256, 2, 4, 1, 2, true>(a, stream); break;
            case VAR_64x256: rc = launch<DT, 64, 256, 1, 8, 1, 2, true>(a, stream); break;
            case VAR_128x128:
            case VAR_128x128_R2: rc = launch<DT, 128, 128, 2, 4, 1, 6, true>(a, stream); break;
            default: rc = launch<DT, 64, 128, 1, 8, 1, 4, true>(a, stream); break;
        }
    } else if (a.ksplit > 1) {
        switch (v) {
            case VAR_256x256: rc = launch<DT, 256, 256, 2, 4, 1, 2, false, 2, true>(a, stream); break;
            case VAR_256x224: rc = launch<DT, 256, 256, 4, 2, 1, 2, false, 2, true, 224>(a, stream); break;
            case VAR_128x512: rc = launch<DT, 128, 512, 2, 4, 1, 2, false, 2, true>(a, stream); break;
            case VAR_128x256: rc = launch<DT, 128, 256, 2, 4, 1, 2, false, 3, true>(a, stream); break;
            case VAR_64x256: rc = launch<DT, 64, 256, 1, 8, 1, 2, false, 3, true>(a, stream); break;
            case VAR_128x128:
            case VAR_128x128_R2: rc = launch<DT, 128, 128, 2, 4, 1, 6, false, 3, true>(a, stream); break;
            default: rc = launch<DT, 64, 128, 1, 8, 1, 4, false, 3, true>(a, stream); break;
        }
    } else {
        switch (v) {
            case VAR_256x256: rc = launch<DT, 256, 256, 2, 4, 1, 2, false, 2>(a, stream); break;
            case VAR_256x224: rc = launch<DT, 256, 256, 4, 2, 1, 2, false, 2, false, 224>(a, stream); break;
            case VAR_128x512: rc = launch<DT, 128, 512, 2, 4, 1, 2, false, 2>(a, stream); break;
            case VAR_128x256: rc = launch<DT, 128, 256, 2, 4, 1, 2, false>(a, stream); break;
            case VAR_64x256: rc = launch<DT, 64, 256, 1, 8, 1, 2, false>(a, stream); break;
            case VAR_128x128:
            case VAR_128x128_R2: rc = launch<DT, 128, 128, 2, 4, 1, 6, false>(a, stream); break;
            default: rc = launch<DT, 64, 128, 1, 8, 1, 4, false>(a, stream); break;
        }
    }
    if (rc != AF_OK || a.ksplit == 1) return rc;
    const long long quads = a.M * (a.Cout / 4);
    hipLaunchKernelGGL((splitk_finish_kernel<DT>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, a.ws, a.ksplit, a.M, a.Cout,
                       a.scale, a.shift, a.res, a.relu, a.out, a.out_ld);
    AF_CHECK_LAUNCH("splitk_finish_kernel");
    return AF_OK;
}

}  // namespace af

extern "C" int af_conv_variant(const af_conv_desc* d, const af_conv_desc* d2) {
    AF_REQUIRE(d && d->cout > 0 && d->cin > 0 && af::dtype_ok(d->dtype), "conv_variant: bad descriptor");
    if (af::conv_small_applies(d, d2, nullptr, 0)) return af::VAR_SMALL;
    if (!d2 && af::conv133_applies(d, nullptr, 0)) return af::VAR_C133;
    if (!d2 && af::conv133g_applies(d, nullptr, 0)) return af::VAR_C133G;
    if (!d2 && af::conv311g_applies(d, nullptr, 0)) return af::VAR_C311G;
    if (!d2 && af::conv311_applies(d, nullptr, 0)) return af::VAR_C311;
    if (af::conv111_applies(d, d2, nullptr, 0)) return af::VAR_C111;
    const int bk = d->dtype == AF_F32 ? 32 : 64;
    return af::pick_variant((d->cout + 63) / 64 * 64, (d->cin + bk - 1) / bk * bk, d->kt * d->kh * d->kw, d->dtype,
                            (long long)d->n * d->to * d->ho * d->wo, d2 ? (d2->cin + bk - 1) / bk * bk : 0, d->tpool);
}

extern "C" const char* af_conv_variant_name(int variant) {
    return (variant >= 0 && variant < af::VAR_COUNT) ? af::kVariantNames[variant] : "?";
}

extern "C" int64_t af_conv_workspace_bytes(const af_conv_desc* d) {
    using namespace af;
    if (!d || d->cout <= 0 || d->cin <= 0 || !dtype_ok(d->dtype) || d->n <= 0 || d->to <= 0 || d->ho <= 0 || d->wo <= 0) return 0;
    const int bk = d->dtype == AF_F32 ? 32 : 64;
    const int coutp = (d->cout + 63) / 64 * 64, cinp = (d->cin + bk - 1) / bk * bk, taps = d->kt * d->kh * d->kw;
    const long long M = (long long)d->n * d->to * d->ho * d->wo;
    const int v = pick_variant(coutp, cinp, taps, d->dtype, M, 0, d->tpool);
    const int ks = plan_ksplit(v, M, coutp, d->cout, taps * (cinp / bk), d->tpool, false);
    return ks > 1 ? (int64_t)ks * M * d->cout * (int64_t)sizeof(float) : 0;
}

static int conv_common(const af_conv_desc* d, const void* in, const void* w_packed, const af_conv_desc* d2,
                       const void* in2, const void* w2_packed, const float* scale, const float* shift,
                       const void* residual, void* out, int out_ld, void* workspace, int64_t workspace_bytes, void* stream) {
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
    const int bk = d->dtype == AF_F32 ? 32 : 64, epc = d->dtype == AF_F32 ? 4 : 8;
    AF_REQUIRE(d->cin % epc == 0, "conv: cin=%d must be a multiple of %d for this dtype", d->cin, epc);
    AF_REQUIRE(d->cout % epc == 0, "conv: cout=%d must be a multiple of %d for this dtype", d->cout, epc);
    if (out_ld == 0) out_ld = d->cout;
    AF_REQUIRE(out_ld >= d->cout && out_ld % 8 == 0, "conv: bad out_ld %d", out_ld);
    AF_REQUIRE(aligned16(in) && aligned16(w_packed) && aligned16(scale) && aligned16(shift) && aligned16(out) &&
                   aligned16(residual), "conv: buffers must be 16-byte aligned");
    AF_REQUIRE((long long)(d->cout + 63) * d->kt * d->kh * d->kw * (d->cin + bk) < (1LL << 31), "conv: weight too large");
    AF_REQUIRE(d->kt * d->kh * d->kw <= 31, "conv: at most 31 kernel taps (got %d)", d->kt * d->kh * d->kw);
    AF_REQUIRE((long long)d->kt * d->h * d->w * d->cin * dtype_size(d->dtype) < (1LL << 31), "conv: tap offset overflows");
    // buffer addressing: a tile's rows are reached by 32-bit offsets from its first row, which needs input offsets
    // that grow with the output position (padding at most half the kernel) and a tile that spans less than 2 GiB
    AF_REQUIRE(2 * d->pt <= d->kt && 2 * d->ph <= d->kh && 2 * d->pw <= d->kw, "conv: padding larger than half the kernel");
    AF_REQUIRE((256 / ((long long)to * ho * wo) + 3) * d->t * d->h * d->w * d->cin * dtype_size(d->dtype) < (1LL << 31),
               "conv: one clip of input is too large for 32-bit tile offsets");

    const bool small = conv_small_applies(d, d2, residual, out_ld);
    if (!small && !d2 && conv133_applies(d, residual, out_ld))
        return conv133_run(d, in, w_packed, scale, shift, out, (hipStream_t)stream);
    if (!small && !d2 && conv133g_applies(d, residual, out_ld))
        return conv133g_run(d, in, w_packed, scale, shift, out, out_ld, (hipStream_t)stream);
    if (!small && !d2 && conv311g_applies(d, residual, out_ld))
        return conv311g_run(d, in, w_packed, scale, shift, out, out_ld, (hipStream_t)stream);
    if (!small && !d2 && conv311_applies(d, residual, out_ld))
        return conv311_run(d, in, w_packed, scale, shift, out, out_ld, (hipStream_t)stream);

    ConvArgs a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift;
    a.res = (const char*)residual; a.out = (char*)out;
    a.T = d->t; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout;
    a.kt = d->kt; a.kh = d->kh; a.kw = d->kw; a.st = d->st; a.sh = d->sh; a.sw = d->sw;
    a.pt = d->pt; a.ph = d->ph; a.pw = d->pw; a.To = to; a.Ho = ho; a.Wo = wo;
    a.relu = d->relu; a.out_ld = out_ld;
    AF_REQUIRE(aligned16(workspace) && workspace_bytes >= 0, "conv: the workspace must be 16-byte aligned");
    a.ws = (float*)workspace; a.ws_bytes = workspace ? workspace_bytes : 0;
#ifdef AF_STAMPS
    { const char* ep = getenv("AF_STAMP_PTR"); a.stamps = ep ? (unsigned long long*)strtoull(ep, nullptr, 0) : nullptr;
      const char* ed = getenv("AF_G_DBG"); a.dbg = ed ? atoi(ed) : 0; }
#endif
    AF_REQUIRE(d->tpool >= 0 && d->tpool <= 2, "conv: tpool must be 0, 1 (temporal pairs) or 2 (2x2 pixels)");
    a.tpool = d->tpool;
    AF_REQUIRE(a.tpool != 1 || (to % 2 == 0), "conv: fused temporal pool needs an even number of output frames (%d)", to);
    AF_REQUIRE(a.tpool != 2 || (ho % 2 == 0 && wo % 2 == 0 && !residual && !d2),
               "conv: the fused 2x2 pool needs even output height / width (%d x %d), no residual and no second segment", ho, wo);
    a.M = (long long)d->n * to * ho * wo;
    AF_REQUIRE(a.M < (1LL << 31), "conv: %lld output positions (the row index is 32-bit)", a.M);
    {
        auto magic = [](unsigned dv, unsigned& mm, unsigned& l) {
            l = 0; while ((1ull << l) < dv) ++l;
            mm = (unsigned)(((1ull << 32) * ((1ull << l) - dv)) / dv + 1);
        };
        magic((unsigned)(a.tpool == 2 ? (wo >> 1) : wo), a.div_w_m, a.div_w_l);
        magic((unsigned)(a.tpool == 2 ? (ho >> 1) : ho), a.div_h_m, a.div_h_l);
        magic((unsigned)(a.tpool == 1 ? (to >> 1) : to), a.div_t_m, a.div_t_l);
    }
    a.in2 = nullptr; a.w2 = (const char*)w_packed; a.T2 = a.H2 = a.W2 = 1; a.Cin2 = a.Cin2P = bk; a.st2 = a.sh2 = a.sw2 = 1; a.kpt2 = 0;
    if (d2) {
        AF_REQUIRE(in2 && w2_packed && aligned16(in2) && aligned16(w2_packed), "conv: second segment needs in2 / w2");
        AF_REQUIRE(d2->dtype == d->dtype && d2->n == d->n && d2->cout == d->cout, "conv: second segment dtype/n/cout mismatch");
        AF_REQUIRE(d2->kt == 1 && d2->kh == 1 && d2->kw == 1 && d2->pt == 0 && d2->ph == 0 && d2->pw == 0,
                   "conv: second segment must be a 1x1x1 convolution without padding");
        AF_REQUIRE(d2->st > 0 && d2->sh > 0 && d2->sw > 0 && (d2->t - 1) / d2->st + 1 == to && (d2->h - 1) / d2->sh + 1 == ho &&
                       (d2->w - 1) / d2->sw + 1 == wo, "conv: second segment does not land on the same output positions");
        AF_REQUIRE(d2->cin > 0 && d2->cin % epc == 0, "conv: second segment cin=%d must be a multiple of %d", d2->cin, epc);
        AF_REQUIRE((256 / ((long long)to * ho * wo) + 3) * d2->t * d2->h * d2->w * d2->cin * dtype_size(d->dtype) < (1LL << 31),
                   "conv: one clip of the second input is too large for 32-bit tile offsets");
        a.in2 = (const char*)in2; a.w2 = (const char*)w2_packed;
        a.T2 = d2->t; a.H2 = d2->h; a.W2 = d2->w; a.Cin2 = d2->cin; a.st2 = d2->st; a.sh2 = d2->sh; a.sw2 = d2->sw;
    }
    if (small)
        return conv_small_run(d, in, w_packed, d2, in2, w2_packed, scale, shift, residual, out, out_ld, (hipStream_t)stream);
    if (conv111_applies(d, d2, residual, out_ld))
        return conv111_run(d, in, w_packed, d2, in2, w2_packed, scale, shift, residual, out, out_ld, (hipStream_t)stream);
    hipStream_t s = (hipStream_t)stream;
    switch (d->dtype) {
        case AF_F32: return dispatch<AF_F32>(a, s);
        case AF_BF16: return dispatch<AF_BF16>(a, s);
        default: return dispatch<AF_F16>(a, s);
    }
}

extern "C" int af_conv3d_bn_act(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale,
                                const float* shift, const void* residual, void* out, int out_ld, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    return conv_common(d, in, w_packed, nullptr, nullptr, nullptr, scale, shift, residual, out, out_ld, workspace, workspace_bytes, stream);
}

extern "C" int af_conv_ca_fusable(const af_conv_desc* dc, const af_conv_desc* d1, const af_conv_desc* da) {
    return af::conv_ca_applies(dc, d1, da) ? 1 : 0;
}

extern "C" int af_conv3d_ca_bn_act(const af_conv_desc* dc, const void* in_b, const void* wc_packed, const af_conv_desc* d1,
                                   const void* in1, const void* w1_packed, const float* scale_c, const float* shift_c,
                                   const void* residual, void* out_x, const af_conv_desc* da, const void* wa_packed,
                                   const float* scale_a, const float* shift_a, void* out_a, void* stream) {
    using namespace af;
    AF_REQUIRE(dc && da && in_b && wc_packed && scale_c && shift_c && out_x && wa_packed && scale_a && shift_a && out_a, "conv_ca: null argument");
    AF_REQUIRE((d1 != nullptr) == (in1 != nullptr) && (d1 != nullptr) == (w1_packed != nullptr) && (d1 != nullptr) != (residual != nullptr),
               "conv_ca: either a residual (plain block) or the projection shortcut (d1, in1, w1: block 0), not both");
    AF_REQUIRE(aligned16(in_b) && aligned16(wc_packed) && aligned16(in1) && aligned16(w1_packed) && aligned16(scale_c) && aligned16(shift_c) &&
                   aligned16(residual) && aligned16(out_x) && aligned16(wa_packed) && aligned16(scale_a) && aligned16(shift_a) && aligned16(out_a),
               "conv_ca: buffers must be 16-byte aligned");
    AF_REQUIRE(dc->to == dc->t && dc->ho == dc->h && dc->wo == dc->w && da->to == da->t && da->ho == da->h && da->wo == da->w,
               "conv_ca: both convolutions keep the tensor size");
    AF_REQUIRE(conv_ca_applies(dc, d1, da), "conv_ca: this (1x1x1 c, 3x1x1 a) pair does not take the fused path (ask af_conv_ca_fusable first)");
    return conv_ca_run(dc, in_b, wc_packed, in1, w1_packed, scale_c, shift_c, residual, out_x, da, wa_packed, scale_a, shift_a, out_a,
                       (hipStream_t)stream);
}

extern "C" int af_conv_bc_fusable(const af_conv_desc* db, const af_conv_desc* dc) {
    return af::conv133g_fused_applies(db, dc, 0) ? 1 : 0;
}

extern "C" int af_conv3d_bc_bn_act(const af_conv_desc* db, const void* in, const void* wb_packed, const float* scale_b,
                                   const float* shift_b, const af_conv_desc* dc, const void* wc_packed, const float* scale_c,
                                   const float* shift_c, const void* residual, void* out, int out_ld, void* stream) {
    using namespace af;
    AF_REQUIRE(db && dc && in && wb_packed && scale_b && shift_b && wc_packed && scale_c && shift_c && out, "conv_bc: null argument");
    AF_REQUIRE(aligned16(in) && aligned16(wb_packed) && aligned16(wc_packed) && aligned16(scale_b) && aligned16(shift_b) &&
                   aligned16(scale_c) && aligned16(shift_c) && aligned16(residual) && aligned16(out), "conv_bc: buffers must be 16-byte aligned");
    AF_REQUIRE(db->to == db->t && db->ho == db->h && db->wo == db->w && dc->to == dc->t && dc->ho == dc->h && dc->wo == dc->w,
               "conv_bc: both convolutions keep the spatial size");
    AF_REQUIRE(conv133g_fused_applies(db, dc, out_ld),
               "conv_bc: this (1x3x3, 1x1x1) pair does not take the fused path (ask af_conv_bc_fusable first)");
    return conv133g_fused_run(db, in, wb_packed, scale_b, shift_b, dc, wc_packed, scale_c, shift_c, residual, out, out_ld,
                              (hipStream_t)stream);
}

extern "C" int af_conv3d_dual_bn_act(const af_conv_desc* d, const void* in, const void* w_packed,
                                     const af_conv_desc* d2, const void* in2, const void* w2_packed, const float* scale,
                                     const float* shift, void* out, int out_ld, void* stream) {
    AF_REQUIRE(d2, "conv_dual: null second descriptor");
    return conv_common(d, in, w_packed, d2, in2, w2_packed, scale, shift, nullptr, out, out_ld, nullptr, 0, stream);
}

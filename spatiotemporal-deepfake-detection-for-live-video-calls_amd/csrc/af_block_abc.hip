// A whole NARROW bottleneck block as one launch (round 3): SlowFast's Fast pathway (reference
// altfreezing/slowfast/models/resnet_helper.py:255-326 BottleneckTransform + :411-444 ResBlock, built with
// dim_inner = 8 / 16 by video_model_builder.py:146-387), identity shortcut:
//     y = relu(x + bn_c(Wc . relu(bn_b(Wb (*)1x3x3 relu(bn_a(Wa (*)kTx1x1 x))))))
// with inner width c = 8, 16 or 32 and trunk width C = 4 c (32 / 64 / 128 channels), 16-bit operands.
//
// As three launches these blocks were latency- and launch-bound, not bandwidth-bound: the tensors between a, b and c are 16 / 32
// bytes per position and every kernel re-gathers them lane by lane (s2 of the Fast pathway: 0.27 ms per block against 0.03 ms
// for reading and writing the trunk once).  Here a workgroup owns a 14-column x PH-row patch of one clip for a run of frames
// and walks time:
//   * the trunk patch of a frame, one halo pixel around it, enters LDS by LDS-DMA (out-of-image lanes = out-of-range offsets =
//     zeros) into a ring of four frame slots: frames t - 1, t, t + 1 feed the temporal taps of a(t), frame t + 2 is in flight;
//   * a(t) on the haloed patch (a row of 16 pixels = one MFMA position tile; outside the image it is forced to 0 - the spatial
//     zero padding of b applies to a's OUTPUT), b(t) and c(t) are three small MFMA phases whose operands never leave the CU:
//     a -> 16 / 32 bytes per pixel in LDS, the nine taps of b are constant pixel shifts in that image, c reads b's pixels and
//     adds the residual from the ring's frame t;
//   * y(t) crosses an LDS tile and leaves as whole pixel rows (16 bytes per lane) - during the NEXT frame's a phase, and by
//     waves 4-7 only, while all LDS-DMA is issued by waves 0-3: the top-of-step vmcnt(0) that makes the next frame visible then
//     waits for loads only (with one wave doing both it also waited ~1.5 us per frame for the store acknowledgements).
// HBM traffic = the trunk in (x 1.3 - 1.5 for the halo) and out; the matrix work is a few dozen MFMAs per wave and frame.
// PROJ form (block 0 of the Fast pathway's s2: 8 -> 32 channels, stride 1): the input has CIN channels instead of C, the shortcut
// is the 1x1x1 projection `branch1` (resnet_helper.py:411-436) accumulated into c's tile like af_conv3d_dual_bn_act does - both
// weight sets carry their BN scale (folded in fp32 by the caller), shift = shift_c + shift_1 - and no residual is added.
#include "af_common.h"
#include <stdlib.h>

namespace af {

struct ABCArgs {
    const char* x;            // [N][T][H][W][C]
    const char *wa, *wb, *wc; // packed [64][taps][64] (af_pack_conv_weight)
    const char* w1;           // PROJ: the projection shortcut's packed weight (BN scale folded), else null
    const float *sa, *ha, *sb, *hb, *sc, *hc;   // BN scale / shift of a, b, c
    char* y;                  // [N][T][H][W][out_ld]
    int T, H, W, kta;
    int PH;                   // patch rows (haloed patch: PH + 2 rows x 16 columns)
    int py_n, px_n, ts_n, TS; // patches per frame (rows, columns), time segments per clip, frames per segment
    int out_ld;
};

constexpr int kAbcRing = 4;         // frames t - 1 .. t + 1 in use, t + 2 in flight (six slots + a two-step prefetch measured 8 % slower)

template <int INNER, int CIN> struct AbcDims {
    static constexpr int C = 4 * INNER;                // trunk channels out (= CIN for an identity-shortcut block)
    static constexpr bool PROJ = CIN != C;
    static constexpr int PB = C * 2, AB = INNER * 2;   // bytes per output / inner pixel
    static constexpr int PBI = CIN * 2;                // bytes per input pixel
    static constexpr int SPTA = CIN / 8;               // 16-byte K-slots per temporal tap of a (and K-slots of the projection)
    static constexpr int SPT = INNER / 8;              // 16-byte K-slots per tap of b (and K-slots of c)
    static constexpr int NBB = (9 * SPT + 3) / 4;      // K-blocks of b
    static constexpr int NB1 = PROJ ? (SPTA + 3) / 4 : 0;   // K-blocks of the projection shortcut
    static constexpr int NTA = (INNER + 15) / 16;      // channel tiles of a and b
    static constexpr int NTC = C / 16;                 // channel tiles of c
    static constexpr int CPP = PB / 16;                // 16-byte chunks per output pixel
    static constexpr int CPPI = PBI / 16;              // ... per input pixel
};

static inline int abc_lds_bytes(int inner, int cin, int ph, int kta) {
    const int C = 4 * inner, PB = C * 2, PBI = cin * 2, AB = inner * 2, rows = ph + 2;
    const int nta = (inner + 15) / 16;
    const int nfrag = ((kta * (cin / 8) + 3) / 4 + (9 * (inner / 8) + 3) / 4) * nta + C / 16 * (1 + (cin != C ? (cin / 8 + 3) / 4 : 0));
    return kAbcRing * ((rows * 16 * PBI + 1023) & ~1023) + (rows * 16 + 34) * AB + rows * 16 * AB + ph * 16 * PB + nfrag * 1024 + (2 * 32 + 2 * 32 + 2 * C) * 4;
}

template <int DT, int INNER, int CIN>
__global__ __launch_bounds__(512, 2) void block_abc_kernel(const ABCArgs a) {
    typedef Elem<DT> E;
    typedef AbcDims<INNER, CIN> D;
    static_assert(E::EPC == 8 && (INNER == 8 || INNER == 16 || INNER == 32), "16-bit operands, inner width 8, 16 or 32");
    constexpr int C = D::C, PB = D::PB, PBI = D::PBI, AB = D::AB, SPTA = D::SPTA, SPT = D::SPT, NBB = D::NBB, NB1 = D::NB1, NTA = D::NTA, NTC = D::NTC,
                  CPP = D::CPP, CPPI = D::CPPI;
    constexpr bool PROJ = D::PROJ;
    constexpr int CINP = CIN > 64 ? (CIN + 63) / 64 * 64 : 64;       // packed row pitch per tap of a / the projection

    extern __shared__ uint4 smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int ROWS = a.PH + 2;
    const int slot_bytes = (ROWS * 16 * PBI + 1023) & ~1023;         // whole DMA pieces: a piece's lanes beyond the patch write zeros
    char* abuf = sm + kAbcRing * slot_bytes;                       // a(t): haloed pixel p at entry 17 + p (guards: taps of dropped columns)
    char* bbuf = abuf + (ROWS * 16 + 34) * AB;                     // b(t)
    char* ybuf = bbuf + ROWS * 16 * AB;                            // y(t): rows 1 .. PH of the haloed patch
    uint4* wfr = reinterpret_cast<uint4*>(ybuf + a.PH * 16 * PB);  // A fragments: a, b, c (, projection); 64 lanes x 16 B each
    // K of a as 16-byte slots: slot s = (tap dt = s / SPTA, 8-channel part s % SPTA) of kta * SPTA; four slots per K-block
    const int nsa = a.kta * SPTA, nba = (nsa + 3) >> 2;
    const int nfa = nba * NTA, nfb = NBB * NTA, nf1 = NB1 * NTC;   // fragment index: (K-block, channel tile)
    float* bn = reinterpret_cast<float*>(wfr + (nfa + nfb + NTC + nf1) * 64);   // a: scale[32] shift[32]; b: the same; c: scale[C] shift[C]

    // ---- weights -> LDS in fragment order (lane = (output channel row frow, K-group fg)); BN parameters
    for (int idx = tid; idx < (nfa + nfb + NTC + nf1) * 64; idx += 512) {
        const int f = idx >> 6, l = idx & 63, g = l >> 4;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (f < nfa) {
            const int blk = f / NTA, o = (f - blk * NTA) * 16 + (l & 15), sl = 4 * blk + g, dt = sl / SPTA, part = sl - dt * SPTA;
            if (sl < nsa) v = *reinterpret_cast<const uint4*>(a.wa + (((long long)o * a.kta + dt) * CINP + part * 8) * 2);
        } else if (f < nfa + nfb) {
            const int blk = (f - nfa) / NTA, o = ((f - nfa) - blk * NTA) * 16 + (l & 15);
            const int sl = 4 * blk + g, tap = sl / SPT, part = sl - tap * SPT;
            if (tap < 9) v = *reinterpret_cast<const uint4*>(a.wb + (((long long)o * 9 + tap) * 64 + part * 8) * 2);
        } else if (f < nfa + nfb + NTC) {
            const int nt = f - nfa - nfb;
            v = *reinterpret_cast<const uint4*>(a.wc + ((long long)(nt * 16 + (l & 15)) * 64 + g * 8) * 2);   // columns >= INNER are zero
        } else {
            const int q1 = f - nfa - nfb - NTC, blk = q1 / NTC, nt = q1 - blk * NTC, sl = 4 * blk + g;
            if (sl < SPTA) v = *reinterpret_cast<const uint4*>(a.w1 + ((long long)(nt * 16 + (l & 15)) * CINP + sl * 8) * 2);
        }
        wfr[idx] = v;
    }
    if (tid < 32) {
        bn[tid] = a.sa[tid]; bn[32 + tid] = a.ha[tid]; bn[64 + tid] = a.sb[tid]; bn[96 + tid] = a.hb[tid];
    }
    if (tid < C) { bn[128 + tid] = a.sc[tid]; bn[128 + C + tid] = a.hc[tid]; }

    // ---- work unit: (clip, patch row, patch column, time segment)
    int q = blockIdx.x;
    const int ts = q % a.ts_n; q /= a.ts_n;
    const int px = q % a.px_n; q /= a.px_n;
    const int py = q % a.py_n; const int n = q / a.py_n;
    const int t0 = ts * a.TS, t1 = (t0 + a.TS < a.T) ? t0 + a.TS : a.T;
    const int h0 = py * a.PH, w0 = px * 14;
    const int pt = a.kta >> 1;
    const long long frame_bytes = (long long)a.H * a.W * PBI;

    // trunk patch producer (waves 0-3): a frame's haloed patch is ROWS * 16 * CPPI 16-byte chunks, lane-linear in LDS; piece g is
    // chunks 64 g .. 64 g + 63 (chunk -> pixel chunk / CPPI = (row, column), part chunk % CPPI); a producer wave issues pieces
    // wave, wave + 4, ... (<= 6 each: <= 24 pieces per frame, host-checked)
    constexpr int MAXP = 6;
    const int npieces = (ROWS * 16 * CPPI + 63) >> 6;
    unsigned xoff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int ck = (wave + 4 * i) * 64 + lane, pix = ck / CPPI, part = ck - pix * CPPI, r = pix >> 4, pxl = pix & 15;
        const int hh = h0 - 1 + r, ww = w0 - 1 + pxl;
        const bool ok = wave < 4 && r < ROWS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        xoff[i] = ok ? (unsigned)(((hh * a.W + ww) * PBI) + part * 16) : kOutOfRange;
    }
    auto issue_frame = [&](int f) {                                    // uniform f in [0, T); producer waves only
        if (wave >= 4) return;
        const i32x4 desc = make_desc(a.x + ((long long)n * a.T + f) * frame_bytes);
        const unsigned base = lds0 + ((f + kAbcRing) & (kAbcRing - 1)) * slot_bytes;
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
            if (wave + 4 * i < npieces) blds16(xoff[i], desc, 0, __builtin_amdgcn_readfirstlane(base + (wave + 4 * i) * 1024));
    };
    // y(t) leaves as whole pixels, 16 bytes per lane: patch row rr, pixel pxl (haloed column pxl + 1); store waves 4-7 only
    auto store_frame = [&](int t) {
        if (wave < 4) return;
        char* yt = a.y + (((long long)n * a.T + t) * a.H) * a.W * a.out_ld * 2;
        for (int idx = tid - 256; idx < a.PH * 14 * CPP; idx += 256) {
            const int rr = idx / (14 * CPP), rem = idx - rr * (14 * CPP), pxl = rem / CPP, part = rem - pxl * CPP;
            if (h0 + rr < a.H && w0 + pxl < a.W) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(ybuf + (rr * 16 + pxl + 1) * PB + part * 16);
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(yt + ((long long)(h0 + rr) * a.W + w0 + pxl) * a.out_ld * 2 + part * 16));
            }
        }
    };
    // frames the segment needs: t0 - pt .. t1 - 1 + pt, clipped to the clip (a frame outside it is a skipped tap)
    const int f_lo = t0 - pt > 0 ? t0 - pt : 0, f_hi = (t1 - 1 + pt < a.T - 1) ? t1 - 1 + pt : a.T - 1;
    for (int f = f_lo; f <= f_hi && f <= t0 + 1; ++f) issue_frame(f);

    for (int t = t0; t < t1; ++t) {
        if (wave < 4) wait_vmcnt<0>();                                 // frame t + 1 has landed (these waves never store)
        __syncthreads();                                               // ... for everybody; y(t - 1) is complete in its tile
        if (t + 2 <= f_hi) issue_frame(t + 2);                         // its slot held frame t - 2: dead since a(t - 1)
        if (t > t0) store_frame(t - 1);                                // under this frame's a phase

        // (Handing a wave its tiles in pairs with two interleaved accumulators was tried and measured 5 - 20 % slower: the clamped
        //  second tile of an odd count is redundant work and the phases are short enough that wave-level parallelism wins.)
        // ---- a(t): kT x 1 x 1 over the haloed patch, one row of 16 pixels per MFMA tile
        for (int it = wave; it < ROWS * NTA; it += 8) {
            const int r = it / NTA, nt = it - r * NTA;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (SPTA >= 4) {
                // whole K-blocks per tap: the tap's frame is uniform.  All fragments are read first (a frame outside the clip or a
                // tap beyond kT reads frame t and is zeroed: no branch, ONE LDS round trip for the tile instead of one per tap),
                // then the MFMAs run back to back
                constexpr int KBA = SPTA / 4 > 0 ? SPTA / 4 : 1;
                uint4 bx[3][KBA];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const int f = t + dt - pt;
                    const bool ok = dt < a.kta && f >= 0 && f < a.T;
                    const char* xs = sm + (((ok ? f : t) + kAbcRing) & (kAbcRing - 1)) * slot_bytes + (r * 16 + frow) * PBI + fg * 16;
#pragma unroll
                    for (int kb = 0; kb < KBA; ++kb) {
                        bx[dt][kb] = *reinterpret_cast<const uint4*>(xs + kb * 64);
                        if (!ok) bx[dt][kb] = uint4{0u, 0u, 0u, 0u};
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int kb = 0; kb < KBA; ++kb)
                        Mma<DT>::run(wfr[(((dt < a.kta ? dt : 0) * KBA + kb) * NTA + nt) * 64 + lane], bx[dt][kb], acc);
            } else {
                // narrow input (projection form, 8 channels): a K-block's four 16-byte slots are different taps - each lane group
                // reads its own frame of the ring, or nothing
                for (int blk = 0; blk < nba; ++blk) {
                    const int sl = 4 * blk + fg, dt = sl / SPTA, part = sl - dt * SPTA, f = t + dt - pt;
                    uint4 b = uint4{0u, 0u, 0u, 0u};
                    if (sl < nsa && f >= 0 && f < a.T)
                        b = *reinterpret_cast<const uint4*>(sm + ((f + kAbcRing) & (kAbcRing - 1)) * slot_bytes + (r * 16 + frow) * PBI + part * 16);
                    Mma<DT>::run(wfr[(blk * NTA + nt) * 64 + lane], b, acc);
                }
            }
            const int hh = h0 - 1 + r, ww = w0 - 1 + frow, ch = nt * 16 + fg * 4;
            const bool inside = (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
            if (ch < INNER) {
                const f32x4 sv = *reinterpret_cast<const f32x4*>(bn + ch), hv = *reinterpret_cast<const f32x4*>(bn + 32 + ch);
                f32x4 v = acc * sv + hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = inside ? relu_f(v[e]) : 0.f;
                Vec4<DT>::store(abuf + (17 + r * 16 + frow) * AB + ch * 2, v);
            }
        }
        __syncthreads();
        // ---- b(t): 1 x 3 x 3 on rows 1 .. PH; K-slot s = 4 blk + fg = (tap, 8-channel part); a tap is a constant pixel shift
        for (int it = wave; it < a.PH * NTA; it += 8) {
            const int r = 1 + it / NTA, nt = it % NTA;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            uint4 bb[NBB], wb[NBB];                                      // all fragments first: one LDS round trip per tile
#pragma unroll
            for (int blk = 0; blk < NBB; ++blk) {
                const int sl = 4 * blk + fg, tap = sl < 9 * SPT ? sl / SPT : 4, part = sl < 9 * SPT ? sl - tap * SPT : 0;
                const int dh = tap / 3, dw = tap - dh * 3;
                bb[blk] = *reinterpret_cast<const uint4*>(abuf + (17 + (r + dh - 1) * 16 + frow + dw - 1) * AB + part * 16);
                if (sl >= 9 * SPT) bb[blk] = uint4{0u, 0u, 0u, 0u};
                wb[blk] = wfr[(nfa + blk * NTA + nt) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int blk = 0; blk < NBB; ++blk) Mma<DT>::run(wb[blk], bb[blk], acc);
            const int ch = nt * 16 + fg * 4;
            if (ch < INNER) {
                const f32x4 sv = *reinterpret_cast<const f32x4*>(bn + 64 + ch), hv = *reinterpret_cast<const f32x4*>(bn + 96 + ch);
                f32x4 v = acc * sv + hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
                Vec4<DT>::store(bbuf + (r * 16 + frow) * AB + ch * 2, v);
            }
        }
        __syncthreads();
        // ---- c(t): 1 x 1 x 1 back to C channels, + the shortcut (the ring's frame t: its own channels, or the projection as a
        // second K segment), ReLU -> the output tile (every wave is past its store of y(t - 1): two barriers ago)
        {
            const char* xt = sm + ((t + kAbcRing) & (kAbcRing - 1)) * slot_bytes;
            for (int it = wave; it < a.PH * NTC; it += 8) {
                const int r = 1 + it / NTC, nt = it % NTC;
                uint4 b = uint4{0u, 0u, 0u, 0u};
                if (fg < SPT) b = *reinterpret_cast<const uint4*>(bbuf + (r * 16 + frow) * AB + fg * 16);
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                Mma<DT>::run(wfr[(nfa + nfb + nt) * 64 + lane], b, acc);
                const int ch = nt * 16 + fg * 4;
                f32x4 res = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PROJ) {
#pragma unroll
                    for (int blk = 0; blk < NB1; ++blk) {
                        const int sl = 4 * blk + fg;
                        uint4 p = uint4{0u, 0u, 0u, 0u};
                        if (sl < SPTA) p = *reinterpret_cast<const uint4*>(xt + (r * 16 + frow) * PBI + sl * 16);
                        Mma<DT>::run(wfr[(nfa + nfb + NTC + blk * NTC + nt) * 64 + lane], p, acc);
                    }
                } else {
                    res = Vec4<DT>::load(xt + (r * 16 + frow) * PBI + ch * 2);
                }
                const f32x4 sv = *reinterpret_cast<const f32x4*>(bn + 128 + ch), hv = *reinterpret_cast<const f32x4*>(bn + 128 + C + ch);
                f32x4 v = acc * sv + hv + res;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
                Vec4<DT>::store(ybuf + ((r - 1) * 16 + frow) * PB + ch * 2, v);
            }
        }
    }
    __syncthreads();
    if (t1 > t0) store_frame(t1 - 1);
}

// patch rows / time segments for a layer; false: the block does not take this path.  d1 != null: block 0 of a stage with a
// 1x1x1 stride-1 projection shortcut over the block's input (cin = d1->cin channels)
static bool abc_geometry(const af_conv_desc* da, const af_conv_desc* db, const af_conv_desc* dc, const af_conv_desc* d1, int* ph, int* ts_n) {
    if (!da || !db || !dc) return false;
    if (da->dtype == AF_F32 || db->dtype != da->dtype || dc->dtype != da->dtype) return false;
    const int inner = da->cout, cin = da->cin;
    // (inner 32 - s4 of the Fast pathway - runs and is parity-tested, but its 50 KB of weight fragments leave room for 3-row patches
    //  only and it measured 0.082 ms against 0.075 for the three launches: offered only with AF_ABC_INNER32=1)
    if (inner == 32 && !(getenv("AF_ABC_INNER32") && atoi(getenv("AF_ABC_INNER32")))) return false;
    if ((inner != 8 && inner != 16 && inner != 32) || db->cin != inner || db->cout != inner || dc->cin != inner || dc->cout != 4 * inner) return false;
    if (d1) {   // projection form: instantiated for the Fast pathway's s2 (8 -> 8 -> 8 -> 32)
        if (inner != 8 || cin != 8 || d1->dtype != da->dtype || d1->cin != cin || d1->cout != dc->cout) return false;
        if (d1->kt != 1 || d1->kh != 1 || d1->kw != 1 || d1->st != 1 || d1->sh != 1 || d1->sw != 1 || d1->pt || d1->ph || d1->pw || d1->tpool) return false;
        if (d1->n != da->n || d1->t != da->t || d1->h != da->h || d1->w != da->w) return false;
    } else if (cin != 4 * inner) {
        return false;
    }
    if ((da->kt != 1 && da->kt != 3) || da->kh != 1 || da->kw != 1 || da->pt != da->kt / 2 || da->ph || da->pw) return false;
    if (db->kt != 1 || db->kh != 3 || db->kw != 3 || db->pt || db->ph != 1 || db->pw != 1) return false;
    if (dc->kt != 1 || dc->kh != 1 || dc->kw != 1 || dc->pt || dc->ph || dc->pw) return false;
    const af_conv_desc* all[3] = {da, db, dc};
    for (const af_conv_desc* d : all) {
        if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->tpool || !d->relu) return false;
        if (d->n != da->n || d->t != da->t || d->h != da->h || d->w != da->w || d->to != da->t || d->ho != da->h || d->wo != da->w) return false;
    }
    if ((long long)da->h * da->w * 8 * inner >= (1LL << 31)) return false;
    // patch rows: the phases of a frame are short dependent chains (LDS read -> MFMA -> BN -> LDS write, a barrier between them),
    // so what pays is latency hiding, not the smaller halo of a taller patch.  Measured on SlowFast's Fast pathway (B = 16, bf16;
    // AF_ABC_PH sweeps it): s2 (inner 8) 14 / 10 / 7 / 5 / 4 / 3 rows -> 0.097 / 0.098 / 0.080 / 0.082 / 0.099 / 0.104 ms per
    // block (7 rows = 58 KB of LDS: two workgroups per CU, 8 even bands of a 56-row frame); s3 (inner 16) 10 / 7 / 5 / 4 / 3 rows ->
    // 0.070 / 0.053 / 0.081 / 0.054 / 0.082 ms.
    const char* eph = getenv("AF_ABC_PH");
    int p = eph ? atoi(eph) : 7;
    if (p > da->h) p = da->h;
    if (p < 1) p = 1;
    while (p > 1 && abc_lds_bytes(inner, cin, p, da->kt) > 160 * 1024) --p;
    p = (da->h + (da->h + p - 1) / p - 1) / ((da->h + p - 1) / p);                  // even bands
    if (abc_lds_bytes(inner, cin, p, da->kt) > 160 * 1024 || ((p + 2) * 16 * (cin / 8) + 63) / 64 > 24) return false;   // <= 6 DMA pieces per producer wave and frame
    const long long units = (long long)da->n * ((da->h + p - 1) / p) * ((da->w + 13) / 14);
    int s = 1;
    while (units * s < 256 && da->t / (2 * s) >= 4) s *= 2;           // time segments of >= 4 frames until the chip is covered
    if (units * s > 0x7fffffffLL) return false;
    *ph = p; *ts_n = s;
    return true;
}

template <int DT, int INNER, int CIN>
static int launch_abc(const ABCArgs& a, int units, int lds, hipStream_t stream) {
    AF_SET_MAX_LDS((&block_abc_kernel<DT, INNER, CIN>), 160 * 1024, "block_abc");
    hipLaunchKernelGGL((block_abc_kernel<DT, INNER, CIN>), dim3(units), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("block_abc_kernel");
    return AF_OK;
}

}  // namespace af

extern "C" int af_block_abc_fusable(const af_conv_desc* da, const af_conv_desc* db, const af_conv_desc* dc, const af_conv_desc* d1) {
    int ph, ts;
    return af::abc_geometry(da, db, dc, d1, &ph, &ts) ? 1 : 0;
}

extern "C" int af_block_abc_bn_act(const af_conv_desc* da, const void* x, const void* wa_packed, const float* scale_a, const float* shift_a,
                                   const af_conv_desc* db, const void* wb_packed, const float* scale_b, const float* shift_b,
                                   const af_conv_desc* dc, const void* wc_packed, const float* scale_c, const float* shift_c,
                                   const af_conv_desc* d1, const void* w1_packed, void* out, int out_ld, void* stream) {
    using namespace af;
    AF_REQUIRE(da && db && dc && x && wa_packed && wb_packed && wc_packed && scale_a && shift_a && scale_b && shift_b && scale_c && shift_c && out,
               "block_abc: null argument");
    AF_REQUIRE((d1 != nullptr) == (w1_packed != nullptr), "block_abc: the projection shortcut needs both its descriptor and its weight");
    AF_REQUIRE(aligned16(x) && aligned16(wa_packed) && aligned16(wb_packed) && aligned16(wc_packed) && aligned16(w1_packed) && aligned16(scale_a) &&
                   aligned16(shift_a) && aligned16(scale_b) && aligned16(shift_b) && aligned16(scale_c) && aligned16(shift_c) && aligned16(out),
               "block_abc: buffers must be 16-byte aligned");
    int ph = 0, ts_n = 0;
    AF_REQUIRE(abc_geometry(da, db, dc, d1, &ph, &ts_n), "block_abc: this block does not take the fused path (ask af_block_abc_fusable first)");
    if (out_ld == 0) out_ld = dc->cout;
    AF_REQUIRE(out_ld >= dc->cout && out_ld % 8 == 0, "block_abc: bad out_ld %d", out_ld);
    AF_REQUIRE(x != out, "block_abc: in-place is not possible (neighbouring patches read each other's halo)");
    ABCArgs a;
    a.x = (const char*)x; a.wa = (const char*)wa_packed; a.wb = (const char*)wb_packed; a.wc = (const char*)wc_packed; a.w1 = (const char*)w1_packed;
    a.sa = scale_a; a.ha = shift_a; a.sb = scale_b; a.hb = shift_b; a.sc = scale_c; a.hc = shift_c;
    a.y = (char*)out; a.T = da->t; a.H = da->h; a.W = da->w; a.kta = da->kt; a.PH = ph;
    a.py_n = (da->h + ph - 1) / ph; a.px_n = (da->w + 13) / 14; a.ts_n = ts_n; a.TS = (da->t + ts_n - 1) / ts_n;
    a.out_ld = out_ld;
    const int units = da->n * a.py_n * a.px_n * a.ts_n;
    const int lds = abc_lds_bytes(da->cout, da->cin, ph, da->kt);
    hipStream_t s = (hipStream_t)stream;
    const bool bf = da->dtype == AF_BF16;
    if (d1) return bf ? launch_abc<AF_BF16, 8, 8>(a, units, lds, s) : launch_abc<AF_F16, 8, 8>(a, units, lds, s);
    if (da->cout == 8) return bf ? launch_abc<AF_BF16, 8, 32>(a, units, lds, s) : launch_abc<AF_F16, 8, 32>(a, units, lds, s);
    if (da->cout == 32) return bf ? launch_abc<AF_BF16, 32, 128>(a, units, lds, s) : launch_abc<AF_F16, 32, 128>(a, units, lds, s);
    return bf ? launch_abc<AF_BF16, 16, 64>(a, units, lds, s) : launch_abc<AF_F16, 16, 64>(a, units, lds, s);
}

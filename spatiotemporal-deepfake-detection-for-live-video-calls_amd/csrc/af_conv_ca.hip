// c(i) -> a(i+1) across a block boundary of s2, as ONE launch (16-bit operands):
//     x    = relu( bn_c(conv1x1x1_c(b)) + res )        the end of ResBlock i   (resnet_helper.py:304-325, 438-444)
//     aout = relu( bn_a(conv3x1x1_a(x)) )              the first conv of ResBlock i+1 (resnet_helper.py:267-281)
// b: [N][T][HW][64], res / x: [N][T][HW][C] (C = 256 in s2), aout: [N][T][HW][64].
//
// s2 is an HBM stream: per block the 822-MB trunk is written by `c`, read back by the next block's `a`, and
// read a third time as the residual.  The temporal conv needs frames t-1, t, t+1 of the SAME pixel, so conv311 already
// tiles M as (clip, P pixels, ALL T frames): a tile holds its own temporal halo.  This kernel keeps that tile and
// PRODUCES its K-slab image instead of fetching it: per 64-channel slab of the trunk the residual slab is DMA'd into the
// image rows, the c conv's 64 output channels of that slab are computed from the b tile (K = 64: 16 MFMAs per wave) and
// added in place (BN, + residual, ReLU, the one rounding), the finished slab is stored to HBM as whole 128-byte rows (the
// trunk must exist: it is the next residual) and multiplied by the three temporal taps right there.  The `a` conv's
// 822-MB read disappears: 2.05 GB per pair instead of 2.88.
// Persistent workgroups, stage stream across tiles: a-weights of the slab in a 2-slot ring, images in a 3-slot ring with the
// residual slab fetched TWO stages ahead (Little's law: one 56-KB stage in flight per CU gave 4.1 TB/s), the c weights and
// the b fragments (K = 64: four 16-byte fragments per lane and tile) go global -> registers a stage / a tile ahead; BN
// parameters sit in LDS (a plain global load between the DMA issue and its use would make hipcc wait for the DMA in flight).
#include "af_common.h"
#include <stdlib.h>

namespace af {

struct CAArgs {
    const char* inb;     // [N][T][HW][64]
    const char* wc;      // packed [C][64]
    const float* scale_c;
    const float* shift_c;
    const char* res;     // [N][T][HW][C]   (plain blocks)
    const char* in1;     // [N][T][HW][64]  DUAL: the projection shortcut's input (block 0 of the stage), no residual
    const char* w1;      // packed [C][64]  DUAL: shortcut weights (both weight sets carry their BN scale, scale_c = ones)
    char* outx;          // [N][T][HW][C]
    const char* wa;      // packed [64][3][C]
    const float* scale_a;
    const float* shift_a;
    char* outa;          // [N][T][HW][64]
    int T, HW, C, kslabs;
    int P, chunks, tiles;
#ifdef AF_STAMPS
    int dbg;             // diagnostic build only: timing-only ablations (AF_CA_DBG; outputs are then garbage): 1 no c-weight loads after
                         // the first stage, 2 no b-fragment loads after the first tile, 4 no `a` output stores, 8 no trunk stores,
                         // 16 the b / x0 bytes as whole 128-byte rows (8 rows x 8 chunks per instruction; the operands are then garbage)
#endif
};
#ifdef AF_STAMPS
#define CA_DBG(bit) (a.dbg & (bit))
#else
#define CA_DBG(bit) false
#endif

// DUAL: block 0 of a stage - x = relu(bn_c(c(b)) + bn_1(branch1(x0))) with the 1x1x1 projection shortcut as a second K segment
// of the same accumulator (no residual tensor: the image rows are written, not updated; their padding frames are zeroed once).
// CWL (round 4, plain form): the c weights of a stage come in as ONE 8-KB LDS-DMA image per workgroup (a piece per wave) and every
// wave takes its fragments from there.  Timing builds (tools/exp_ca_dbg.py): the eight fragment-order global loads per wave and
// stage - the same 8 KB for all eight waves, 16 rows x 64 bytes per instruction - were 15 % of the launch (0.432 -> 0.374 ms
// without them) although they add nothing to the HBM bytes: a memory instruction holds its wave for 200-300 cycles whatever it
// fetches.  The 8 KB come from the padding frames: consecutive image slots now SHARE the zero rows between them ([Z][I0][Z][I1][Z][I2][Z],
// zeroed once, never fetched - which also makes it exactly four image pieces per wave and stage).
template <int DT, bool DUAL, bool CWL>
__global__ __launch_bounds__(512, 2) void conv_ca_kernel(const CAArgs a) {
    typedef Elem<DT> E;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int BM = 256, TN = 4, TM = 2;
    constexpr int WROWS = 3 * 64;                      // a-weight rows per stage: (dt, channel)
    constexpr int WPIECES = WROWS / 64;                // weight DMA pieces per wave
    constexpr int XPW = 4;                             // image pieces per wave: the tile's 256 rows (the padding frames are never fetched)
    static_assert(!(DUAL && CWL), "the projection form keeps both c weight sets in LDS anyway");
    constexpr int WBYTES = WROWS * 128;

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int P = a.P;
    const int slot_bytes = (BM + P) * 128;             // a slot = its leading padding frame + the tile's rows; the trailing one is the next slot's
    constexpr int NIMG = DUAL ? 1 : 3;
    // LDS: a-weight ring (2 slots) | image ring (3 slots: the residual slab is fetched TWO stages ahead - one stage in flight
    // is ~56 KB per CU, which at the loaded HBM latency is 4 TB/s; two are what the stream needs) + one more padding frame |
    // (CWL) the c weights of the stage | BN parameters
    char* sm = reinterpret_cast<char*>(smem);
    // (DUAL: no residual is fetched, the c conv WRITES the slab: one image buffer; the room goes to the two c weight sets,
    //  [C][64] each, resident in LDS for the life of the workgroup)
    char* img0 = sm + 2 * WBYTES;
    const int img_all = NIMG * slot_bytes + P * 128;
    char* cwl = img0 + img_all;                        // CWL: [64 channels][64 k], rows swizzled like the rings
    char* wlds = cwl + (CWL ? 64 * 128 : 0);
    float* bnp = reinterpret_cast<float*>(wlds + (DUAL ? 2 * a.C * 128 : 0));     // scale_c[C] shift_c[C] scale_a[64] shift_a[64]
    const int wm = wave;                               // 32-row group of the tile

    for (int i = tid; i < a.C; i += 512) { bnp[i] = a.scale_c[i]; bnp[a.C + i] = a.shift_c[i]; }
    {                                                  // the padding frames are zero for good: no DMA touches them, the c conv writes tile rows only
        uint4* z = reinterpret_cast<uint4*>(img0);
        for (int i = tid; i < img_all / 16; i += 512) z[i] = uint4{0u, 0u, 0u, 0u};
    }
    if (tid < 64) { bnp[2 * a.C + tid] = a.scale_a[tid]; bnp[2 * a.C + 64 + tid] = a.shift_a[tid]; }

    // ---- producer state
    const int drow = lane >> 3, chunk = (lane & 7) ^ drow;
    const long long Kw = 3LL * a.C;                                  // a-weight row length (elements)
    const i32x4 wdesc = make_desc(a.wa);
    unsigned woff[WPIECES];
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
        const int row = (wave + 8 * i) * 8 + drow, dt = row / 64, ch = row % 64;
        woff[i] = (unsigned)((ch * Kw + (long long)dt * a.C) * 2 + chunk * 16);
    }
    unsigned xoff[XPW];
    int xp[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int row = (wave + 8 * i) * 8 + drow;                   // tile row = t * P + p
        const int t = row / P, p = row % P;
        xp[i] = p;
        xoff[i] = (unsigned)((((long long)t * a.HW + p) * a.C) * 2 + chunk * 16);
    }
    const long long clipx = (long long)a.T * a.HW * a.C * 2, clipb = (long long)a.T * a.HW * 64 * 2;

    const int my_tiles = (a.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * a.kslabs;
    // stage index -> (tile, slab)
    auto stage_tile = [&](int g) { return (int)blockIdx.x + (g / a.kslabs) * (int)gridDim.x; };
    auto issue_weights = [&](int g) {                                // a-weights of stage g -> weight slot g & 1
        const unsigned base = lds0 + (g & 1) * WBYTES + wave * (8 * 128);
        const int soff = (g % a.kslabs) * 128;
#pragma unroll
        for (int i = 0; i < WPIECES; ++i) blds16_m0(woff[i], wdesc, soff, base + i * (64 * 128));
    };
    auto issue_image = [&](int g) {                                  // residual slab of stage g -> image slot g % 3
        const int tile = stage_tile(g), n = tile / a.chunks, hw0 = (tile % a.chunks) * P;
        const i32x4 xdesc = make_desc(a.res + n * clipx + (long long)hw0 * a.C * 2);
        const unsigned base = lds0 + 2 * WBYTES + (g % 3) * slot_bytes + P * 128 + wave * (8 * 128);
        const int soff = (g % a.kslabs) * 128;
#pragma unroll
        for (int i = 0; i < XPW; ++i)
            blds16_nt_m0(hw0 + xp[i] < a.HW ? xoff[i] : kOutOfRange, xdesc, soff, base + i * (64 * 128));
    };
    const i32x4 cdesc = make_desc(a.wc);
    auto issue_cw = [&](int g) {                                     // CWL: c weights of stage g's slab, one piece (8 channels) per wave
        blds16_m0((unsigned)((wave * 8 + drow) * 128 + chunk * 16), cdesc, (g % a.kslabs) * (64 * 128), lds0 + (unsigned)(cwl - sm) + wave * (8 * 128));
    };
    // c weights of a stage (4 channel tiles x 2 k-halves) and the b fragments of a tile (2 row tiles x 2 k-halves):
    // global -> registers, a stage / a tile ahead
    uint4 wcur[TN][2], bcur[TM][2], x0cur[TM][2];
    uint4 wnext[TN][2], bnext[TM][2], x0next[TM][2];
    auto load_wc = [&](uint4 (&dst)[TN][2], const char* wsrc, int kc) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                dst[i][kk] = gload16_uncounted(wsrc + ((long long)(kc * 64 + i * 16 + frow) * 64 + kk * 32 + fg * 8) * 2);
    };
    auto load_b = [&](uint4 (&dst)[TM][2], const char* bsrc, int tile) {
        const int n = tile / a.chunks, hw0 = (tile % a.chunks) * P;
        const char* bb = bsrc + n * clipb + (long long)hw0 * 64 * 2;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * 32 + j * 16 + frow, t = r / P, p = r - t * P;
            const int pc = hw0 + p < a.HW ? p : 0;                    // a pixel beyond the frame: any valid row (its outputs are dropped)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (CA_DBG(16)) {                                    // timing only: the same bytes as whole 128-byte rows (8 rows x 8 chunks per instruction)
                    const int r8 = wm * 32 + (2 * j + kk) * 8 + (lane >> 3), t8 = r8 / P, p8 = r8 - t8 * P;
                    dst[j][kk] = gload16_uncounted(bb + (((long long)t8 * a.HW + (hw0 + p8 < a.HW ? p8 : 0)) * 64 + (lane & 7) * 8) * 2);
                } else
                dst[j][kk] = gload16_uncounted(bb + (((long long)t * a.HW + pc) * 64 + kk * 32 + fg * 8) * 2);
            }
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (total > 0) {
        if (!DUAL && !CWL) load_wc(wnext, a.wc, 0);
        load_b(bnext, a.inb, blockIdx.x);
        if (DUAL) {
            load_b(x0next, a.in1, blockIdx.x);
            const i32x4 d0 = make_desc(a.wc), d1 = make_desc(a.w1);          // both c weight sets -> LDS, rows swizzled like the rings
            const unsigned off = (unsigned)(drow * 128 + chunk * 16), base = lds0 + (unsigned)(wlds - sm) + wave * (8 * 128);
            for (int g = 0; g < a.C / 64; ++g) {
                blds16(off, d0, (g * 64 + wave * 8) * 128, base + g * (64 * 128));
                blds16(off, d1, (g * 64 + wave * 8) * 128, base + a.C * 128 + g * (64 * 128));
            }
        }
        issue_weights(0);
        if (CWL) issue_cw(0);
        if (!DUAL) {
            issue_image(0);
            if (total > 1) issue_image(1);
        }
    }
    __syncthreads();                                                 // BN parameters visible
    int c_tile = blockIdx.x, c_kc = 0;                               // consumer cursor
    for (int q = 0; q < total; ++q) {
        // weights(q) and image(q) have landed once only image(q+1)'s pieces (issued right behind weights(q)) and the 4 trunk-row
        // stores of the previous iteration are still in flight (the 8 `a` output stores behind a tile's last slab are waited for:
        // once per tile, and the count does not depend on how hipcc emits them; leaving them in flight too measured no gain)
        if (q == 0) wait_vmcnt<0>();
        else if (!DUAL && q + 1 < total) wait_vmcnt<8>();
        else wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();                                // ... for everyone; weight slot (q+1)&1, image slot (q+2)%3 are free
        // (the c weights / b fragments are loads hipcc does not count - its own waits for them were vmcnt(7)..(0) at this point,
        //  which drained the image DMA issued behind them: ONE stage in flight instead of two; the empty asm orders the copies
        //  behind our wait)
        if (CWL) {                                                   // this stage's c weights: fragments from the LDS image
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    wcur[i][kk] = *reinterpret_cast<const uint4*>(cwl + (i * 16 + frow) * 128 + (((kk * 4 + fg) ^ (frow & 7)) << 4));
        } else if (!DUAL) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 t = __builtin_bit_cast(u32x4, wnext[i][kk]);
                    asm volatile("" : "+v"(t));
                    wcur[i][kk] = __builtin_bit_cast(uint4, t);
                }
        }
        if (c_kc == 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 t = __builtin_bit_cast(u32x4, bnext[j][kk]);
                    asm volatile("" : "+v"(t));
                    bcur[j][kk] = __builtin_bit_cast(uint4, t);
                    if (DUAL) {
                        u32x4 t0 = __builtin_bit_cast(u32x4, x0next[j][kk]);
                        asm volatile("" : "+v"(t0));
                        x0cur[j][kk] = __builtin_bit_cast(uint4, t0);
                    }
                }
        }
        const bool last = c_kc + 1 == a.kslabs;
        if (q + 1 < total) {
            // plain loads BEFORE the DMA issue: hipcc's wait for them must not cover the DMA
            if (!DUAL && !CWL && !CA_DBG(1)) load_wc(wnext, a.wc, last ? 0 : c_kc + 1);
            if (last && !CA_DBG(2)) {
                load_b(bnext, a.inb, c_tile + gridDim.x);
                if (DUAL) load_b(x0next, a.in1, c_tile + gridDim.x);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_weights(q + 1);
            if (!DUAL && !CWL && q + 2 < total) issue_image(q + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int n = c_tile / a.chunks, hw0 = (c_tile % a.chunks) * P;
        char* img = img0 + (DUAL ? 0 : q % 3) * slot_bytes;          // image rows: (t + 1) * P + p
        // ---- c conv, 64 trunk channels of this slab, in place: image = relu(bn_c(Wc b) + image)
        {
            f32x4 cc[TN][TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) cc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (round 4) the eight residual cells of this wave are read BEFORE the c conv's MFMAs and arrive under them: read where
            // they are added, every cell was an LDS round trip of its own in front of its update - read, lgkmcnt(0), 25 vector
            // operations, write, eight times per stage on an in-order wave
            u32x2 rcell[TN][TM];
            if (!DUAL) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        const int R = P + wm * 32 + j * 16 + frow;
                        rcell[i][j] = *reinterpret_cast<const u32x2*>(img + R * 128 + (((i * 2 + (fg >> 1)) ^ (frow & 7)) << 4) + (fg & 1) * 8);
                    }
            }
            // ... and so are the BN parameters of channel tiles 0 and 1; those of tiles 2 and 3 are read into the same registers as
            // soon as tiles 0 and 1 are done (two tiles of look-ahead instead of a round trip per tile)
            f32x4 scv[2], sfv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                scv[i] = *reinterpret_cast<const f32x4*>(bnp + c_kc * 64 + i * 16 + fg * 4);
                sfv[i] = *reinterpret_cast<const f32x4*>(bnp + a.C + c_kc * 64 + i * 16 + fg * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (DUAL) {                                          // both weight slabs from their LDS images
                    const uint4* wl = reinterpret_cast<const uint4*>(wlds) + (c_kc * 64 + frow) * 8 + ((kk * 4 + fg) ^ (frow & 7));
#pragma unroll
                    for (int i = 0; i < TN; ++i) {
                        const uint4 w0 = wl[i * 16 * 8], w1 = wl[a.C * 8 + i * 16 * 8];
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            Mma<DT>::run(w0, bcur[j][kk], cc[i][j]);
                            Mma<DT>::run(w1, x0cur[j][kk], cc[i][j]);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j) Mma<DT>::run(wcur[i][kk], bcur[j][kk], cc[i][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const f32x4 sc = scv[i & 1], sf = sfv[i & 1];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int R = P + wm * 32 + j * 16 + frow;       // image row of this position (R & 7 == frow & 7: P % 8 == 0)
                    char* cell = img + R * 128 + (((i * 2 + (fg >> 1)) ^ (frow & 7)) << 4) + (fg & 1) * 8;
                    f32x4 v = cc[i][j] * sc + sf;
                    if (!DUAL) v += Vec4<DT>::unpack(rcell[i][j]);
                    Vec4<DT>::store_relu(cell, v);
                }
                if (i + 2 < TN) {
                    scv[i & 1] = *reinterpret_cast<const f32x4*>(bnp + c_kc * 64 + (i + 2) * 16 + fg * 4);
                    sfv[i & 1] = *reinterpret_cast<const f32x4*>(bnp + a.C + c_kc * 64 + (i + 2) * 16 + fg * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the slab of x is complete (raw barrier: the DMA stays in flight)
        if (CWL && q + 1 < total) {
            // everyone has its fragments of this stage's c weights: the next stage's go in - in FRONT of image(q + 2), so that the
            // counted wait at the top of the next stage (which must cover them) still leaves that image in flight
            if (!CA_DBG(1)) issue_cw(q + 1);
            if (q + 2 < total) issue_image(q + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the finished slab leaves for HBM as whole 128-byte row segments
        {
            char* ox = a.outx + ((long long)n * a.T * a.HW + hw0) * a.C * 2 + c_kc * 128;
#pragma unroll
            for (int it = 0; it < BM * 8 / 512; ++it) {
                const int row = (tid >> 3) + 64 * it, ck = tid & 7;
                const int t = row / P, p = row - t * P;
                const u32x4 o = *reinterpret_cast<const u32x4*>(img + (P + row) * 128 + ((ck ^ ((P + row) & 7)) << 4));
                if (hw0 + p < a.HW && !CA_DBG(8))
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(ox + ((long long)t * a.HW + p) * a.C * 2 + ck * 16));
            }
        }
        // ---- a conv: three temporal taps of the slab
        const uint4* ws = smem + (q & 1) * (WBYTES / 16) + frow * 8;
        const uint4* xs = reinterpret_cast<const uint4*>(img) + (wm * 32 + frow) * 8;
        // the six (tap, k-half) steps, software-pipelined by hand: step s + 1's fragments are read while step s multiplies
        // (left to hipcc the reads sat right in front of their MFMAs: a chain of LDS latencies, as in conv311)
        {
            uint4 af[2][TN], bf[2][TM];
            auto read_step = [&](uint4 (&fa)[TN], uint4 (&fb)[TM], int step) {
                const int dt = step >> 1, c = ((step & 1) * 4 + fg) ^ (frow & 7);
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[i] = ws[(dt * 64 + i * 16) * 8 + c];
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
        }
        if (!last) { ++c_kc; continue; }

        // ---- tile finished: BN + ReLU + the one rounding; two channel tiles trade halves between lane rows (v_permlane16_swap), so a
        // lane stores 16 contiguous bytes (8 channels) and a wave instruction 64-byte row segments - no LDS transposition
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * 32 + j * 16 + frow;                   // tile row = t * P + p
            const int t = r / P, p = r - t * P;
            const long long pos = ((long long)n * a.T + t) * a.HW + hw0 + p;
#pragma unroll
            for (int ip = 0; ip < TN / 2; ++ip) {
                u32x2 half[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = 2 * ip + k;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(bnp + 2 * a.C + i * 16 + fg * 4);
                    const f32x4 sf = *reinterpret_cast<const f32x4*>(bnp + 2 * a.C + 64 + i * 16 + fg * 4);
                    half[k] = Vec4<DT>::pack_relu(acc[i][j] * sc + sf);
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const u32x4 o = swap_pair16(half[0], half[1]);
                const int ch = (2 * ip + (fg & 1)) * 16 + (fg >> 1) * 8;
                if (hw0 + p < a.HW && !CA_DBG(4)) *reinterpret_cast<u32x4*>(a.outa + (pos * 64 + ch) * 2) = o;
            }
        }
        c_kc = 0; c_tile += gridDim.x;
    }
}

// dynamic LDS of a launch: a-weight ring (2 slots) + image slot(s) (+ both c-side weight sets, DUAL) + the BN parameters
static long long conv_ca_lds_bytes(bool dual, int P, int C, bool cwl = false) {
    return 2LL * 3 * 64 * 128 + (dual ? (256LL + 2 * P) * 128 + 2LL * C * 128 : (3LL * (256 + P) + P) * 128 + (cwl ? 64 * 128 : 0)) + (2LL * C + 128) * 4;
}

template <int DT, bool DUAL, bool CWL>
static int launch_ca(const CAArgs& a, int blocks, hipStream_t stream) {
    const int lds = (int)conv_ca_lds_bytes(DUAL, a.P, a.C, CWL);
    if (lds > 160 * 1024) return set_error(AF_ERR_ARG, "conv_ca: %d bytes of LDS needed", lds);
    AF_SET_MAX_LDS((&conv_ca_kernel<DT, DUAL, CWL>), 160 * 1024, "conv_ca");
    hipLaunchKernelGGL((conv_ca_kernel<DT, DUAL, CWL>), dim3(blocks), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv_ca_kernel");
    return AF_OK;
}

// dc: the 1x1x1 `c` conv (64 -> C, with residual + ReLU), da: the 3x1x1 `a` conv of the next block (C -> 64) over its output
// d1 (optional): the projection shortcut of block 0 - a 1x1x1 stride-1 conv from 64 channels onto the same positions
bool conv_ca_applies(const af_conv_desc* dc, const af_conv_desc* d1, const af_conv_desc* da) {
    if (d1) {
        if (d1->dtype != dc->dtype || d1->tpool || d1->kt != 1 || d1->kh != 1 || d1->kw != 1 || d1->st != 1 || d1->sh != 1 || d1->sw != 1) return false;
        if (d1->pt || d1->ph || d1->pw || d1->cin != 64 || d1->cout != dc->cout) return false;
        if (d1->n != dc->n || d1->t != dc->t || d1->h != dc->h || d1->w != dc->w) return false;
    }
    if (!dc || !da || dc->dtype == AF_F32 || da->dtype != dc->dtype || dc->tpool || da->tpool) return false;
    if (dc->kt != 1 || dc->kh != 1 || dc->kw != 1 || dc->st != 1 || dc->sh != 1 || dc->sw != 1 || dc->pt || dc->ph || dc->pw) return false;
    if (da->kt != 3 || da->kh != 1 || da->kw != 1 || da->st != 1 || da->sh != 1 || da->sw != 1 || da->pt != 1 || da->ph || da->pw) return false;
    if (dc->cin != 64 || da->cout != 64 || dc->cout != da->cin || dc->cout % 64 != 0 || dc->cout > 1024 || !dc->relu || !da->relu) return false;
    if (dc->n != da->n || dc->t != da->t || dc->h != da->h || dc->w != da->w) return false;
    if (dc->t != 16 && dc->t != 32) return false;                     // tile = all T frames x 256 / T pixels
    const long long hw = (long long)dc->h * dc->w;
    if ((long long)dc->t * hw * dc->cout * 2 >= (1LL << 31)) return false;     // 32-bit offsets inside a clip
    const int p = 256 / dc->t;
    // the launch must fit LDS (wide trunks: the plain form with T = 16 and C >= 512, the projection form with C >= 512 do not):
    // such a pair keeps its two launches instead of failing in launch_ca
    if (conv_ca_lds_bytes(d1 != nullptr, p, dc->cout) > 160 * 1024) return false;
    // whole tiles only: the kernel's counted s_waitcnt vmcnt assume that every wave issues every store of an iteration; a
    // ragged last chunk could mask ALL lanes of a wave's store, hipcc would branch around it and the count would be off
    if (hw % p != 0) return false;
    const long long tiles = (long long)dc->n * (hw / p);
    // persistent stream: pays with several tiles per workgroup (small batches keep the two launches)
    return tiles >= 4LL * device_cus() && tiles < (1LL << 31);
}

int conv_ca_run(const af_conv_desc* dc, const void* inb, const void* wc, const void* in1, const void* w1, const float* scale_c,
                const float* shift_c, const void* residual, void* outx, const af_conv_desc* da, const void* wa, const float* scale_a,
                const float* shift_a, void* outa, hipStream_t stream) {
    CAArgs a;
    a.inb = (const char*)inb; a.wc = (const char*)wc; a.scale_c = scale_c; a.shift_c = shift_c; a.res = (const char*)residual;
    a.in1 = (const char*)in1; a.w1 = (const char*)w1;
    a.outx = (char*)outx; a.wa = (const char*)wa; a.scale_a = scale_a; a.shift_a = shift_a; a.outa = (char*)outa;
    a.T = dc->t; a.HW = dc->h * dc->w; a.C = dc->cout; a.kslabs = dc->cout / 64;
    a.P = 256 / dc->t; a.chunks = (a.HW + a.P - 1) / a.P; a.tiles = dc->n * a.chunks;
#ifdef AF_STAMPS
    const char* ed = getenv("AF_CA_DBG");
    a.dbg = ed ? atoi(ed) : 0;
#endif
    const int cus = device_cus();
    const int blocks = a.tiles < cus ? a.tiles : cus;
    if (in1) return dc->dtype == AF_BF16 ? launch_ca<AF_BF16, true, false>(a, blocks, stream) : launch_ca<AF_F16, true, false>(a, blocks, stream);
    // the stage's c weights as an LDS image where its 8 KB fit (T = 32 with a 256-channel trunk; AF_CA_CWL=0: the fragment loads, for A/B runs)
    const char* ecw = getenv("AF_CA_CWL");
    if (conv_ca_lds_bytes(false, a.P, a.C, true) <= 160 * 1024 && !(ecw && atoi(ecw) == 0))
        return dc->dtype == AF_BF16 ? launch_ca<AF_BF16, false, true>(a, blocks, stream) : launch_ca<AF_F16, false, true>(a, blocks, stream);
    return dc->dtype == AF_BF16 ? launch_ca<AF_BF16, false, false>(a, blocks, stream) : launch_ca<AF_F16, false, false>(a, blocks, stream);
}

}  // namespace af

// 1x3x3 / stride 1 / pad (0,1,1) convolution, 128 -> 128 or 256 -> 256 channels (any Cin % 64 == 0), 16-bit operands,
// + BN + ReLU: the `b` convs of the s3 / s4 bottlenecks (reference altfreezing/slowfast/models/resnet_helper.py:283-297).
// Round 3: also kT = 3 (3x3x3, pad (1,1,1)) - the synthetic "3x3x3 Conv3d" of BASELINE.json's metric; the reference model has
// no such layer - as a longer K loop over (dt, channel slab) whose patches come from frames t - 1, t, t + 1 (a frame outside
// the clip = a patch of out-of-range lanes = zeros), and a 64-channel instantiation (8 position groups of 4 m-tiles).
//
// The generic implicit GEMM brings every activation row into LDS once PER TAP (9x) and cuts M into 256 / 512-row tiles:
// 196 / 392 tiles for the 256 frames of a 16-clip batch, i.e. 77 % of the CUs in the last round.  Here the work unit is
// a FRAME (s4: 14 x 14) or a band of rows of one (s3: 14 of 28 rows) and all output channels of it:
//   * per 64-channel K slab the unit's input patch ((R + 2) rows, zero halo included) enters LDS ONCE (LDS-DMA, buffer
//     loads: out-of-image lanes = out-of-range offsets = zeros) and all 9 taps read it; with the rows kept at a padded
//     pitch WP = W + 2 a tap (dh, dw) is the constant row shift dh * WP + dw (as in conv133_c64), so every B fragment
//     is a swizzled ds_read_b128 at an immediate offset; patches are double-buffered, the next slab's pieces trickle
//     in one per K-step.  (Round 3: the pitch is W + 1 - the right halo of row r IS the left halo of row r + 1, both are
//     zero - so a 56-wide band of 8 rows + halo fits twice next to the weight ring: 2 x 72 KB + 16 KB = all of LDS.);
//   * only the weights stream per K-step (tap, slab): a [Cout x 64] tile through a 2-slot LDS-DMA ring - half the
//     L2 -> LDS ingest per MAC of the generic 256 x 256 tile;
//   * output positions are the padded band (R x WP; the two halo columns per row are computed and dropped: 12.5 % at
//     W = 14, 6.7 % at W = 28), 8 waves = WN channel groups of 64 x WM position groups of 7 m-tiles (28 accumulator
//     tiles per wave); the K loop is software-pipelined by k-halves like the generic kernel's 2-slot variant;
//   * at B = 16 the 256 frames of s4 are exactly one unit per CU, the 512 half-frames of s3 exactly two rounds.
#include "af_common.h"
#include <stdlib.h>

namespace af {

// Diagnostic build (-DAF_STAMPS, tools/stamps_lib.sh; never the shipped library): shader-clock stamps around the phases of a
// unit, kept in scalar registers and written behind the last output store, to a buffer nothing else reads.
#ifdef AF_STAMPS
#define AF_STAMP_DECL unsigned long long stamp_v[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define AF_DBG(bit) (a.dbg & (bit))
#define AF_STAMP(slot) stamp_v[slot] = (slot) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime()
#define AF_STAMP_FLUSH do { if (a.stamps && lane < 8) a.stamps[((long long)blockIdx.x * 8 + wave) * 8 + lane] = \
    lane == 0 ? stamp_v[0] : lane == 1 ? stamp_v[1] : lane == 2 ? stamp_v[2] : lane == 3 ? stamp_v[3] : lane == 4 ? stamp_v[4] : lane == 5 ? stamp_v[5] : lane == 6 ? stamp_v[6] : stamp_v[7]; } while (0)
#else
#define AF_STAMP_DECL do {} while (0)
#define AF_DBG(bit) false
#define AF_STAMP(slot) do {} while (0)
#define AF_STAMP_FLUSH do {} while (0)
#endif

struct C133GArgs {
    const char* in;
    const char* w;       // packed [Cout][9][Cin]
    const float* scale;
    const float* shift;
    char* out;
    int H, W, Cin, Cout, frames;
    int T, kt;           // frames per clip; temporal kernel size 1 or 3 (pad kt / 2)
    int R, upf;          // output rows per unit, units per frame
    int WP;              // padded pitch W + 1 (one shared zero column between consecutive rows)
    int prows;           // LDS rows per patch buffer (multiple of 8)
    int kslabs;          // Cin / 64
    int relu, out_ld;
    float inv_wp;
    // fused `c` conv (FUSEC instantiations): out = relu(bn2(conv1x1x1(b_out)) + res), b_out never leaves LDS
    const char* w2;      // packed [Cout2][1][Cout]
    const float* scale2;
    const float* shift2;
    const char* res;     // [frames][H][W][Cout2] or null
    int Cout2, relu2;
    int stagger;         // waves 4-7 take the K loop's barrier half an MFMA group late (AF_G_STAGGER=0 for A/B runs)
#ifdef AF_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/stamps_lib.sh): [unit][wave][8] shader-clock / wall-clock stamps
    int dbg;                      // timing-only ablations of the K loop (AF_G_DBG; outputs are then garbage, addresses unchanged):
                                  // 1 no vmcnt wait, 2 no barrier, 4 no weight DMA, 8 no LDS fragment reads, 16 no patch DMA
#endif
};

template <int DT, int WN, int WM, int MT, bool FUSEC, int MAXP, int NSLOT, bool TEMPORAL>
__global__ __launch_bounds__(512, 2) void conv133g_kernel(const C133GArgs a) {
    typedef Elem<DT> E;
    typedef typename E::type OT;
    static_assert(E::EPC == 8 && WN * WM == 8, "16-bit operands, 8 waves");
    constexpr int NT = 4;                              // 16x16 tiles per wave: 64 channels x MT * 16 positions (7: 112)
    constexpr int BN = WN * 64;                        // output channels of the workgroup (= Cout)
    constexpr int RW = BN / 64;                        // weight DMA pieces per wave per stage
    constexpr int WSTAGE = BN * 128;                   // bytes of a weight stage ([BN][64 k])
    // MAXP: patch DMA pieces per wave (patch rows <= 64 MAXP; one piece per tap, so <= 9: 576 rows)
    constexpr int PROW = 64 + 8;                       // epilogue patch row stride (elements)
    // TEMPORAL (round 4): the same kernel for 3x1x1 / stride 1 / pad (1,0,0) convs (the `a` convs of s3 / s4; reference
    // resnet_helper.py:267-281).  Work unit = (clip, P consecutive pixels, all T frames); the patch of a K slab is (T + 2) x P rows
    // (row j = frame j / P - 1 of pixel j % P; the two padding frames are out-of-range lanes), a tap is the row shift dt * P, and
    // WM x MT x 16 = T x P positions are all real.  Three taps share a patch where the generic kernel fetched the tile per tap.
    constexpr int NTAPS = TEMPORAL ? 3 : 9;
    // Patch pieces a K-step carries: the next slab's patch rides on this slab's steps.  Every piece the next slab's FIRST step reads
    // must be issued one step before this slab's last (whose wait + barrier then cover it; the look-ahead reads of the next slab
    // are issued inside the last step).  Spatial: nine pieces on nine steps - piece 8 is patch rows 512.., which tap 0 never reads;
    // temporal: every tap reads the whole patch, so the pieces ride on the first NTAPS - 1 steps.
    constexpr int PSTEPS = TEMPORAL ? NTAPS - 1 : NTAPS;
    constexpr int PPS = (MAXP + PSTEPS - 1) / PSTEPS;
    static_assert(!(TEMPORAL && FUSEC) && NSLOT <= NTAPS, "temporal mode: no fused c conv; the prologue's stages are taps of slab 0");

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int frow = lane & 15, fg = lane >> 4;
    const int WP = a.WP;
    const int pbytes = a.prows * 128;
    const unsigned ring0 = lds0 + 2 * pbytes;

    // ---- work units: (frame, band of R output rows); TEMPORAL: (clip = `frame`, chunk of R = P pixels starting at pixel h0).
    // Round 4: the workgroup is PERSISTENT - unit blockIdx.x, + gridDim.x, ... - and the next unit's first operands (the first NSLOT
    // weight stages, the patch of slab 0) are issued behind the K loop, under this unit's epilogue: a layer of several rounds of
    // units (s3: 2, the 56 x 56 `a` conv: 7) pays one DMA latency per workgroup instead of one per unit.
    const int units = a.frames * a.upf;
    int unit = blockIdx.x;
    int frame = unit / a.upf, h0 = (unit - frame * a.upf) * a.R;
    const int HW = a.H * a.W;
    // ---- producers.  Weights: thread (lrow = tid >> 3, slot = tid & 7) fetches chunk slot ^ (lrow & 7) of weight rows
    // lrow + 64 i; the K-step's (tap, slab) offset goes in an SGPR.
    const int lrow = tid >> 3, wchunk = (tid & 7) ^ (lrow & 7);
    const long long Kw = (long long)NTAPS * a.kt * a.Cin;              // weight row length (elements; TEMPORAL: kt is a tap, a.kt = 1)
    const i32x4 wdesc = make_desc(a.w);
    const unsigned woff = (unsigned)((lrow * Kw + wchunk * 8) * 2);    // piece i: + 64 i weight rows (added to the scalar offset)
    const int wpiece_bytes = (int)(64 * Kw * 2);
    // Patch: LDS row j <-> padded pixel q = j - 1 = (r, c) = (q / WP, q % WP) <-> input pixel (h0 - 1 + r, c - 1); rows
    // beyond the band, halo columns and rows outside the image are out-of-range lanes (zeros).  Piece g = rows 8g .. 8g+7.
    const int NP = a.prows >> 3;
    // K slab ks = (dt, channel slab cs): its patch comes from frame t + dt - kt / 2 of the clip; the descriptor starts kt / 2
    // frames early, dt advances by whole frames in the SGPR offset
    const int pt = a.kt >> 1;
    int tclip = frame % a.T;
    const int frame_bytes = a.H * a.W * a.Cin * 2;                     // < 2^29 (host-checked)
    auto unit_desc = [&]() {
        return make_desc(TEMPORAL ? a.in + ((((long long)frame * a.T - 1) * HW + h0) * a.Cin) * 2
                                  : a.in + ((((long long)frame - pt) * a.H + h0 - 1) * a.W) * a.Cin * 2);
    };
    i32x4 xdesc = unit_desc();
    // The per-lane source offsets of the patch pieces (row of the patch -> input element, or out of range).  256 channels: MAXP <= 5
    // registers; 64 channels: 9, next to 64 accumulator registers only.  128 channels (MAXP 8 - 9): a table in LDS, one word per patch row ([prows], behind the ring), read a step
    // ahead of its use - nine cold registers next to 200 of accumulators and fragments were what hipcc spilled, and a scratch
    // reload's vmcnt(0) inside the K loop drains the DMA.
    constexpr bool POFF_LDS = WN == 2;
    unsigned poff[POFF_LDS ? 1 : MAXP];
    const unsigned tab0 = ring0 + NSLOT * WSTAGE;                        // the offset table (POFF_LDS)
    const unsigned tabl = tab0 + (unsigned)((wave * 8 + (lane >> 3)) * 4);   // this lane's word of piece 0; piece i: + 256 i bytes
    const unsigned pchunk16 = (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
    // patch piece i of this wave (rows 8 (wave + 8 i) ..) of K slab (frame offset + channel slab = `soff`) into patch buffer at `bufoff`
    auto patch_slab_offset = [&](int ks, bool& inclip) {
        const int dt = (ks >= a.kslabs) + (ks >= 2 * a.kslabs), cs = ks - dt * a.kslabs;
        inclip = (unsigned)(tclip + dt - pt) < (unsigned)a.T;
        return __builtin_amdgcn_readfirstlane(dt * frame_bytes + cs * 128);
    };
    // (voff: the piece's per-lane offset - poff[i], or the table word of the lane's row + its 16-byte chunk)
    auto issue_patch_piece = [&](int bufoff, int soff, bool inclip, int i, unsigned voff) {
        if (wave + 8 * i < NP)
            blds16_m0(inclip ? voff : kOutOfRange, xdesc, soff, __builtin_amdgcn_readfirstlane(lds0 + bufoff + (wave + 8 * i) * 1024));
    };
    auto table_offset = [&](unsigned word) { return word == kOutOfRange ? kOutOfRange : word + pchunk16; };
    // weight stage of K-step (tap, slab ks) into the ring slot at byte offset `stoff`, one piece (64 rows) at a time; packed
    // [Cout][kt * 9][Cin]: offset inside a weight row = w_slab_offset(ks) + tap * Cin * 2
    auto w_slab_offset = [&](int ks) {
        const int dt = (ks >= a.kslabs) + (ks >= 2 * a.kslabs), cs = ks - dt * a.kslabs;
        return __builtin_amdgcn_readfirstlane((dt * NTAPS * a.Cin + cs * 64) * 2);
    };
    const int tap_bytes = a.Cin * 2;
    auto issue_w_piece = [&](int stoff, int soff, int i) {
        blds16_m0(woff, wdesc, soff + i * wpiece_bytes, __builtin_amdgcn_readfirstlane(ring0 + stoff + (64 * i + 8 * wave) * 128));
    };

    f32x4 acc[NT][MT];

    const int nslabs = a.kt * a.kslabs;                // K slabs (dt, cs)
    AF_STAMP_DECL;
    AF_STAMP(0); AF_STAMP(6);
    // ---- a unit's first operands: the first NSLOT weight stages (taps 0 .. NSLOT - 1 of slab 0) are issued before the per-lane
    // patch offsets are even computed (~100 vector instructions), then the patch of slab 0 into the buffer at `bufoff`; slab 1's
    // patch rides on the K-steps of slab 0 like every later one
    // source offset of patch row j (without the lane's chunk), or kOutOfRange
    auto row_offset = [&](int j) -> unsigned {
        if (TEMPORAL) {
            // LDS row j = (frame tf = j / P of the padded clip, pixel q = j % P) <-> input (n, tf - 1, h0 + q); the descriptor
            // starts one frame early
            const int tf = (int)(((float)j + 0.5f) * a.inv_wp), q = j - tf * a.R;
            const bool ok = tf >= 1 && tf <= a.T && h0 + q < HW;
            return ok ? (unsigned)((((long long)tf * HW + q) * a.Cin) * 2) : kOutOfRange;
        }
        const int q = j - 1;
        const int r = (int)(((float)q + 0.5f) * a.inv_wp), c = q - r * WP;
        const bool ok = q >= 0 && r < a.R + 2 && c >= 1 && c <= a.W && (unsigned)(h0 - 1 + r) < (unsigned)a.H;
        return ok ? (unsigned)(((r * a.W + (c - 1)) * a.Cin) * 2) : kOutOfRange;
    };
    auto compute_poff = [&]() {
        // (from an opaque copy of the thread / lane id: hipcc otherwise computes the unit-independent half of these offsets once,
        //  in front of the unit loop, and spills it - a scratch reload's vmcnt(0) in the prefetch block would drain the DMA around it)
        if constexpr (POFF_LDS) {
            int t = tid;
            asm volatile("" : "+v"(t));
            unsigned* tab = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + (tab0 - lds0));
            for (int j = t; j < a.prows; j += 512) tab[j] = row_offset(j);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // the table is complete for every wave
        } else {
            int drow = lane >> 3;
            asm volatile("" : "+v"(drow));
#pragma unroll
            for (int i = 0; i < MAXP; ++i) poff[i] = table_offset(row_offset((wave + 8 * i) * 8 + drow));
        }
    };
    auto issue_unit_head = [&](int bufoff) {
#pragma unroll
        for (int k = 0; k < NSLOT; ++k)
#pragma unroll
            for (int i = 0; i < RW; ++i) issue_w_piece(k * WSTAGE, w_slab_offset(0) + k * tap_bytes, i);
        compute_poff();
        bool in0;
        const int ps0 = patch_slab_offset(0, in0);
        if constexpr (POFF_LDS) {
            unsigned w[MAXP];
            static_for<MAXP>([&](auto i) { w[i] = lds_read4_uncounted<i * 256>(tabl); });
            wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < MAXP; ++i) { asm volatile("" : "+v"(w[i])); issue_patch_piece(bufoff, ps0, in0, i, table_offset(w[i])); }
        } else {
#pragma unroll
            for (int i = 0; i < MAXP; ++i) issue_patch_piece(bufoff, ps0, in0, i, poff[i]);
        }
    };
    issue_unit_head(0);
    int bstart = 0;                                    // patch buffer of the unit's slab 0 (every later unit: the second buffer)
    const unsigned arow = ring0 + (wn * 64 + frow) * 128, brow = lds0 + (wm * MT * 16 + frow) * 128;
    // ---- the K loop: every instruction of it is an asm statement (MFMAs in place, uncounted LDS reads, LDS-DMA, waits), issued
    // in program order; hipcc allocates the registers and does the address arithmetic.
    //   * The nine taps of a slab are unrolled: tap, its (dh, dw) row shift, the patch piece that rides on the step and the
    //     step two ahead are compile-time facts of each body.  (The dynamic cursor + a nine-way branch chain that picked the
    //     patch piece cost ~200 scalar instructions per 56 MFMAs, which two in-order waves per SIMD could not hide: with every
    //     memory operation removed the round-3 loop still ran at 0.8 of the MFMA rate.)
    //   * The fragments of the NEXT half-step are read between the MFMAs of this one, one ds_read_b128 per two MFMAs, so a
    //     read burst never stands between a barrier and the matrix pipe.
    //   * LATE waves (4-7, the SIMD partners of 0-3) take the step's one barrier H1 MFMAs into their first group instead of
    //     behind it: the partners' barrier waits, DMA issue and address arithmetic fall on each other's MFMA stretches
    //     (MI355X_MICROARCH.md, "two waves that run the same program with one barrier per block").  Same LDS protocol: a wave
    //     issues - and waits for - its reads of ring slot `stoff` / the previous slab's patch in front of its barrier.
    u32x4 a0[NT], b0[MT], a1[NT], b1[MT];
    auto frag_read = [&](u32x4 (&af)[NT], u32x4 (&bf)[MT], unsigned wsb, unsigned xsb, auto idx) {
        constexpr int r = decltype(idx)::value;
        if constexpr (r < NT) af[r] = lds_read16_uncounted<r * (16 * 128)>(wsb);
        else if constexpr (r < NT + MT) bf[r - NT] = lds_read16_uncounted<(r - NT) * (16 * 128)>(xsb);
    };
    auto w_addr = [&](int stoff, int kk) { return arow + stoff + ((((kk << 2) + fg) ^ (frow & 7)) << 4); };
    auto x_addr = [&](int bufoff, int sh, int kk) { return brow + bufoff + sh * 128 + ((((kk << 2) + fg) ^ ((frow + sh) & 7)) << 4); };
    auto pin_half = [&](u32x4 (&af)[NT], u32x4 (&bf)[MT]) {
#pragma unroll
        for (int i = 0; i < NT; ++i) pin_frag(af[i]);
#pragma unroll
        for (int j = 0; j < MT; ++j) pin_frag(bf[j]);
    };
    constexpr int NTH = NT * MT, NR = NT + MT;
    // MFMAs [T0, T1) of a group on fragments (af, bf).  The NR reads of the next fragment set ride on the head of the group (RLIM > 0):
    // read r is issued behind MFMA r, so the last of them has the rest of the group - 17 MFMAs - to come back.
    auto mma_group = [&](auto t0c, auto t1c, auto rlimc, const u32x4 (&af)[NT], const u32x4 (&bf)[MT], u32x4 (&naf)[NT], u32x4 (&nbf)[MT],
                         unsigned wsb, unsigned xsb) {
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value, RLIM = decltype(rlimc)::value;
        static_for<T1 - T0>([&](auto tt) {
            constexpr int t = T0 + decltype(tt)::value;
            MmaAsm<DT>::run(af[t / MT], bf[t % MT], acc[t / MT][t % MT]);
            if constexpr (RLIM > 0) {
                static_for<NR>([&](auto rc) {
                    constexpr int r = decltype(rc)::value, tr = r;
                    static_assert(NR <= RLIM, "every read in front of MFMA RLIM");
                    if constexpr (tr == t) frag_read(naf, nbf, wsb, xsb, rc);
                });
            }
        });
    };

    const bool late = a.stagger && wave >= 4;
    constexpr bool PERSIST = !FUSEC;
    int epi_frame = 0, epi_h0 = 0;
    bool epi_more = false;
#pragma nounroll
    for (;;) {
    wait_vmcnt<0>();                                   // this unit's first operands (and the previous unit's output stores) are through
    __builtin_amdgcn_s_barrier();                      // ... for every wave; nobody is in the previous unit's epilogue patch any more
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_live(acc[i][j]); }
    mfma_operands_settled();
    {
        constexpr int H1 = 3 * NTH / 4;                                    // MFMAs of a LATE wave's first group in front of its barrier
        static_assert(NR + RW + PPS <= NTH, "a DMA piece or a read per MFMA of the second group");
        int stoff = 0;                                                     // ring slot of the current step (bytes): (s % NSLOT) * WSTAGE
        int bufc = bstart, bufn = pbytes - bstart;                         // patch buffers of this slab / the next one
        {
            const unsigned wsb = w_addr(0, 0), xsb = x_addr(bufc, 0, 0);
            static_for<NR>([&](auto r) { frag_read(a0, b0, wsb, xsb, r); });
        }
        // (no LDS read stays pending across a tap body's end, and the slab loop is neither unrolled nor peeled: af_conv.hip's K loop
        //  has the reasons)
        wait_lgkmcnt<0>();
        pin_half(a0, b0);
#pragma nounroll
        for (int slab = 0; slab < nslabs; ++slab) {
            const bool lastslab = slab + 1 == nslabs;
            const int wsl_c = w_slab_offset(slab), wsl_n = w_slab_offset(lastslab ? slab : slab + 1);
            // the next slab's patch goes into the other buffer (where slab - 1 lived), one piece per K-step.  (Piece 8 - patch rows
            // 512.. - is issued behind the first fragment reads of the next slab: tap 0 reads rows < 512 only.)
            bool in1;
            const int ps1 = patch_slab_offset(lastslab ? slab : slab + 1, in1);
            const bool pp = !lastslab;
            static_for<NTAPS>([&](auto tapc) {
                constexpr int tap = tapc, ntap = (tap + 1) % NTAPS;
                constexpr int t2 = (tap + NSLOT) % NTAPS;                  // the tap of stage s + NSLOT, which refills this step's slot
                // row shift of a tap: spatial (dh, dw) -> dh * WP + dw; temporal dt -> dt * P.  (The shifts pass through an empty
                // asm: hipcc otherwise hoists the fragment addresses of all unrolled bodies out of the slab loop and spills to keep them.)
                int sh = TEMPORAL ? tap * a.R : (tap / 3) * WP + tap % 3, nsh = TEMPORAL ? ntap * a.R : (ntap / 3) * WP + ntap % 3;
                asm volatile("" : "+s"(sh), "+s"(nsh));
                // first group: MFMA(a0, b0) [stage s, k-half 0] while the fragments of k-half 1 come in
                const unsigned wsb1 = w_addr(stoff, 1), xsb1 = x_addr(bufc, sh, 1);
                // (table words of the patch pieces this step carries: read here, waited for with the group's fragment reads)
                unsigned pw[PPS];
                if constexpr (POFF_LDS && tap < PSTEPS) {
                    static_for<PPS>([&](auto sl) {
                        constexpr int piece = tap * PPS + decltype(sl)::value;
                        if constexpr (piece < MAXP) pw[sl] = lds_read4_uncounted<piece * 256>(tabl);
                    });
                }
                // (the last K-step runs the same instruction stream with its barrier, look-ahead reads and DMA switched off: a
                //  separate tail would meet this path in 112 accumulator phis)
                const bool laststep = tap == NTAPS - 1 && lastslab;
                // One instruction stream for both kinds of wave: a LATE wave takes the step's barrier behind MFMA H1, the others
                // behind the whole group.  In front of it: this wave's reads of step s are back, its pieces of stage s + 1 (and any
                // patch piece) have landed; behind it that holds for every wave, and ring slot `stoff` / the previous slab's patch
                // are no longer read.
                mma_group(IC<0>{}, IC<H1>{}, IC<H1>{}, a0, b0, a1, b1, wsb1, xsb1);
                // What must have landed: stage s + 1.  Two slots: it is the only stage in flight (vmcnt 0).  Three slots: stage s + 2
                // was issued behind it in the previous step - a patch piece first, then RW weight pieces - and stays in flight
                // (vmcnt RW), unless that step was past the end of the refills (the last slab's last NSLOT - 1 taps wait for everything).
                auto wait_stage = [&]() {
                    if (AF_DBG(1)) return;
                    if (NSLOT == 2 || (tap - 1 + NSLOT >= NTAPS && lastslab)) wait_vmcnt<0>();
                    else wait_vmcnt<RW>();
                };
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
                // second group: MFMA(a1, b1) [stage s, k-half 1] while the fragments of stage s + 1, k-half 0 come in; then
                // the DMA pieces of stage s + 2 (and one piece of the next slab's patch), MPG MFMAs apart
                const int stnext = NSLOT == 2 ? (stoff ^ WSTAGE) : (stoff == (NSLOT - 1) * WSTAGE ? 0 : stoff + WSTAGE);
                const unsigned wsb0 = w_addr(stnext, 0), xsb0 = x_addr(tap == NTAPS - 1 ? bufn : bufc, nsh, 0);
                const bool refill = (tap + NSLOT < NTAPS || !lastslab) && !AF_DBG(4);
                const int wsoff = (tap + NSLOT < NTAPS ? wsl_c : wsl_n) + t2 * tap_bytes;
                // behind the MFMAs of the group, one each (group2_slot: piece, read, read, piece, ...): PPS pieces of the next slab's
                // patch, the RW weight pieces, the NR look-ahead reads (behind the last step: unused reads of valid LDS)
                static_for<NTH>([&](auto tc) {
                    constexpr int t = tc, slot = group2_slot(t, RW + PPS, NR);
                    MmaAsm<DT>::run(a1[t / MT], b1[t % MT], acc[t / MT][t % MT]);
                    if constexpr (slot != kNoSlot && slot < 0) frag_read(a0, b0, wsb0, xsb0, IC<-1 - slot>{});
                    else if constexpr (slot != kNoSlot && slot < PPS) {
                        constexpr int piece = tap * PPS + slot;
                        if constexpr (piece < MAXP && tap < PSTEPS) {
                            if (pp && !AF_DBG(16)) {
                                if constexpr (POFF_LDS) { asm volatile("" : "+v"(pw[slot])); issue_patch_piece(bufn, ps1, in1, piece, table_offset(pw[slot])); }
                                else issue_patch_piece(bufn, ps1, in1, piece, poff[piece]);
                            }
                        }
                    } else if constexpr (slot != kNoSlot) { if (refill) issue_w_piece(stoff, wsoff, slot - PPS); }
                });
                stoff = stnext;
                wait_lgkmcnt<0>();                     // a0 / b0 of the next step (read under this group) are back
                pin_half(a0, b0);
            });
            const int tb = bufc; bufc = bufn; bufn = tb;
        }
    }
    AF_STAMP(1);
    AF_STAMP(2);
    mfma_drain();                                      // the accumulators are read by ordinary vector code from here on

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is done with the patches and the ring
    if constexpr (FUSEC) break;
    // (the epilogue's lane-derived values start from an opaque copy of the lane id: nothing of its address arithmetic can be
    //  computed - and kept in registers - across the K loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int frow_e = lane_e & 15, fg_e = lane_e >> 4;
    // BN parameters: loads hipcc does not count, issued BEFORE the next unit's DMA - its own wait for them would be vmcnt(0),
    // i.e. for the DMA issued behind them as well - and waited for with the number of younger operations of this wave
    f32x4 sc[NT], sf[NT];
    {
        uint4 scr[NT], sfr[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            scr[i] = gload16_uncounted(a.scale + wn * 64 + i * 16 + fg_e * 4);
            sfr[i] = gload16_uncounted(a.shift + wn * 64 + i * 16 + fg_e * 4);
        }
        // ---- the next unit of this workgroup: geometry, then its first operands into the ring and the patch buffer the epilogue
        // does not use (the transposition patches live at the start of buffer 0)
        const int e_frame = frame, e_h0 = h0;
        const int next = unit + (int)gridDim.x;
        const bool has_next = PERSIST && next < units;
        int younger = 0;
        if (has_next) {
            unit = next;
            frame = unit / a.upf; h0 = (unit - frame * a.upf) * a.R; tclip = frame % a.T;
            xdesc = unit_desc();
            issue_unit_head(pbytes);
            bstart = pbytes;
            int np = (NP - wave + 7) >> 3;
            younger = NSLOT * RW + (np < MAXP ? np : MAXP);
        }
        wait_vmcnt_dyn<NSLOT * RW + MAXP>(younger);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            u32x4 t0 = __builtin_bit_cast(u32x4, scr[i]), t1 = __builtin_bit_cast(u32x4, sfr[i]);
            pin_frag(t0); pin_frag(t1);
            sc[i] = __builtin_bit_cast(f32x4, t0); sf[i] = __builtin_bit_cast(f32x4, t1);
        }
        epi_frame = e_frame; epi_h0 = e_h0; epi_more = has_next;
    }
    const int rr = lane_e >> 3, cc = (lane_e & 7) * 8;
    if (!FUSEC) {
        // ---- epilogue: BN + ReLU + the one rounding, transposed through a per-wave patch (the patches / ring are dead),
        // whole 128-byte rows out.  Padded position p = r * WP + c -> output pixel (h0 + r, c - 1); halo columns dropped.
        OT* patch = reinterpret_cast<OT*>(smem) + wave * (16 * PROW);
        char* obase = TEMPORAL ? a.out + (((long long)epi_frame * a.T) * HW + epi_h0) * a.out_ld * 2 + wn * 128
                               : a.out + (((long long)epi_frame * a.H + epi_h0) * a.W) * a.out_ld * 2 + wn * 128;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                f32x4 v = acc[i][j] * sc[i] + sf[i];
                if (a.relu) { v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]); }
                Vec4<DT>::store(reinterpret_cast<char*>(patch + frow_e * PROW + i * 16 + fg_e * 4), v);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + rr;
                const int p = (wm * MT + j) * 16 + row;
                const int r = (int)(((float)p + 0.5f) * a.inv_wp), c = p - r * (TEMPORAL ? a.R : WP);
                const u32x4 o = *reinterpret_cast<const u32x4*>(patch + row * PROW + cc);
                if (TEMPORAL) {                          // position p = (frame r = p / P, pixel c = p % P): all real but a ragged last chunk
                    if (r < a.T && epi_h0 + c < HW)
                        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(obase + ((long long)r * HW + c) * a.out_ld * 2 + cc * 2));
                } else if (r < a.R && c >= 1 && c <= a.W && epi_h0 + r < a.H)
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(obase + (long long)(r * a.W + c - 1) * a.out_ld * 2 + cc * 2));
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (epi_more) continue;
        AF_STAMP(3); AF_STAMP(7);
        AF_STAMP_FLUSH;
        return;
    }
    }                                                  // (unit loop; the fused `c` instantiations leave it behind their one unit's K loop)
    f32x4 sc[NT], sf[NT];
    const int frow_e = frow, fg_e = fg;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        sc[i] = *reinterpret_cast<const f32x4*>(a.scale + wn * 64 + i * 16 + fg_e * 4);
        sf[i] = *reinterpret_cast<const f32x4*>(a.shift + wn * 64 + i * 16 + fg_e * 4);
    }
    const int rr = lane >> 3, cc = (lane & 7) * 8;

    // ---- fused `c` conv.  The band's b output (BN + ReLU, rounded once) becomes the B operand of a 1x1x1 convolution without
    // leaving the CU: T[slab = wn][position p][64 channels] in LDS, swizzled like the patches (halo columns hold garbage that
    // only feeds dropped outputs).  The c weights go global -> registers (16 bytes per lane = one A fragment, a K-step ahead):
    // no ring, no barrier - every wave runs its 64 output channels x 112 positions on its own, Cout2 / BN passes.
    constexpr int MPAD = WM * MT * 16;
    char* T = reinterpret_cast<char*>(smem);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int p = (wm * MT + j) * 16 + frow_e;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4 v = acc[i][j] * sc[i] + sf[i];
            if (a.relu) { v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]); }
            const int chunk = i * 2 + (fg_e >> 1);
            Vec4<DT>::store(T + (wn * MPAD + p) * 128 + ((chunk ^ (p & 7)) << 4) + (fg_e & 1) * 8, v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // T complete
    constexpr int SROW = 64 + 4;                       // fp32 staging row stride (floats)
    float* stg = reinterpret_cast<float*>(T + WN * MPAD * 128) + wave * (16 * SROW);
    const long long pix0 = ((long long)frame * a.H + h0) * a.W;
    const int passes = a.Cout2 / BN;
    const int Cmid = BN;                               // K of the c conv = the b conv's output channels
    for (int pass = 0; pass < passes; ++pass) {
        const int ch0 = pass * BN + wn * 64;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A fragments (c weights) straight from global memory, two k-halves ahead of their MFMAs (3 register sets)
        uint4 an[3][NT];
        auto load_a = [&](uint4 (&dst)[NT], int h) {            // half-step h = (K-step h >> 1, k-half h & 1)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                dst[i] = *reinterpret_cast<const uint4*>(a.w2 + ((long long)(ch0 + i * 16 + frow_e) * Cmid + h * 32 + fg_e * 8) * 2);
        };
        load_a(an[0], 0);
        load_a(an[1], 1);
#pragma unroll
        for (int h = 0; h < 2 * WN; ++h) {
            if (h + 2 < 2 * WN) load_a(an[(h + 2) % 3], h + 2);
            uint4 bf[MT];
            const char* xsb = T + ((h >> 1) * MPAD + wm * MT * 16 + frow_e) * 128 + (((((h & 1) << 2) + fg_e) ^ (frow_e & 7)) << 4);
#pragma unroll
            for (int j = 0; j < MT; ++j) bf[j] = *reinterpret_cast<const uint4*>(xsb + j * (16 * 128));
#pragma unroll
            for (int t = 0; t < NTH; ++t) Mma<DT>::run(an[h % 3][t / MT], bf[t % MT], acc[t / MT][t % MT]);
        }
        // epilogue of the pass: BN in registers -> fp32 staging rows -> (+ residual) -> ReLU -> the one rounding -> 16-byte stores
        f32x4 sc2[NT], sf2[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            sc2[i] = *reinterpret_cast<const f32x4*>(a.scale2 + ch0 + i * 16 + fg_e * 4);
            sf2[i] = *reinterpret_cast<const f32x4*>(a.shift2 + ch0 + i * 16 + fg_e * 4);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
                *reinterpret_cast<f32x4*>(stg + frow_e * SROW + i * 16 + fg_e * 4) = acc[i][j] * sc2[i] + sf2[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + rr;
                const int p = (wm * MT + j) * 16 + row;
                const int r = (int)(((float)p + 0.5f) * a.inv_wp), c = p - r * WP;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * SROW + cc);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * SROW + cc + 4);
                if (r < a.R && c >= 1 && c <= a.W && h0 + r < a.H) {
                    const long long pix = pix0 + r * a.W + c - 1;
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (a.res) {
                        const uint4 rraw = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.res + (pix * a.Cout2 + ch0 + cc) * 2)));
                        const OT* re = reinterpret_cast<const OT*>(&rraw);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += E::to_f32(re[e]);
                    }
                    uint4 o;
                    OT* oe = reinterpret_cast<OT*>(&o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) oe[e] = E::from_f32(a.relu2 ? relu_f(v[e]) : v[e]);
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(a.out + (pix * a.out_ld + ch0 + cc) * 2));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int DT, int WN, int WM, bool FUSEC, int MT, int MAXP, int NSLOT, bool TEMPORAL = false>
static int launch133g_n(const C133GArgs& a, hipStream_t stream) {
    if (a.prows > 64 * MAXP) return set_error(AF_ERR_ARG, "conv133g: %d patch rows for %d pieces per wave", a.prows, MAXP);
    int lds = 2 * a.prows * 128 + NSLOT * WN * 64 * 128 + (WN == 2 ? a.prows * 4 : 0);      // patches, ring, the patch-row offset table (128 channels)
    const int lds_c = WN * (WM * MT * 16) * 128 + 8 * 16 * (64 + 4) * 4;      // T + the per-wave fp32 staging rows
    if (FUSEC && lds_c > lds) lds = lds_c;
    const int lds_e = 8 * 16 * (64 + 8) * 2;                                   // the epilogue's per-wave patches
    if (lds < lds_e) lds = lds_e;
    if (lds > 160 * 1024) return set_error(AF_ERR_ARG, "conv133g: %d bytes of LDS needed", lds);
    AF_SET_MAX_LDS((&conv133g_kernel<DT, WN, WM, MT, FUSEC, MAXP, NSLOT, TEMPORAL>), 160 * 1024, "conv133g");
    // persistent workgroups (one per CU: the kernel uses all of LDS): unit blockIdx.x, + gridDim.x, ...
    const int units = a.frames * a.upf, cus = device_cus();
    static const int persist = [] { const char* e = getenv("AF_G_PERSIST"); return e ? atoi(e) : 1; }();   // 0: one unit per workgroup (A/B runs)
    const int grid = (FUSEC || !persist || units <= cus || cus <= 0) ? units : cus;
    hipLaunchKernelGGL((conv133g_kernel<DT, WN, WM, MT, FUSEC, MAXP, NSLOT, TEMPORAL>), dim3(grid), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv133g_kernel");
    return AF_OK;
}

// three ring slots (two K-steps of weights in flight) where the two patch buffers leave room for them (s4: 2 x 31 KB + 3 x 32 KB)
template <int DT, int WN, int WM, bool FUSEC, int MT = 7, int MAXP = 9>
static int launch133g(const C133GArgs& a, hipStream_t stream) {
    if (!FUSEC && 2 * a.prows * 128 + 3 * WN * 64 * 128 + (WN == 2 ? a.prows * 4 : 0) <= 160 * 1024) return launch133g_n<DT, WN, WM, FUSEC, MT, MAXP, FUSEC ? 2 : 3>(a, stream);
    return launch133g_n<DT, WN, WM, FUSEC, MT, MAXP, 2>(a, stream);
}

// rows of a frame one unit covers (0: the layer does not take this path)
static int conv133g_rows(const af_conv_desc* d) {
    if (d->dtype == AF_F32 || d->tpool) return 0;
    if ((d->kt != 1 && d->kt != 3) || d->kh != 3 || d->kw != 3 || d->st != 1 || d->sh != 1 || d->sw != 1) return 0;
    if (d->pt != d->kt / 2 || d->ph != 1 || d->pw != 1) return 0;
    // 64 output channels only for the 3x3x3 case (the 1x3x3 64 -> 64 layers of s2 keep their weights-in-registers kernel)
    if (d->cin % 64 != 0 || (d->cout != 128 && d->cout != 256 && !(d->cout == 64 && d->kt == 3))) return 0;
    const int mpad = d->cout == 256 ? 224 : d->cout == 128 ? 448 : 512;   // positions of a unit: WM x MT m-tiles
    const int wp = d->w + 1;
    int r = mpad / wp;
    if (r > d->h) r = d->h;
    auto patch_rows = [&](int rows) { return ((rows + 2) * wp + 1 + 7) & ~7; };
    const int tab = d->cout == 128 ? 4 : 0;                   // bytes per patch row of the offset table (128 channels)
    while (r >= 7 && (patch_rows(r) > 576 || 2 * patch_rows(r) * 128 + 2 * d->cout * 128 + patch_rows(r) * tab > 160 * 1024)) --r;   // two patches + the ring (+ the table)
    if (r < 7) return 0;                                       // bands of >= 7 rows: the halo rows stay <= 2/7 of the patch
    int upf = (d->h + r - 1) / r;
    r = (d->h + upf - 1) / upf;                                // even bands
    upf = (d->h + r - 1) / r;
    // worth it when the units fill the chip and most of a unit's positions are real
    const long long units = (long long)d->n * d->t * upf;
    if (units < 192 || units > 0x7fffffffLL) return 0;
    if ((double)d->h * d->w / ((double)upf * mpad) < 0.6) return 0;
    if ((long long)(d->h + 2) * d->w * d->cin * 2 * (d->kt + 1) >= (1LL << 31)) return 0;
    return r;
}

bool conv133g_applies(const af_conv_desc* d, const void* residual, int out_ld) {
    return !residual && (out_ld == 0 || out_ld % 8 == 0) && conv133g_rows(d) != 0;
}

static void fill133g(C133GArgs& a, const af_conv_desc* d, const void* in, const void* w_packed, const float* scale,
                     const float* shift) {
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift;
    a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.frames = d->n * d->t;
    a.T = d->t; a.kt = d->kt;
    a.R = conv133g_rows(d); a.upf = (d->h + a.R - 1) / a.R; a.WP = d->w + 1;
    a.prows = ((a.R + 2) * a.WP + 1 + 7) & ~7;
    a.kslabs = d->cin / 64; a.relu = d->relu;
    a.inv_wp = 1.0f / (float)a.WP;
    a.w2 = nullptr; a.scale2 = a.shift2 = nullptr; a.res = nullptr; a.Cout2 = 0; a.relu2 = 0;
    const char* es = getenv("AF_G_STAGGER");
    a.stagger = es ? atoi(es) : 1;
#ifdef AF_STAMPS
    const char* ep = getenv("AF_STAMP_PTR");
    a.stamps = ep ? (unsigned long long*)strtoull(ep, nullptr, 0) : nullptr;
    const char* ed = getenv("AF_G_DBG");
    a.dbg = ed ? atoi(ed) : 0;
#endif
}

// ---- TEMPORAL mode: 3x1x1 / stride 1 / pad (1,0,0) convs into 128 / 256 channels (the `a` convs of s3 / s4) on the same kernel.
// P = pixels per unit so that T x P is the kernel's position count (224 for 256 channels, 448 for 128); 0: not this path.
static int conv311g_pixels(const af_conv_desc* d) {
    if (d->dtype == AF_F32 || d->tpool) return 0;
    if (d->kt != 3 || d->kh != 1 || d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1) return 0;
    if (d->pt != 1 || d->ph != 0 || d->pw != 0) return 0;
    if (d->cin % 64 != 0 || (d->cout != 128 && d->cout != 256)) return 0;
    const int mpad = d->cout == 256 ? 224 : 448;
    if (d->t < 4 || mpad % d->t != 0) return 0;
    const int P = mpad / d->t;
    const int prows = ((d->t + 2) * P + 7) & ~7;
    // (patch rows: 64 per DMA piece of a wave - 5 pieces in the 256-channel instantiation, 9 in the 128-channel one)
    if (P < 7 || prows > (d->cout == 256 ? 320 : 576) || 2 * prows * 128 + 2 * d->cout * 128 + (d->cout == 128 ? prows * 4 : 0) > 160 * 1024) return 0;
    const long long hw = (long long)d->h * d->w, chunks = (hw + P - 1) / P, units = (long long)d->n * chunks;
    if (units < 192 || units > 0x7fffffffLL) return 0;
    if ((double)hw / ((double)chunks * P) < 0.6) return 0;                       // most positions of a unit are real
    if ((long long)(d->t + 2) * hw * d->cin * 2 >= (1LL << 31)) return 0;          // 32-bit offsets inside a clip (+ the halo frames)
    return P;
}

bool conv311g_applies(const af_conv_desc* d, const void* residual, int out_ld) {
    static const int enabled = [] { const char* e = getenv("AF_T311G"); return e ? atoi(e) : 1; }();
    return enabled && !residual && (out_ld == 0 || out_ld % 8 == 0) && conv311g_pixels(d) != 0;
}

int conv311g_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                 void* out, int out_ld, hipStream_t stream) {
    C133GArgs a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift;
    a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.frames = d->n;   // a "frame" of the unit decomposition is a clip
    a.T = d->t; a.kt = 1;                                                         // (the temporal taps are the kernel's taps)
    a.R = conv311g_pixels(d); a.upf = (d->h * d->w + a.R - 1) / a.R; a.WP = a.R;
    a.prows = ((d->t + 2) * a.R + 7) & ~7;
    a.kslabs = d->cin / 64; a.relu = d->relu;
    a.inv_wp = 1.0f / (float)a.R;
    a.w2 = nullptr; a.scale2 = a.shift2 = nullptr; a.res = nullptr; a.Cout2 = 0; a.relu2 = 0;
    const char* es = getenv("AF_G_STAGGER");
    a.stagger = es ? atoi(es) : 1;
#ifdef AF_STAMPS
    const char* ep = getenv("AF_STAMP_PTR");
    a.stamps = ep ? (unsigned long long*)strtoull(ep, nullptr, 0) : nullptr;
    const char* ed = getenv("AF_G_DBG");
    a.dbg = ed ? atoi(ed) : 0;
#endif
    a.out = (char*)out; a.out_ld = out_ld ? out_ld : d->cout;
    const bool three = 2 * a.prows * 128 + 3 * d->cout * 128 + (d->cout == 128 ? a.prows * 4 : 0) <= 160 * 1024;     // 256 channels, P = 14: 2 x 32 KB + 3 x 32 KB = all of LDS
    if (d->cout == 256) {
        if (a.prows > 64 * 5) return set_error(AF_ERR_ARG, "conv311g: %d patch rows", a.prows);
        if (three) return d->dtype == AF_BF16 ? launch133g_n<AF_BF16, 4, 2, false, 7, 5, 3, true>(a, stream) : launch133g_n<AF_F16, 4, 2, false, 7, 5, 3, true>(a, stream);
        return d->dtype == AF_BF16 ? launch133g_n<AF_BF16, 4, 2, false, 7, 5, 2, true>(a, stream) : launch133g_n<AF_F16, 4, 2, false, 7, 5, 2, true>(a, stream);
    }
    if (three) return d->dtype == AF_BF16 ? launch133g_n<AF_BF16, 2, 4, false, 7, 9, 3, true>(a, stream) : launch133g_n<AF_F16, 2, 4, false, 7, 9, 3, true>(a, stream);
    return d->dtype == AF_BF16 ? launch133g_n<AF_BF16, 2, 4, false, 7, 9, 2, true>(a, stream) : launch133g_n<AF_F16, 2, 4, false, 7, 9, 2, true>(a, stream);
}

// b (1x3x3) + c (1x1x1, + residual, + ReLU) of a bottleneck as one launch: true iff `db` takes the frame-resident path and
// `dc` is a plain 1x1x1 convolution over db's output whose channel count is a multiple of db's
bool conv133g_fused_applies(const af_conv_desc* db, const af_conv_desc* dc, int out_ld) {
    if (!db || !dc || !conv133g_applies(db, nullptr, 0) || db->kt != 1 || db->cout == 64) return false;
    if (dc->dtype != db->dtype || dc->tpool || dc->kt != 1 || dc->kh != 1 || dc->kw != 1) return false;
    if (dc->st != 1 || dc->sh != 1 || dc->sw != 1 || dc->pt || dc->ph || dc->pw) return false;
    if (dc->n != db->n || dc->t != db->to || dc->h != db->ho || dc->w != db->wo || dc->cin != db->cout) return false;
    if (dc->cout % db->cout != 0) return false;
    return out_ld == 0 || (out_ld >= dc->cout && out_ld % 8 == 0);
}

int conv133g_fused_run(const af_conv_desc* db, const void* in, const void* wb, const float* scale_b, const float* shift_b,
                       const af_conv_desc* dc, const void* wc, const float* scale_c, const float* shift_c, const void* residual,
                       void* out, int out_ld, hipStream_t stream) {
    C133GArgs a;
    fill133g(a, db, in, wb, scale_b, shift_b);
    a.out = (char*)out; a.out_ld = out_ld ? out_ld : dc->cout;
    a.w2 = (const char*)wc; a.scale2 = scale_c; a.shift2 = shift_c; a.res = (const char*)residual; a.Cout2 = dc->cout; a.relu2 = dc->relu;
    if (db->cout == 256) return db->dtype == AF_BF16 ? launch133g<AF_BF16, 4, 2, true, 7, 5>(a, stream) : launch133g<AF_F16, 4, 2, true, 7, 5>(a, stream);
    return db->dtype == AF_BF16 ? launch133g<AF_BF16, 2, 4, true>(a, stream) : launch133g<AF_F16, 2, 4, true>(a, stream);
}

int conv133g_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                 void* out, int out_ld, hipStream_t stream) {
    C133GArgs a;
    fill133g(a, d, in, w_packed, scale, shift);
    a.out = (char*)out; a.out_ld = out_ld ? out_ld : d->cout;
    if (d->cout == 256) return d->dtype == AF_BF16 ? launch133g<AF_BF16, 4, 2, false, 7, 5>(a, stream) : launch133g<AF_F16, 4, 2, false, 7, 5>(a, stream);
    if (d->cout == 64) return d->dtype == AF_BF16 ? launch133g<AF_BF16, 1, 8, false, 4>(a, stream) : launch133g<AF_F16, 1, 8, false, 4>(a, stream);
    // pieces per wave: 256 channels: <= 224 + 2 WP + 1 <= 296 patch rows (5); 128 channels: the s3 band of 472 rows takes 8
    if (a.prows <= 512) return d->dtype == AF_BF16 ? launch133g<AF_BF16, 2, 4, false, 7, 8>(a, stream) : launch133g<AF_F16, 2, 4, false, 7, 8>(a, stream);
    return d->dtype == AF_BF16 ? launch133g<AF_BF16, 2, 4, false>(a, stream) : launch133g<AF_F16, 2, 4, false>(a, stream);
}

}  // namespace af

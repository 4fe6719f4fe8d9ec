// The s2 -> s3 boundary as ONE launch (16-bit operands): c of the last s2 block, pathway0_pool, a of the first s3 block
//     x    = relu( bn_c(conv1x1x1_c(b)) + res )        the end of s2's last ResBlock   (resnet_helper.py:304-325, 438-444)
//     xp   = max( x[2t'], x[2t'+1] )                   pathway0_pool, MaxPool3d [2,1,1] (video_model_builder.py:566-569)
//     aout = relu( bn_a(conv3x1x1_a(xp)) )             the first conv of s3's block 0  (resnet_helper.py:267-281)
// b: [N][T][H][W][64], res: [N][T][H][W][C] (C = 256), aout: [N][T/2][H][W][128].
//
// As two launches the 411-MB pooled trunk was written by the c conv, read back (with its temporal halo) by the a conv and read
// a third time - one position in four - by the stage's projection shortcut.  Here the tile of conv_ca (all T frames of 8
// pixels: the temporal halo is inside the tile) is kept, the c conv's slab is pooled in registers (the two frames of a pixel
// are the two halves of an MFMA row group: one DPP row rotation) and written over the residual image it came from, and the
// three temporal taps run on it there.  The trunk leaves the chip only where somebody else reads it: with x_sub == 2 the
// even (h, w) positions, packed as [N][T/2][H/2][W/2][C] - what a 1x1x1 conv of stride (1,2,2) touches.  1.33 GB per
// launch instead of 2.05 (c, pooled) + 0.62 (a).
//
// Tile = 2 rows x 4 columns x 32 frames (every tile owns two even positions: each wave's number of trunk stores is a
// constant, which the counted s_waitcnt need).  LDS: three residual images of 280 rows (fetched two stages ahead), the a
// weights of the slab as two k-half slots [3 taps][128][32 channels] of 24 KB - two whole 48-KB stages do not fit next to the
// images - refilled as soon as their half is consumed.  vmcnt is an in-order counter: a wave waiting for weights issued
// late would drain the residual DMA issued before them, so waves 0-3 issue (and wait for) the images and the trunk stores,
// waves 4-7 the weights.
//
// MEASURED (round 3, B = 16, bf16, one box): 0.57 ms against 0.25 + 0.19 ms for the two launches it replaces; 0.46 ms once ReLU
// and the pair max were single v_maximum3_f32 instructions - a tie end to end, so the engine keeps the two launches unless
// AF_FUSE_CPA=1.  The first measurement:  With parts switched off (timing builds): no c conv / epilogue 0.39 ms, no a taps
// 0.51, neither and no fetches 0.32 (the barrier / DMA-latency skeleton alone is slower than the 0.27 ms the traffic would take
// at conv_ca's rate); weight or residual DMA lanes out of range -0.02 / -0.03 ms (not ingest-bound).  A stage here has 50 KB
// of HBM traffic against conv_ca's 80 KB but more instructions (SQ counters: 523 vector + 263 scalar per wave and stage against
// 356 + 206) and five barriers: the stage is bound by its own critical path, not by HBM.
#include "af_common.h"

namespace af {

struct CPAArgs {
    const char* inb;     // [N][T][H][W][64]
    const char* wc;      // packed [C][64]
    const float* scale_c;
    const float* shift_c;
    const char* res;     // [N][T][H][W][C]
    char* outx;          // sub: [N][T/2][H/2][W/2][C]; full: [N][T/2][H][W][C]
    const char* wa;      // packed [128][3][C]
    const float* scale_a;
    const float* shift_a;
    char* outa;          // [N][T/2][H][W][128]
    int T, H, W, C, kslabs;
    int wq, per_clip, tiles;     // 4-column patches per row pair, tiles per clip, tiles
    int sub;                     // 1: only the even (h, w) positions of the pooled trunk are stored, packed
};

constexpr int kCpaRows = 280;                          // image rows of a slot: 8 (halo t' = -1) + 256 + 16
constexpr int kCpaImg = kCpaRows * 128;
constexpr int kCpaW = 3 * 128 * 64;                    // one k-half slot of a weights
static long long conv_cpa_lds_bytes(int C) { return 2LL * kCpaW + 3LL * kCpaImg + (2LL * C + 256) * 4; }

template <int DT>
__global__ __launch_bounds__(512) void conv_cpa_kernel(const CPAArgs a) {
    typedef Elem<DT> E;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int TN = 4, TM = 2, T = 32, TO = 16;

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_w = wave >= 4;                       // weight producer (else image producer)
    const int frow = lane & 15, fg = lane >> 4;
    char* sm = reinterpret_cast<char*>(smem);
    char* img0 = sm + 2 * kCpaW;
    float* bnp = reinterpret_cast<float*>(img0 + 3 * kCpaImg);     // scale_c[C] shift_c[C] scale_a[128] shift_a[128]
    const int HW = a.H * a.W;

    for (int i = tid; i < a.C; i += 512) { bnp[i] = a.scale_c[i]; bnp[a.C + i] = a.shift_c[i]; }
    if (tid < 128) { bnp[2 * a.C + tid] = a.scale_a[tid]; bnp[2 * a.C + 128 + tid] = a.shift_a[tid]; }
    // the temporal padding frames of the pooled image (rows 0..7 and 272..279 of every slot) are never written again
    if (tid < 3 * 16 * 8) {
        const int s = tid / 128, r = (tid >> 3) & 15, ck = tid & 7;
        *reinterpret_cast<uint4*>(img0 + s * kCpaImg + ((r < 8 ? r : 264 + r) * 128) + ck * 16) = uint4{0u, 0u, 0u, 0u};
    }

    // ---- producer state
    // image piece (waves 0-3, 8 per stage): 8 rows = the 8 pixels of frame (wave + 4 i); lane -> (pixel, 16-byte chunk)
    const int drow = lane >> 3, chunk = (lane & 7) ^ drow;
    const unsigned xoff0 = (unsigned)((((long long)(wave & 3) * HW + (drow >> 2) * a.W + (drow & 3)) * a.C) * 2 + chunk * 16);
    const unsigned xstep = (unsigned)(4LL * HW * a.C * 2);
    // weight piece (waves 4-7, 6 per k-half): 16 rows of 64 bytes; lane -> (row, chunk)
    const long long Kw = 3LL * a.C;
    const i32x4 wdesc = make_desc(a.wa);
    unsigned woff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = ((wave & 3) + 4 * i) * 16 + (lane >> 2), dt = row >> 7, co = row & 127;
        // 64-byte rows: ds_read_b128 reads by 16-lane groups {fg, rows 0-3 + 12-15} + {fg + 1, rows 4-11} on a 256-byte bank row;
        // rows 8..15 of every 16 hold their chunks XOR 2, which makes the 16 slots of a group distinct (linear: 2-way)
        woff[i] = (unsigned)((co * Kw + (long long)dt * a.C) * 2 + ((lane & 3) ^ ((lane >> 4) & 2)) * 16);
    }
    const long long clipx = (long long)T * HW * a.C * 2, clipb = (long long)T * HW * 64 * 2;

    const int my_tiles = (a.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * a.kslabs;
    auto stage_tile = [&](int g) { return (int)blockIdx.x + (g / a.kslabs) * (int)gridDim.x; };
    auto tile_origin = [&](int tile, int& n, int& h0, int& w0) {
        n = tile / a.per_clip;
        const int r = tile - n * a.per_clip, hp = r / a.wq;
        h0 = 2 * hp; w0 = 4 * (r - hp * a.wq);
    };
    auto issue_weights = [&](int g, int kh) {                        // k-half kh of stage g's a weights -> slot kh
        const unsigned base = lds0 + kh * kCpaW + (wave & 3) * 1024;
        const int soff = __builtin_amdgcn_readfirstlane(((g % a.kslabs) * 64 + kh * 32) * 2);
#pragma unroll
        for (int i = 0; i < 6; ++i) blds16_m0(woff[i], wdesc, soff, base + i * 4096);
    };
    auto issue_image = [&](int g) {                                  // residual slab of stage g -> rows 8..263 of slot g % 3
        int n, h0, w0;
        tile_origin(stage_tile(g), n, h0, w0);
        const i32x4 xdesc = make_desc(a.res + n * clipx + ((long long)h0 * a.W + w0) * a.C * 2);
        const unsigned base = lds0 + 2 * kCpaW + (g % 3) * kCpaImg + (8 + (wave & 3) * 8) * 128;
        const int soff = __builtin_amdgcn_readfirstlane((g % a.kslabs) * 128);
#pragma unroll
        for (int i = 0; i < 8; ++i) blds16_nt_m0(xoff0 + i * xstep, xdesc, soff, base + i * (32 * 128));
    };
    // c weights of a stage (4 channel tiles x 2 k-halves) and the b fragments of a tile (2 row tiles x 2 k-halves):
    // global -> registers, a stage / a tile ahead
    uint4 wcur[TN][2], bcur[TM][2], wnext[TN][2], bnext[TM][2];
    auto load_wc = [&](uint4 (&dst)[TN][2], int kc) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                dst[i][kk] = gload16_uncounted(a.wc + ((long long)(kc * 64 + i * 16 + frow) * 64 + kk * 32 + fg * 8) * 2);
    };
    auto load_b = [&](uint4 (&dst)[TM][2], int tile) {
        int n, h0, w0;
        tile_origin(tile < a.tiles ? tile : a.tiles - 1, n, h0, w0);
        const char* bb = a.inb + n * clipb + ((long long)h0 * a.W + w0) * 64 * 2;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wave * 32 + j * 16 + frow, t = r >> 3, p = r & 7;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                dst[j][kk] = gload16_uncounted(bb + (((long long)t * HW + (p >> 2) * a.W + (p & 3)) * 64 + kk * 32 + fg * 8) * 2);
        }
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (total > 0) {
        load_wc(wnext, 0);
        load_b(bnext, blockIdx.x);
        if (is_w) issue_weights(0, 0);
        else {
            issue_image(0);
            if (total > 1) issue_image(1);
        }
    }
    __syncthreads();                                                 // BN parameters and the zero rows visible
    const int wm_a = wave & 3, wn = wave >> 2;                       // a conv: 32 pooled rows x 64 output channels per wave
    int c_tile = blockIdx.x, c_kc = 0;
    for (int q = 0; q < total; ++q) {
        const bool last = c_kc + 1 == a.kslabs;
        // ---- T0: everyone has finished stage q - 1: weight slot 1 is free.  Its refill is issued BEFORE the wait for the image:
        // with two 24-KB slots a k-half can only be fetched once the previous stage has consumed its slot, and behind the
        // wait its L2 latency would be exposed in every stage (measured: 6 us per stage instead of 3)
        if (q > 0) __builtin_amdgcn_s_barrier();
        if (is_w) issue_weights(q, 1);
        // ---- T1: image(q) and the register loads of this stage have landed
        if (is_w) wait_vmcnt<12>();                                  // k-half 0 (issued at B3 of stage q - 1) and k-half 1 stay in flight
        else if (q + 1 < total) {                                    // image(q + 1) (8 pieces) and the trunk stores behind it stay in flight
            if (q == 0) wait_vmcnt<8>();
            else if (a.sub) wait_vmcnt<9>();
            else wait_vmcnt<12>();
        } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                // ... for everyone; image slot (q + 2) % 3 is free
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 t = __builtin_bit_cast(u32x4, wnext[i][kk]);
                asm volatile("" : "+v"(t));
                wcur[i][kk] = __builtin_bit_cast(uint4, t);
            }
        if (c_kc == 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 t = __builtin_bit_cast(u32x4, bnext[j][kk]);
                    asm volatile("" : "+v"(t));
                    bcur[j][kk] = __builtin_bit_cast(uint4, t);
                }
        }
        // the register loads of the next stage; image waves: the DMA behind them (T1 of the next stage leaves IT in flight)
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < total) {
            load_wc(wnext, last ? 0 : c_kc + 1);
            if (last) load_b(bnext, c_tile + gridDim.x);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!is_w && q + 2 < total) issue_image(q + 2);
        __builtin_amdgcn_sched_barrier(0);
        int n, h0, w0;
        tile_origin(c_tile, n, h0, w0);
        char* img = img0 + (q % 3) * kCpaImg;
        // ---- c conv, 64 trunk channels of this slab; + residual, ReLU, max over the frame pair; the pooled frame t' goes to
        // image rows 16 (t' + 1) + p - inside this wave's own 32 residual rows, which it has read by then
        {
            f32x4 cc[TN][TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) cc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<DT>::run(wcur[i][kk], bcur[j][kk], cc[i][j]);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bnp + c_kc * 64 + i * 16 + fg * 4);
                const f32x4 sf = *reinterpret_cast<const f32x4*>(bnp + a.C + c_kc * 64 + i * 16 + fg * 4);
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int R = 8 + wave * 32 + j * 16 + frow;     // image row of this position (R & 7 == frow & 7)
                    const char* cell = img + R * 128 + (((i * 2 + (fg >> 1)) ^ (frow & 7)) << 4) + (fg & 1) * 8;
                    f32x4 v = cc[i][j] * sc + sf + Vec4<DT>::load(cell);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // lanes frow and frow ^ 8 hold frames 2t' and 2t' + 1 of the same pixel: row_ror:8.  The maximum first, the
                        // ReLU behind it (they commute, NaN included) - on the half of the values this lane stores
                        const float x = v[e];
                        const float y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
                        v[e] = max_nan(x, y);
                    }
                    cc[i][j] = v;
                }
            }
            // both lanes of a pair hold the maximum: frow < 8 writes channel tiles 0 and 1, frow >= 8 tiles 2 and 3
            const bool hi = frow >= 8;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int Rp = wave * 32 + j * 16 + 16 + (frow & 7);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_f(hi ? cc[ii + 2][j][e] : cc[ii][j][e]);
                    const int i = ii + (hi ? 2 : 0);
                    Vec4<DT>::store(img + Rp * 128 + (((i * 2 + (fg >> 1)) ^ (frow & 7)) << 4) + (fg & 1) * 8, v);
                }
            }
        }
        // k-half 0 of the a weights has landed: behind it only k-half 1 and this stage's register loads are in the queue
        if (is_w) {
            if (q + 1 >= total) wait_vmcnt<6>();
            else if (last) wait_vmcnt<18>();
            else wait_vmcnt<14>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // B2: the pooled slab is complete (raw barrier: the DMA stays in flight)
        // ---- the pooled slab leaves for HBM where the next stage's shortcut reads it (image waves only)
        if (!is_w) {
            if (a.sub) {
                const int tp = tid >> 4, px = (tid >> 3) & 1, ck = tid & 7, R = 16 * (tp + 1) + 2 * px;
                const u32x4 o = *reinterpret_cast<const u32x4*>(img + R * 128 + ((ck ^ (R & 7)) << 4));
                char* dst = a.outx + (((((long long)n * TO + tp) * (a.H >> 1) + (h0 >> 1)) * (a.W >> 1) + (w0 >> 1) + px) * a.C) * 2 +
                            c_kc * 128 + ck * 16;
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int cellid = tid + 256 * it, rho = cellid >> 3, ck = cellid & 7, tp = rho >> 3, p = rho & 7, R = 16 * (tp + 1) + p;
                    const u32x4 o = *reinterpret_cast<const u32x4*>(img + R * 128 + ((ck ^ (R & 7)) << 4));
                    char* dst = a.outx + (((((long long)n * TO + tp) * a.H + h0 + (p >> 2)) * a.W + w0 + (p & 3)) * a.C) * 2 + c_kc * 128 + ck * 16;
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));
                }
            }
        }
        // ---- a conv: k-half 0, then k-half 1 of the slab, three temporal taps each; fragments read one tap ahead
        const uint4* xs = reinterpret_cast<const uint4*>(img) + (16 * (4 * wm_a) + 16 * (frow >> 3) + (frow & 7)) * 8;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            if (kh == 1) {
                // weights(q, 1) have landed (the register loads issued behind them stay in flight); slot 0 is free after B3
                if (is_w) {
                    if (q + 1 >= total) wait_vmcnt<0>();
                    else if (last) wait_vmcnt<12>();
                    else wait_vmcnt<8>();
                }
                __builtin_amdgcn_s_barrier();
                if (is_w && q + 1 < total) issue_weights(q + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4* ws = smem + kh * (kCpaW / 16) + (wn * 64 + frow) * 4 + (fg ^ ((frow & 8) >> 2));
            const int c = (kh * 4 + fg) ^ (frow & 7);
            uint4 af[2][TN], bf[2][TM];
            auto read_tap = [&](uint4 (&fa)[TN], uint4 (&fb)[TM], int dt) {
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[i] = ws[(dt * 128 + i * 16) * 4];
#pragma unroll
                for (int j = 0; j < TM; ++j) fb[j] = xs[(16 * (2 * j + dt)) * 8 + c];
            };
            read_tap(af[0], bf[0], 0);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                if (dt + 1 < 3) read_tap(af[(dt + 1) & 1], bf[(dt + 1) & 1], dt + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<DT>::run(af[dt & 1][i], bf[dt & 1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!last) { ++c_kc; continue; }

        // ---- tile finished: BN + ReLU + the one rounding; pairs of channel tiles trade halves between lane rows (swap_pair16): a
        // lane stores 16 contiguous bytes, a wave instruction 64-byte row segments
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int rho = wm_a * 32 + j * 16 + frow, tp = rho >> 3, p = rho & 7;
            const long long pos = (((long long)n * TO + tp) * a.H + h0 + (p >> 2)) * a.W + w0 + (p & 3);
#pragma unroll
            for (int ip = 0; ip < TN / 2; ++ip) {
                u32x2 half[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = 2 * ip + k;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(bnp + 2 * a.C + wn * 64 + i * 16 + fg * 4);
                    const f32x4 sf = *reinterpret_cast<const f32x4*>(bnp + 2 * a.C + 128 + wn * 64 + i * 16 + fg * 4);
                    half[k] = Vec4<DT>::pack_relu(acc[i][j] * sc + sf);
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const u32x4 o = swap_pair16(half[0], half[1]);
                const int ch = wn * 64 + (2 * ip + (fg & 1)) * 16 + (fg >> 1) * 8;
                *reinterpret_cast<u32x4*>(a.outa + (pos * 128 + ch) * 2) = o;
            }
        }
        c_kc = 0; c_tile += gridDim.x;
    }
}

template <int DT>
static int launch_cpa(const CPAArgs& a, int blocks, hipStream_t stream) {
    const int lds = (int)conv_cpa_lds_bytes(a.C);
    if (lds > 160 * 1024) return set_error(AF_ERR_ARG, "conv_cpa: %d bytes of LDS needed", lds);
    AF_SET_MAX_LDS((&conv_cpa_kernel<DT>), 160 * 1024, "conv_cpa");
    hipLaunchKernelGGL((conv_cpa_kernel<DT>), dim3(blocks), dim3(512), lds, stream, a);
    AF_CHECK_LAUNCH("conv_cpa_kernel");
    return AF_OK;
}

// dc: the 1x1x1 `c` conv (64 -> C, residual + ReLU) with the temporal pool behind it (tpool = 1);
// da: the 3x1x1 conv (C -> 128) over the pooled tensor; x_sub: 1 = store the whole pooled trunk, 2 = its even (h, w) positions
bool conv_cpa_applies(const af_conv_desc* dc, const af_conv_desc* da, int x_sub) {
    if (!dc || !da || (x_sub != 1 && x_sub != 2)) return false;
    if (dc->dtype == AF_F32 || da->dtype != dc->dtype || dc->tpool != 1 || da->tpool) return false;
    if (dc->kt != 1 || dc->kh != 1 || dc->kw != 1 || dc->st != 1 || dc->sh != 1 || dc->sw != 1 || dc->pt || dc->ph || dc->pw) return false;
    if (da->kt != 3 || da->kh != 1 || da->kw != 1 || da->st != 1 || da->sh != 1 || da->sw != 1 || da->pt != 1 || da->ph || da->pw) return false;
    if (dc->cin != 64 || da->cout != 128 || dc->cout != da->cin || dc->cout % 64 != 0 || !dc->relu || !da->relu) return false;
    if (dc->t != 32 || da->t != 16 || dc->n != da->n || dc->h != da->h || dc->w != da->w) return false;
    if (dc->to != dc->t || dc->ho != dc->h || dc->wo != dc->w || da->to != da->t || da->ho != da->h || da->wo != da->w) return false;
    if (dc->h % 2 != 0 || dc->w % 4 != 0) return false;               // tile = 2 rows x 4 columns
    if ((long long)dc->t * dc->h * dc->w * dc->cout * 2 >= (1LL << 31)) return false;     // 32-bit offsets inside a clip
    if (conv_cpa_lds_bytes(dc->cout) > 160 * 1024) return false;
    const long long tiles = (long long)dc->n * (dc->h / 2) * (dc->w / 4);
    return tiles >= 4LL * device_cus() && tiles < (1LL << 31);       // persistent stream: several tiles per workgroup
}

int conv_cpa_run(const af_conv_desc* dc, const void* inb, const void* wc, const float* scale_c, const float* shift_c,
                 const void* residual, void* outx, int x_sub, const af_conv_desc* da, const void* wa, const float* scale_a,
                 const float* shift_a, void* outa, hipStream_t stream) {
    CPAArgs a;
    a.inb = (const char*)inb; a.wc = (const char*)wc; a.scale_c = scale_c; a.shift_c = shift_c; a.res = (const char*)residual;
    a.outx = (char*)outx; a.wa = (const char*)wa; a.scale_a = scale_a; a.shift_a = shift_a; a.outa = (char*)outa;
    a.T = dc->t; a.H = dc->h; a.W = dc->w; a.C = dc->cout; a.kslabs = dc->cout / 64;
    a.wq = dc->w / 4; a.per_clip = (dc->h / 2) * a.wq; a.tiles = dc->n * a.per_clip;
    a.sub = x_sub == 2;
    const int cus = device_cus();
    const int blocks = a.tiles < cus ? a.tiles : cus;
    return dc->dtype == AF_BF16 ? launch_cpa<AF_BF16>(a, blocks, stream) : launch_cpa<AF_F16>(a, blocks, stream);
}

}  // namespace af

extern "C" int af_conv_cpa_fusable(const af_conv_desc* dc, const af_conv_desc* da, int x_sub) {
    return af::conv_cpa_applies(dc, da, x_sub) ? 1 : 0;
}

extern "C" int af_conv3d_cpa_bn_act(const af_conv_desc* dc, const void* in_b, const void* wc_packed, const float* scale_c,
                                    const float* shift_c, const void* residual, void* out_x, int x_sub, const af_conv_desc* da,
                                    const void* wa_packed, const float* scale_a, const float* shift_a, void* out_a, void* stream) {
    using namespace af;
    AF_REQUIRE(dc && da && in_b && wc_packed && scale_c && shift_c && residual && out_x && wa_packed && scale_a && shift_a && out_a,
               "conv_cpa: null argument");
    AF_REQUIRE(aligned16(in_b) && aligned16(wc_packed) && aligned16(scale_c) && aligned16(shift_c) && aligned16(residual) &&
                   aligned16(out_x) && aligned16(wa_packed) && aligned16(scale_a) && aligned16(shift_a) && aligned16(out_a),
               "conv_cpa: buffers must be 16-byte aligned");
    AF_REQUIRE(conv_cpa_applies(dc, da, x_sub),
               "conv_cpa: this (1x1x1 c + temporal pool, 3x1x1 a) pair does not take the fused path (ask af_conv_cpa_fusable first)");
    return conv_cpa_run(dc, in_b, wc_packed, scale_c, shift_c, residual, out_x, x_sub, da, wa_packed, scale_a, shift_a, out_a,
                        (hipStream_t)stream);
}

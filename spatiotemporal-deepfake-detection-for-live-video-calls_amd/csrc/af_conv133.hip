// 1x3x3 / stride 1 / pad (0,1,1) convolution, 64 -> 64 channels, 16-bit operands: the `b` conv of the three
// s2 bottlenecks (reference altfreezing/slowfast/models/resnet_helper.py:283-297), + BN + ReLU.
//
// The generic implicit GEMM streams every activation row through LDS once PER TAP (9x) and re-reads the
// weight tile from LDS for every 32 positions; at Cin = Cout = 64 that makes the layer ingest / LDS bound
// (~0.37 PFLOP/s).  This kernel is shaped around what is small here:
//   * the whole weight tensor (64 x 9 x 64) lives in REGISTERS: 4 waves, ONE per SIMD, each with all 64 output
//     channels: 72 MFMA A-fragments (9 taps x 2 k-halves x 4 channel tiles = 288 registers, VGPRs + AGPRs - a wave
//     alone on its SIMD has all 512) for the life of the workgroup; the A operand never touches LDS, a B fragment
//     feeds 4 MFMAs, and the fragments of the next (tap, k-half) step are read while this one multiplies (with two
//     waves per SIMD and 32 channels each, the loop was a chain of LDS latencies: read, wait, 2 MFMAs);
//   * a workgroup walks a contiguous run of image strips (R = 4 output rows); each strip's input patch
//     ((R+2) rows, zero halo columns included) is brought in ONCE by LDS-DMA (buffer loads; out-of-range lanes = zeros), double-buffered under the MFMAs
//     of the previous strip, and all 9 taps read it: with the rows stored at a padded pitch WP = W + 2 a tap
//     (dh,dw) is the constant row offset dh*WP + dw, so every B fragment is a plain swizzled ds_read_b128;
//   * positions are the padded strip (R x WP, two halo columns per row computed and discarded: 3.4 % waste).
// Persistent: one workgroup per CU, two barriers per strip (~4 600 MFMA cycles per wave between them).  Results leave
// through an LDS output tile ([position][64 channels], swizzled like the patches): the workgroup stores whole 128-byte
// rows, and does so under the MFMAs of the NEXT strip.
#include "af_common.h"
#include <stdlib.h>

namespace af {


struct C133Args {
    const char* in;
    const char* w;       // packed [64][9][64]
    const float* scale;
    const float* shift;
    char* out;
    int H, W, frames;
    int strips_per_frame, total_strips;
    int rows_alloc;      // LDS rows per patch buffer (multiple of 8)
#ifdef AF_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/stamps_lib.sh): [workgroup][wave][8] shader-clock cycles per phase
    int dbg;                      // timing-only ablations (AF_C64_DBG; outputs are then garbage): 1 no tile stores, 2 no epilogue, 4 no patch DMA
#endif
};

// Diagnostic build (-DAF_STAMPS; never the shipped library): cycles of a wave's life by phase, summed over its strips.
// (s_memtime waits with lgkmcnt(0): the stamps sit where no LDS read is meant to stay in flight.)
#ifdef AF_STAMPS
#define C64_ACC_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime()
#define C64_ACC(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[slot] += t_ - st_prev; st_prev = t_; } while (0)
#ifdef C64_CT_DBG
#define C64_DBG(bit) ((C64_CT_DBG) & (bit))       // compile-time ablations: no branch left in the loop
#else
#define C64_DBG(bit) (a.dbg & (bit))
#endif
#define C64_FLUSH do { if (a.stamps && lane < 8) a.stamps[((long long)blockIdx.x * 8 + wave) * 8 + lane] = \
    lane == 0 ? st_acc[0] : lane == 1 ? st_acc[1] : lane == 2 ? st_acc[2] : lane == 3 ? st_acc[3] : lane == 4 ? st_acc[4] : lane == 5 ? st_acc[5] : lane == 6 ? st_acc[6] : st_acc[7]; } while (0)
#else
#define C64_ACC_DECL do {} while (0)
#define C64_ACC(slot) do {} while (0)
#define C64_DBG(bit) false
#define C64_FLUSH do {} while (0)
#endif


template <int DT, int R>
__global__ __launch_bounds__(256, 1) void conv133_c64_kernel(const C133Args a) {
    typedef Elem<DT> E;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int MT = 4;                              // m-tiles per wave (up to 16 per strip)
    constexpr int OTROWS = 16 * 16;                    // rows of the output tile (16 m-tiles of 16 positions)

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mg = wave;                               // m-tile group: tiles mg, mg + 4, ... (all 64 channels per wave)
    constexpr int NT4 = 4;                             // channel tiles per wave
    const int frow = lane & 15, fg = lane >> 4;
    const int WP = a.W + 2;
    const int NP = a.rows_alloc >> 3;                  // DMA pieces per patch
    const int buf_bytes = a.rows_alloc * 128;
    // output tile of a strip: [position][64 channels] in the output type, 128-byte rows with the same XOR swizzle as the
    // patches; every wave drops its 32 channels x 64 positions in, then the whole workgroup streams full rows out
    char* otile = reinterpret_cast<char*>(smem) + 2 * buf_bytes;

    // ---- weights -> registers (A operand: lane = (channel row, k-group))
    // 288 registers of weights do not fit the 256 VGPRs next to the accumulators and fragments; left to itself hipcc parks
    // the overflow in AGPRs and copies it back with one v_accvgpr_read PER REGISTER PER USE (282 copies for the 288 MFMAs of a
    // strip: with one wave per SIMD every one of them is an issue slot the MFMA stream loses).  An MFMA can take its A
    // operand straight from an AGPR: the empty asm below pins the first AGPR_FRAGS fragments there for the life of the
    // workgroup, the rest stay in VGPRs, and the loop has no copies at all.
    constexpr int AGPR_FRAGS = 44;
    u32x4 wreg[9][2][NT4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < NT4; ++i) {
                const int ch = i * 16 + frow;
                wreg[tap][kk][i] = *reinterpret_cast<const u32x4*>(a.w + ((ch * 9 + tap) * 64 + kk * 32 + fg * 8) * 2);
                if ((tap * 2 + kk) * NT4 + i < AGPR_FRAGS) asm volatile("" : "+a"(wreg[tap][kk][i]));
                else asm volatile("" : "+v"(wreg[tap][kk][i]));
            }
    // BN scale / shift wait in LDS (the weights take 144 VGPRs; a spill would put scratch loads - and their
    // vmcnt(0), which also waits for the patch DMA in flight - into the strip loop)
    float* bn_lds = reinterpret_cast<float*>(otile + OTROWS * 128);
    if (tid < 64) { bn_lds[tid] = a.scale[tid]; bn_lds[64 + tid] = a.shift[tid]; }

    // ---- patch producer.  LDS row j of a patch <-> padded pixel q = j - 1 : (r, c') = (q / WP, q % WP), input pixel
    // (h0 + r - 1, c' - 1).  The (r, c') split and the source offset of every (piece, lane) are the same for every
    // strip, so they are worked out once into an LDS table (entry = r << 24 | byte offset from pixel (h0 - 1, 0);
    // all ones = halo column / beyond the patch); per strip a piece costs one table read, the row test and the DMA.
    // Out-of-image lanes get an offset outside the buffer descriptor's range and the hardware writes zeros.
    unsigned* dma_tab = reinterpret_cast<unsigned*>(bn_lds + 128);
    const int dma_row = lane >> 3, dma_chunk = (lane & 7) ^ dma_row;      // LDS row inside a piece; source chunk
    for (int g = wave; g < NP; g += 4) {
        const int q = g * 8 + dma_row - 1;
        const int r = q / WP, c = q - r * WP;
        const bool ok = q >= 0 && r < R + 2 && c >= 1 && c <= a.W;
        dma_tab[g * 64 + lane] = ok ? ((unsigned)r << 24) | (unsigned)((r * a.W + (c - 1)) * 128 + dma_chunk * 16) : 0xffffffffu;
    }
    __syncthreads();                                   // table + BN parameters visible to every wave
    auto issue_patch = [&](int strip, int buf) {
        const int frame = strip / a.strips_per_frame;
        const int h0 = (strip - frame * a.strips_per_frame) * R;
        // origin = pixel (h0 - 1, 0) of the frame (one row above the image for the first strip: those lanes are masked)
        const i32x4 desc = make_desc(a.in + ((long long)frame * a.H + h0 - 1) * a.W * 128);
        for (int g = wave; g < NP; g += 4) {
            const unsigned e = dma_tab[g * 64 + lane];
            const bool ok = e != 0xffffffffu && (unsigned)(h0 - 1 + (int)(e >> 24)) < (unsigned)a.H;
            blds16(ok ? (e & 0xffffffu) : kOutOfRange, desc, 0, __builtin_amdgcn_readfirstlane(lds0 + buf * buf_bytes + g * 1024));
        }
    };

    // output side: thread -> (row = position in the padded strip, 16-byte chunk) for each of its 4 row stores
    int out_off[OTROWS * 8 / 256], out_row[OTROWS * 8 / 256];
#pragma unroll
    for (int j = 0; j < OTROWS * 8 / 256; ++j) {
        const int p = (tid + 256 * j) >> 3, r = p / WP, c = p - r * WP;
        const bool ok = r < R && c >= 1 && c <= a.W;                // halo columns / rows beyond the strip are not stored
        out_row[j] = r;
        out_off[j] = ok ? (r * a.W + (c - 1)) * 128 + (tid & 7) * 16 : -1;
    }

    // contiguous run of strips for this workgroup
    const int G = gridDim.x, b = blockIdx.x;
    const int s0 = (int)((long long)a.total_strips * b / G), s1 = (int)((long long)a.total_strips * (b + 1) / G);
    if (s0 < s1) issue_patch(s0, 0);

    // the workgroup streams the output tile of strip `sp` out as whole 128-byte rows, 16 bytes per lane
    auto store_tile = [&](int sp) {
        const int frame = sp / a.strips_per_frame;
        const int h0 = (sp - frame * a.strips_per_frame) * R;
        char* obase = a.out + ((long long)frame * a.H + h0) * a.W * 128;
#pragma unroll
        for (int j = 0; j < OTROWS * 8 / 256; ++j) {
            if (out_off[j] >= 0 && h0 + out_row[j] < a.H) {
                const int row = (tid + 256 * j) >> 3, chunk = tid & 7;
                const u32x4 o = *reinterpret_cast<const u32x4*>(otile + row * 128 + ((chunk ^ (row & 7)) << 4));
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(obase + out_off[j]));
            }
        }
    };

    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of strip s have landed
        __builtin_amdgcn_s_barrier();                             // ... everybody's; and buffer buf^1 is no longer read
        if (s + 1 < s1) issue_patch(s + 1, buf ^ 1);

        const char* xb = reinterpret_cast<const char*>(smem) + buf * buf_bytes;
        f32x4 acc[NT4][MT];
#pragma unroll
        for (int i = 0; i < NT4; ++i)
#pragma unroll
            for (int k = 0; k < MT; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 18 (tap, k-half) steps of 16 MFMAs (4 channel tiles x 4 m-tiles); the 4 B fragments of step i+1 are read while
        // step i multiplies.  One wave per SIMD: nothing else hides the LDS latency, but the whole register file is this
        // wave's (288 weight registers + 64 accumulators + two fragment sets).  All MT m-tiles are multiplied
        // unconditionally (a tile index beyond NT reads rows of the other buffer / the output tile - still inside this
        // workgroup's LDS - and is never stored).
        uint4 bf[2][MT];
        auto read_step = [&](int step, uint4 (&b)[MT]) {
            const int tap = step >> 1, kk = step & 1;
            const int row = frow + (tap / 3) * WP + (tap % 3);    // + 16 * m-tile: does not change row & 7
            const char* base = xb + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4);
#pragma unroll
            for (int k = 0; k < MT; ++k) b[k] = *reinterpret_cast<const uint4*>(base + (mg + 4 * k) * (16 * 128));
        };
        read_step(0, bf[0]);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            if (step + 1 < 18) read_step(step + 1, bf[(step + 1) & 1]);
            if (step == 6 && s > s0) store_tile(s - 1);           // the previous strip's tile leaves under the MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < MT; ++k)
#pragma unroll
                for (int i = 0; i < NT4; ++i) Mma<DT>::run(__builtin_bit_cast(uint4, wreg[step >> 1][step & 1][i]), bf[step & 1][k], acc[i][k]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: BN + ReLU + rounding in registers -> the strip's output tile in LDS (8 bytes per lane); the tile is
        // stored to HBM during the NEXT strip's MFMA loop (store_tile above).  Which (row, chunk) a thread stores and
        // where it lands relative to the strip never changes: worked out once (out_off / out_row).
        f32x4 sc[NT4], sf[NT4];
#pragma unroll
        for (int i = 0; i < NT4; ++i) {
            sc[i] = *reinterpret_cast<const f32x4*>(bn_lds + i * 16 + fg * 4);
            sf[i] = *reinterpret_cast<const f32x4*>(bn_lds + 64 + i * 16 + fg * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // every wave has read the previous tile out
        __builtin_amdgcn_s_barrier();                                       // (raw barrier: the patch DMA stays in flight)
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            const int row = (mg + 4 * k) * 16 + frow;                       // position inside the padded strip
#pragma unroll
            for (int i = 0; i < NT4; ++i) {
                f32x4 v = acc[i][k] * sc[i] + sf[i];
                v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]);
                const int ch = i * 16 + fg * 4;                             // 4 channels = 8 bytes: half a 16-byte chunk
                Vec4<DT>::store(otile + row * 128 + (((ch >> 3) ^ (row & 7)) << 4) + (ch & 4) * 2, v);
            }
        }
    }
    if (s0 < s1) {                                                          // the last strip's tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        store_tile(s1 - 1);
    }
}

// Round 3: the same strip walk on EIGHT waves, two per SIMD: wave = (channel half ng = wave >> 2, m-tile group mg = wave & 3),
// 32 output channels x 4 m-tiles each: 144 weight registers (9 taps x 2 k-halves x 2 channel tiles), 32 accumulators, two
// fragment sets - ~210 VGPRs, so two waves fit a SIMD.  With one wave per SIMD every non-MFMA instruction of the strip (patch DMA
// issue: 26 us per launch, tile stores: 16, epilogue: 14, of 145) sat in the MFMA stream's way; with two, one wave's DMA issue /
// epilogue / stores run under the other's MFMAs.  The price: every B fragment is read by both channel halves (2x the LDS
// fragment reads: ~125 B/clk of the CU's 256) and feeds 2 MFMAs instead of 4 - which is why the fragments of step s + 1 are read
// a whole step (8 MFMAs) ahead of their use (round 1's 8-wave form read them right in front: a chain of LDS latencies, 58 %).
template <int DT, int R>
__global__ __launch_bounds__(512, 2) void conv133_c64x2_kernel(const C133Args a) {
    typedef Elem<DT> E;
    static_assert(E::EPC == 8, "16-bit operands only");
    constexpr int MT = 4, NT2 = 2;
    constexpr int OTROWS = 16 * 16;

    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mg = wave & 3, ng = wave >> 2;
    const int frow = lane & 15, fg = lane >> 4;
    const int WP = a.W + 2;
    const int NP = a.rows_alloc >> 3;
    const int buf_bytes = a.rows_alloc * 128;
    char* otile = reinterpret_cast<char*>(smem) + 2 * buf_bytes;
    C64_ACC_DECL;

    // ---- this wave's 32 output channels of the weights -> registers
    // (taps 0-6 in registers: 112; the last two taps' fragments wait in LDS in fragment order - 144 + accumulators + fragments +
    //  the store / DMA temporaries spilled 22 registers, and a scratch reload's vmcnt(0) also waits for the patch DMA in flight)
    constexpr int RT = 7;                                        // taps whose weights live in registers
    // Round 4: the weights come in through LDS.  As 28 global loads per wave in fragment order (a wave instruction = 16 rows x 64
    // bytes, every wave of every workgroup asking for the same 73 KB at the same moment) the prologue took 16-22 k cycles of a
    // workgroup's ~240 k (in-kernel stamps) - ~14 bytes per clock and CU.  Now the [64][9][64] tensor is ONE coalesced LDS-DMA image
    // per workgroup (72 pieces of 8 x 128 bytes, rows swizzled like the patches, in the patch buffers' space), every wave takes its
    // fragments from there, and the first patch is issued once everyone has.
    u32x4 wreg[RT][2][NT2];
    float* bn_lds = reinterpret_cast<float*>(otile + OTROWS * 128);
    uint4* w8 = reinterpret_cast<uint4*>(bn_lds + 128);          // taps RT .. 8: [tap][k-half][channel tile 0..3][64 lanes]
    {
        const i32x4 wdesc = make_desc(a.w);
        const int wrow = lane >> 3, wchunk = (lane & 7) ^ wrow;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int g = wave + 8 * j;                          // rows 8 g .. 8 g + 7 of the 576 (channel, tap) rows
            blds16((unsigned)((g * 8 + wrow) * 128 + wchunk * 16), wdesc, 0, __builtin_amdgcn_readfirstlane(lds0 + g * 1024));
        }
        if (tid < 64) { bn_lds[tid] = a.scale[tid]; bn_lds[64 + tid] = a.shift[tid]; }
        wait_vmcnt<0>();
        __syncthreads();
        const uint4* wimg = smem;
        auto frag = [&](int ch, int tap, int kk) {               // 8 input channels kk * 32 + fg * 8 .. of (output channel ch, tap)
            const int row = ch * 9 + tap;
            return wimg[row * 8 + ((kk * 4 + fg) ^ (row & 7))];
        };
#pragma unroll
        for (int tap = 0; tap < RT; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < NT2; ++i) wreg[tap][kk][i] = __builtin_bit_cast(u32x4, frag(ng * 32 + i * 16 + frow, tap, kk));
        for (int idx = tid; idx < (9 - RT) * 512; idx += 512) {
            const int f = idx >> 6, tap = RT + (f >> 3), kk = (f >> 2) & 1, ct = f & 3;
            w8[idx] = frag(ct * 16 + frow, tap, kk);
        }
#pragma unroll
        for (int tap = 0; tap < RT; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < NT2; ++i) asm volatile("" : "+v"(wreg[tap][kk][i]));    // (registers for good: no re-reads inside the loop)
    }

    unsigned* dma_tab = reinterpret_cast<unsigned*>(w8 + (9 - RT) * 512);
    const int dma_row = lane >> 3, dma_chunk = (lane & 7) ^ dma_row;
    for (int g = wave; g < NP; g += 8) {
        const int q = g * 8 + dma_row - 1;
        const int r = q / WP, c = q - r * WP;
        const bool ok = q >= 0 && r < R + 2 && c >= 1 && c <= a.W;
        dma_tab[g * 64 + lane] = ok ? ((unsigned)r << 24) | (unsigned)((r * a.W + (c - 1)) * 128 + dma_chunk * 16) : 0xffffffffu;
    }
    __syncthreads();
    auto issue_patch = [&](int strip, int buf) {
        const int frame = strip / a.strips_per_frame;
        const int h0 = (strip - frame * a.strips_per_frame) * R;
        const i32x4 desc = make_desc(a.in + ((long long)frame * a.H + h0 - 1) * a.W * 128);
        for (int g = wave; g < NP; g += 8) {
            const unsigned e = dma_tab[g * 64 + lane];
            const bool ok = e != 0xffffffffu && (unsigned)(h0 - 1 + (int)(e >> 24)) < (unsigned)a.H;
            blds16(ok ? (e & 0xffffffu) : kOutOfRange, desc, 0, __builtin_amdgcn_readfirstlane(lds0 + buf * buf_bytes + g * 1024));
        }
    };
    const int G = gridDim.x, b = blockIdx.x;
    const int s0 = (int)((long long)a.total_strips * b / G), s1 = (int)((long long)a.total_strips * (b + 1) / G);
    if (s0 < s1) issue_patch(s0, 0);
    C64_ACC(0);                                                      // 0: prologue (weights, tables, first patch issue)
    auto store_tile = [&](int sp) {
        const int frame = sp / a.strips_per_frame;
        const int h0 = (sp - frame * a.strips_per_frame) * R;
        char* obase = a.out + ((long long)frame * a.H + h0) * a.W * 128;
        if (C64_DBG(1)) return;
#pragma unroll
        for (int j = 0; j < OTROWS * 8 / 512; ++j) {
            const int row = (tid + 512 * j) >> 3, chunk = tid & 7, r = row / WP, c = row - r * WP;
            if (r < R && c >= 1 && c <= a.W && h0 + r < a.H) {            // halo columns / rows beyond the strip are not stored
                const u32x4 o = *reinterpret_cast<const u32x4*>(otile + row * 128 + ((chunk ^ (row & 7)) << 4));
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(obase + (r * a.W + (c - 1)) * 128 + chunk * 16));
            }
        }
    };

    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C64_ACC(1);                                                  // 1: wait for the patch DMA (and the tile stores)
        __builtin_amdgcn_s_barrier();
        C64_ACC(2);                                                  // 2: barrier
        if (s + 1 < s1 && !C64_DBG(4)) issue_patch(s + 1, buf ^ 1);
        C64_ACC(3);                                                  // 3: next patch issue

        const char* xb = reinterpret_cast<const char*>(smem) + buf * buf_bytes;
        f32x4 acc[NT2][MT];
#pragma unroll
        for (int i = 0; i < NT2; ++i)
#pragma unroll
            for (int k = 0; k < MT; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ONE fragment set, refilled in place: m-tile k's fragment of step s + 1 is read right behind the two MFMAs that consumed
        // its fragment of step s - still a whole step (8 MFMAs of this wave) ahead of its use, with half the registers of two
        // sets (two sets + 144 weight registers spilled)
        uint4 bf[MT];
        uint4 aw[2][NT2];                                            // LDS-resident weight fragments of the step / the next step (taps 7, 8)
        auto read_frag = [&](int step, int k) {
            const int tap = step >> 1, kk = step & 1;
            const int row = frow + (tap / 3) * WP + (tap % 3);
            return *reinterpret_cast<const uint4*>(xb + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4) + (mg + 4 * k) * (16 * 128));
        };
#pragma unroll
        for (int k = 0; k < MT; ++k) bf[k] = read_frag(0, k);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            if (step == 6 && s > s0) store_tile(s - 1);
            // (round 4: the LDS-resident weight fragments of the last two taps are read ONCE per step, in front of its four m-tiles;
            //  read per use - hipcc does not merge loads across the sched_barriers - every MFMA pair of those four steps waited
            //  for an LDS round trip of its own: `rr [lgkmcnt(1)] M` in the ISA)
            //  ... and a whole step ahead of its use (two fragment sets, alternating by step parity)
            if (step + 1 >= 2 * RT && step + 1 < 18) {
#pragma unroll
                for (int i = 0; i < NT2; ++i) aw[(step + 1) & 1][i] = w8[((((step + 1) >> 1) - RT) * 8 + ((step + 1) & 1) * 4 + ng * 2 + i) * 64 + lane];
            }
#pragma unroll
            for (int k = 0; k < MT; ++k) {
#pragma unroll
                for (int i = 0; i < NT2; ++i)
                    Mma<DT>::run(step < 2 * RT ? __builtin_bit_cast(uint4, wreg[step < 2 * RT ? step >> 1 : 0][step & 1][i]) : aw[step & 1][i],
                                 bf[k], acc[i][k]);
                if (step + 1 < 18) bf[k] = read_frag(step + 1, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 sc[NT2], sf[NT2];
#pragma unroll
        for (int i = 0; i < NT2; ++i) {
            sc[i] = *reinterpret_cast<const f32x4*>(bn_lds + ng * 32 + i * 16 + fg * 4);
            sf[i] = *reinterpret_cast<const f32x4*>(bn_lds + 64 + ng * 32 + i * 16 + fg * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        C64_ACC(4);                                                  // 4: the 144 MFMAs (+ the previous tile's stores at step 6)
        __builtin_amdgcn_s_barrier();
        C64_ACC(5);                                                  // 5: barrier (everyone has read the previous tile out)
        if (!C64_DBG(2)) {
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            const int row = (mg + 4 * k) * 16 + frow;
#pragma unroll
            for (int i = 0; i < NT2; ++i) {
                f32x4 v = acc[i][k] * sc[i] + sf[i];
                v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]);
                const int ch = ng * 32 + i * 16 + fg * 4;
                Vec4<DT>::store(otile + row * 128 + (((ch >> 3) ^ (row & 7)) << 4) + (ch & 4) * 2, v);
            }
        }
        }
        C64_ACC(6);                                                  // 6: epilogue
    }
    if (s0 < s1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        store_tile(s1 - 1);
    }
    C64_ACC(7);
    C64_FLUSH;
}

// Round 4, tried and NOT kept (DESIGN 3.1b has the stamp tables): a "skewed" form of this kernel - waves 4-7 taking the strip's one
// barrier in the middle of their strip (three patch slots), DMA pieces issued one per tap-step with computed offsets, no output tile
// (register epilogue: permlane16 swaps + four counted 16-byte buffer stores per wave and strip).  Parity green, 129-135 us against
// 120-128 for this form; with the skew switched off 122-127: a tie.  What the stamps showed instead is where this form's time is:
// the MFMA phase of the two waves of a SIMD is MFMA-bound while both multiply (5.0-5.4 k cycles per strip for 288 MFMAs), the older
// wave of the two wins the issue arbitration and the younger one starves until it is alone, and alone a wave with one tap-step of
// LDS look-ahead runs at ~25 cycles per MFMA - so putting one wave's epilogue under the other's MFMAs buys what the lone stretches lose.
template <int DT>
static int launch_c133(C133Args& a, hipStream_t stream) {
    constexpr int R = 4;
    const int g_num_cus = device_cus();
    const int WP = a.W + 2;
    a.strips_per_frame = (a.H + R - 1) / R;
    a.total_strips = a.frames * a.strips_per_frame;
    a.rows_alloc = (((R + 2) * WP + 2 + 16) + 7) & ~7;
    const int lds = 2 * a.rows_alloc * 128 + 16 * 16 * 128 + 128 * 4 + 2 * 512 * 16 + (a.rows_alloc / 8) * 64 * 4;   // (+ the last two taps' fragments: 8-wave form)
    const int grid = a.total_strips < g_num_cus ? a.total_strips : g_num_cus;
    // AF_C64_WAVES=4: the round-1 / 2 form (one wave per SIMD, all 64 channels per wave) for A/B runs
    const char* ew = getenv("AF_C64_WAVES");
    if (ew && atoi(ew) == 4) {
        AF_SET_MAX_LDS((&conv133_c64_kernel<DT, R>), 160 * 1024, "conv133");
        hipLaunchKernelGGL((conv133_c64_kernel<DT, R>), dim3(grid), dim3(256), lds, stream, a);
    } else {
        AF_SET_MAX_LDS((&conv133_c64x2_kernel<DT, R>), 160 * 1024, "conv133");
        hipLaunchKernelGGL((conv133_c64x2_kernel<DT, R>), dim3(grid), dim3(512), lds, stream, a);
    }
    AF_CHECK_LAUNCH("conv133_c64_kernel");
    return AF_OK;
}

// true iff this layer takes the register-resident-weights path (also used by af_conv_variant)
bool conv133_applies(const af_conv_desc* d, const void* residual, int out_ld) {
    return d->dtype != AF_F32 && d->cin == 64 && d->cout == 64 && d->kt == 1 && d->kh == 3 && d->kw == 3 &&
           d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 1 && d->pw == 1 && d->relu && !d->tpool &&
           residual == nullptr && (out_ld == 0 || out_ld == 64) && 4 * (d->w + 2) <= 256 &&
           // a strip is 4 x (W + 2) positions of the 256 every workgroup multiplies: narrow frames (SlowFast's Fast pathway in
           // s5: 7 x 7) would run at 14 % and took 2.4x the generic kernel's time
           4 * (d->w + 2) >= 112;
}

int conv133_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                void* out, hipStream_t stream) {
    C133Args a;
    a.in = (const char*)in; a.w = (const char*)w_packed; a.scale = scale; a.shift = shift; a.out = (char*)out;
    a.H = d->h; a.W = d->w; a.frames = d->n * d->t;
#ifdef AF_STAMPS
    const char* ep = getenv("AF_STAMP_PTR");
    a.stamps = ep ? (unsigned long long*)strtoull(ep, nullptr, 0) : nullptr;
    const char* ed = getenv("AF_C64_DBG");
    a.dbg = ed ? atoi(ed) : 0;
#endif
    return d->dtype == AF_BF16 ? launch_c133<AF_BF16>(a, stream) : launch_c133<AF_F16>(a, stream);
}

}  // namespace af

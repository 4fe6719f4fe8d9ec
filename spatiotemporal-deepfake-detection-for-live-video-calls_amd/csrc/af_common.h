// Internal helpers shared by the gfx950 kernels of libafhip.so.  CDNA4 only: wave64, MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/af_hip.h"

namespace af {

int set_error(int code, const char* fmt, ...);

// ---- LDS-DMA helpers shared by the convolution kernels
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// one 16-byte-per-lane LDS-DMA: LDS[lds_base + lane*16 .. +16) <- desc.base[voff + soff .. +16); a lane whose
// voff is outside the descriptor's 2 GiB window (kOutOfRange) gets zeros - that is how padding taps, rows beyond M
// and channel tails are filled.  hipcc does not count this load: every wait on it is an explicit s_waitcnt vmcnt(N).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOutOfRange = 0x80000000u;
__device__ __forceinline__ void blds16(unsigned voff, const i32x4& desc, int soff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(desc), "s"(lds_base), "s"(soff) : "memory");
}
// the same without saving m0 around the load (2 scalar instructions per piece instead of 4; hipcc is told that m0 is clobbered):
// for the streaming kernels, which are bound by instruction issue at 2 waves per SIMD
__device__ __forceinline__ void blds16_m0(unsigned voff, const i32x4& desc, int soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 : : "v"(voff), "s"(desc), "s"(lds_base), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void blds16_nt_m0(unsigned voff, const i32x4& desc, int soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen nt lds"
                 : : "v"(voff), "s"(desc), "s"(lds_base), "s"(soff) : "memory", "m0");
}
// the same with the non-temporal cache policy: for bytes one CU reads once (a residual stream)
__device__ __forceinline__ void blds16_nt(unsigned voff, const i32x4& desc, int soff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(desc), "s"(lds_base), "s"(soff) : "memory");
}
// A plain 16-byte buffer load that hipcc does NOT count: for register prefetch several tiles ahead.  (For a counted load the
// compiler must assume that nothing younger is in the queue when the value is finally used - the stores and loads issued since
// are conditional for it - and emits s_waitcnt vmcnt(0..3), which drains the LDS-DMA issued behind it as well.)  The caller
// waits with its own counted s_waitcnt vmcnt and passes the registers through an empty asm ("+v") before the first use.
__device__ __forceinline__ u32x4 bload16_nt_uncounted(unsigned voff, const i32x4& desc, int soff) {
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r) : "v"(voff), "s"(desc), "s"(soff) : "memory");
    return r;
}
// (the same for a 64-bit address)
__device__ __forceinline__ uint4 gload16_uncounted(const void* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return __builtin_bit_cast(uint4, r);
}
// raw buffer descriptor over [base, base + 2 GiB): stride 0, no swizzle, 32-bit data format
__device__ __forceinline__ i32x4 make_desc(const char* base) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    d[2] = (int)kOutOfRange;
    d[3] = 0x00020000;
    return d;
}

// ReLU and pooling max with torch's NaN behaviour: a NaN activation stays NaN through clamp_min / max_pool, so a clip
// with a non-finite pixel yields a NaN logit (as in the reference) instead of a plausible score
// (gfx950 has the IEEE-754-2019 maximum / minimum - v_maximum3_f32 / v_minimum3_f32 - which return NaN when an operand is NaN:
//  one instruction each, where a compare + select took two to four and the v_max3 / v_min3 forms needed separate NaN flags)
__device__ __forceinline__ float relu_f(float v) { return __builtin_elementwise_maximum(v, 0.f); }
__device__ __forceinline__ float max_nan(float m, float x) { return __builtin_elementwise_maximum(m, x); }
__device__ __forceinline__ float min_nan(float m, float x) { return __builtin_elementwise_minimum(m, x); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, MAXN] (the immediate is a compile-time constant: a scalar branch chain)
template <int MAXN, int K = 0> __device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    if constexpr (K >= MAXN) wait_vmcnt<MAXN>();
    else { if (n == K) wait_vmcnt<K>(); else wait_vmcnt_dyn<MAXN, K + 1>(n); }
}

// LDS fragment reads hipcc does NOT count (round 4).  A software-pipelined K loop reads the fragments of half-step h + 1 while
// half-step h multiplies; when the reads of h were issued in the previous loop iteration and the body has control flow in it
// (conditional DMA issue), hipcc's wait insertion gives up counting across the back edge and puts s_waitcnt lgkmcnt(0) in front
// of the first MFMA - behind the reads just issued, i.e. the whole LDS round trip of 8 waves x 11 reads is exposed once per
// K-step (conv133g, ISA dump of tools/isa_schedule.py).  With the reads as inline asm the kernel waits itself: lgkmcnt counts LDS
// operations in issue order, so wait_lgkmcnt<N>() = "all but my N youngest reads are back"; the registers then pass through
// pin_frag (an empty asm with a "+v" operand) so that no consumer can be scheduled above the wait.
template <int OFF> __device__ __forceinline__ u32x4 lds_read16_uncounted(unsigned addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF> __device__ __forceinline__ unsigned lds_read4_uncounted(unsigned addr) {
    unsigned r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin_frag(u32x4& v) { asm volatile("" : "+v"(v)); }
// An MFMA as an asm statement, accumulating IN PLACE ("+v"): hipcc's three-address form of the builtin gave every unrolled K-step
// body its own accumulator registers (an unrolled 9-tap loop of 28 accumulator tiles then spilled ~300 registers), and an asm
// statement is issued exactly where it is written.  Hazards are the caller's: an accumulate chain on the same tile needs no
// wait state; anything ELSE that reads or writes the tile behind the last MFMA does (mfma_drain() in front of the epilogue).
template <int DT> struct MmaAsm;
template <> struct MmaAsm<AF_BF16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
};
template <> struct MmaAsm<AF_F16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
};
// The accumulators of an asm-MFMA loop must be ordinary live registers BEFORE the loop: left to itself hipcc sinks "acc = 0" to the
// first use (a peeled first iteration: v_mov zero, zero, MFMA, v_mov ...), i.e. it writes an MFMA's C operand with vector moves
// one or two instructions in front of the asm MFMA that reads it - and reuses a fragment register the previous MFMA is still
// reading for the next tile's zeros - without the wait states its hazard recogniser would give its own MFMAs (whole-network f16
// logits moved by 2e-3, differently from run to run).  acc_live() makes the zeroed tile opaque (no rematerialisation, one register
// set for the whole loop); mfma_operands_settled() = the wait states between the last vector write of an operand and the first MFMA.
__device__ __forceinline__ void acc_live(f32x4& c) { asm volatile("" : "+v"(c)); }
__device__ __forceinline__ void mfma_operands_settled() { asm volatile("s_nop 3" ::: "memory"); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }   // > the 8-pass MFMA's 12 wait states
// What rides behind MFMA t of a K-step's second MFMA group: NP DMA pieces and NR fragment reads dealt out one per MFMA in the
// pattern piece, read, read, piece, ... (a piece first: the DMA of the stage two ahead wants all the lead it can get; the reads
// only have to be back by the next group) until one kind runs out.  group2_slot(t): >= 0: piece index; < 0: read index -1 - r;
// kNoSlot: nothing.
constexpr int kNoSlot = 1 << 20;
constexpr int group2_slot(int t, int NP, int NR) {
    int p = 0, r = 0;
    for (int i = 0;; ++i) {
        int what = kNoSlot;
        if (p < NP && (i % 3 == 0 || r >= NR)) what = p++;
        else if (r < NR) what = -1 - r++;
        if (i == t) return what;
        if (what == kNoSlot) return kNoSlot;
    }
}
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) (asm immediates need constants)
template <int I> struct IC { static constexpr int value = I; constexpr operator int() const { return I; } };
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<N, I + 1>(f); }
}

// af_conv133.hip: register-resident-weights 1x3x3 64->64 kernel (s2 `b` convs), 16-bit dtypes
bool conv133_applies(const af_conv_desc* d, const void* residual, int out_ld);
int conv133_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                void* out, hipStream_t stream);
// af_conv133g.hip: frame / band resident 1x3x3 kernel for 128 / 256 output channels (s3 / s4 `b` convs), 16-bit dtypes
bool conv133g_applies(const af_conv_desc* d, const void* residual, int out_ld);
int conv133g_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                 void* out, int out_ld, hipStream_t stream);
// ... and its TEMPORAL mode: 3x1x1 convs into 128 / 256 channels (s3 / s4 `a` convs): (T + 2) x P patch, three taps share it
bool conv311g_applies(const af_conv_desc* d, const void* residual, int out_ld);
int conv311g_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                 void* out, int out_ld, hipStream_t stream);
bool conv133g_fused_applies(const af_conv_desc* db, const af_conv_desc* dc, int out_ld);
int conv133g_fused_run(const af_conv_desc* db, const void* in, const void* wb, const float* scale_b, const float* shift_b,
                       const af_conv_desc* dc, const void* wc, const float* scale_c, const float* shift_c, const void* residual,
                       void* out, int out_ld, hipStream_t stream);
// af_conv_cpa.hip: the s2 -> s3 boundary - c + residual + ReLU, the temporal max-pool and the next stage's 3x1x1 a conv in one launch
bool conv_cpa_applies(const af_conv_desc* dc, const af_conv_desc* da, int x_sub);
int conv_cpa_run(const af_conv_desc* dc, const void* inb, const void* wc, const float* scale_c, const float* shift_c,
                 const void* residual, void* outx, int x_sub, const af_conv_desc* da, const void* wa, const float* scale_a,
                 const float* shift_a, void* outa, hipStream_t stream);
// af_conv_ca.hip: c(i) -> a(i+1) across a block boundary of s2 in the time-tiled layout (the trunk slab is produced in LDS)
bool conv_ca_applies(const af_conv_desc* dc, const af_conv_desc* d1, const af_conv_desc* da);
int conv_ca_run(const af_conv_desc* dc, const void* inb, const void* wc, const void* in1, const void* w1, const float* scale_c,
                const float* shift_c, const void* residual, void* outx, const af_conv_desc* da, const void* wa, const float* scale_a,
                const float* shift_a, void* outa, hipStream_t stream);
// af_conv_small.hip: direct-gather MFMA path for narrow layers (<= 16 output channels, <= 32 K chunks): SlowFast's Fast pathway
bool conv_small_applies(const af_conv_desc* d, const af_conv_desc* d2, const void* residual, int out_ld);
int conv_small_run(const af_conv_desc* d, const void* in, const void* w_packed, const af_conv_desc* d2, const void* in2,
                   const void* w2_packed, const float* scale, const float* shift, const void* residual, void* out, int out_ld,
                   hipStream_t stream);
// af_conv311.hip: time-tiled 3x1x1 -> 64 channels kernel (s2 `a` convs): the three taps share one LDS image
bool conv311_applies(const af_conv_desc* d, const void* residual, int out_ld);
int conv311_run(const af_conv_desc* d, const void* in, const void* w_packed, const float* scale, const float* shift,
                void* out, int out_ld, hipStream_t stream);

// af_conv111.hip: persistent weights-in-registers 1x1x1 stream (short-K `c` convs of s2 / s3, with the projection
// shortcut, the residual and the temporal pool), 16-bit dtypes
bool conv111_applies(const af_conv_desc* d, const af_conv_desc* d2, const void* residual, int out_ld);
int conv111_run(const af_conv_desc* d, const void* in, const void* w_packed, const af_conv_desc* d2, const void* in2,
                const void* w2_packed, const float* scale, const float* shift, const void* residual, void* out, int out_ld,
                hipStream_t stream);

// One-time PER-DEVICE setup.  hipFuncSetAttribute and the CU count belong to a device, and one process may drive several
// (one engine per device): every "done once" flag is therefore indexed by the current device ordinal.
constexpr int kMaxDevices = 64;
static inline int current_device() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices) ? d : -1;
}
struct DeviceOnce { bool done[kMaxDevices] = {}; };
// compute units of the current device (cached per device; af_api.hip)
int device_cus();

// raise a kernel's dynamic-LDS limit once per (kernel instantiation, device)
#define AF_SET_MAX_LDS(kernel_ptr, bytes, what)                                                                        \
    do {                                                                                                               \
        static af::DeviceOnce once__;                                                                                  \
        const int dev__ = af::current_device();                                                                        \
        if (dev__ < 0 || !once__.done[dev__]) {                                                                        \
            hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                            \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));                 \
            if (e__ != hipSuccess)                                                                                     \
                return af::set_error(AF_ERR_LAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e__));      \
            if (dev__ >= 0) once__.done[dev__] = true;                                                                 \
        }                                                                                                              \
    } while (0)

#define AF_REQUIRE(cond, ...)                                    \
    do {                                                         \
        if (!(cond)) return af::set_error(AF_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define AF_CHECK_LAUNCH(what)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess)                                                       \
            return af::set_error(AF_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int dtype_size(int dt) { return dt == AF_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == AF_F32 || dt == AF_BF16 || dt == AF_F16; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// One 16-byte operand chunk per lane -> MFMA(s) on a 16x16 fp32 accumulator tile.
//   16-bit: one v_mfma_f32_16x16x32 (8 k-values per lane, k = 8*(lane>>4)+j)
//   fp32  : four v_mfma_f32_16x16x4_f32; instruction j takes element j of the chunk, so it
//           reduces over k = {4*(lane>>4)+j}: both operands use the same k order, which is all
//           a dot product needs (exact fp32 fma chain per output element).
template <int DT> struct Mma;
template <> struct Mma<AF_BF16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<AF_F16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<AF_F32> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
    }
};

// element conversions (fp32 <-> storage type)
template <int DT> struct Elem;
template <> struct Elem<AF_F32> {
    typedef float type;
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<AF_BF16> {
    typedef __bf16 type;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
    static __device__ __forceinline__ __bf16 from_f32(float v) { return (__bf16)v; }
};
template <> struct Elem<AF_F16> {
    typedef _Float16 type;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ _Float16 from_f32(float v) { return (_Float16)v; }
};

// 4 consecutive channels <-> one vector store/load (16 B fp32, 8 B 16-bit)
template <int DT> struct Vec4;
template <> struct Vec4<AF_F32> {
    static __device__ __forceinline__ void store(void* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ f32x4 load(const void* p) { return *reinterpret_cast<const f32x4*>(p); }
};
// store_relu: ReLU (NaN kept, like torch's clamp_min) + the one rounding + store
template <> struct Vec4<AF_BF16> {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void store(void* p, f32x4 v) {
        b4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
        *reinterpret_cast<b4*>(p) = o;
    }
    static __device__ __forceinline__ void store_relu(void* p, f32x4 v) {
        v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]);
        store(p, v);
    }
    static __device__ __forceinline__ unsigned __attribute__((ext_vector_type(2))) pack_relu(f32x4 v) {
        b4 o; o[0] = (__bf16)relu_f(v[0]); o[1] = (__bf16)relu_f(v[1]); o[2] = (__bf16)relu_f(v[2]); o[3] = (__bf16)relu_f(v[3]);
        return __builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), o);
    }
    static __device__ __forceinline__ f32x4 load(const void* p) {
        b4 i = *reinterpret_cast<const b4*>(p);
        f32x4 o; o[0] = (float)i[0]; o[1] = (float)i[1]; o[2] = (float)i[2]; o[3] = (float)i[3];
        return o;
    }
    static __device__ __forceinline__ f32x4 unpack(unsigned __attribute__((ext_vector_type(2))) raw) {   // 8 bytes read earlier
        b4 i = __builtin_bit_cast(b4, raw);
        f32x4 o; o[0] = (float)i[0]; o[1] = (float)i[1]; o[2] = (float)i[2]; o[3] = (float)i[3];
        return o;
    }
};
// Two MFMA accumulator tiles (lane = 4 consecutive channels of position frow, k-group fg) -> 16 contiguous bytes per lane
// without LDS: v_permlane16_swap_b32 (gfx950) exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// so after swapping the packed halves of tile A and tile B a lane holds 8 consecutive channels: fg = 0 / 2: A's channels
// 0..7 / 8..15, fg = 1 / 3: B's.  (8-byte stores of 32-byte row segments cost the texture path ~3x their share)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 swap_pair16(u32x2 a, u32x2 b) {
    const u32x2 r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    return u32x4{r0[0], r1[0], r0[1], r1[1]};
}
template <> struct Vec4<AF_F16> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void store(void* p, f32x4 v) {
        h4 o; o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
        *reinterpret_cast<h4*>(p) = o;
    }
    static __device__ __forceinline__ void store_relu(void* p, f32x4 v) {
        v[0] = relu_f(v[0]); v[1] = relu_f(v[1]); v[2] = relu_f(v[2]); v[3] = relu_f(v[3]);
        store(p, v);
    }
    static __device__ __forceinline__ unsigned __attribute__((ext_vector_type(2))) pack_relu(f32x4 v) {
        h4 o; o[0] = (_Float16)relu_f(v[0]); o[1] = (_Float16)relu_f(v[1]); o[2] = (_Float16)relu_f(v[2]); o[3] = (_Float16)relu_f(v[3]);
        return __builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), o);
    }
    static __device__ __forceinline__ f32x4 load(const void* p) {
        h4 i = *reinterpret_cast<const h4*>(p);
        f32x4 o; o[0] = (float)i[0]; o[1] = (float)i[1]; o[2] = (float)i[2]; o[3] = (float)i[3];
        return o;
    }
    static __device__ __forceinline__ f32x4 unpack(unsigned __attribute__((ext_vector_type(2))) raw) {
        h4 i = __builtin_bit_cast(h4, raw);
        f32x4 o; o[0] = (float)i[0]; o[1] = (float)i[1]; o[2] = (float)i[2]; o[3] = (float)i[3];
        return o;
    }
};

}  // namespace af

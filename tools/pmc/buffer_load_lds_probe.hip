// Stand-alone probe (hipcc --offload-arch=gfx950 -O2, run on the GPU box): checks what the conv kernels rely on for
// `buffer_load_dwordx4 ... offen lds` - LDS image = M0 base + lane * 16, SGPR offset added, out-of-range lanes write zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* p, unsigned* out, int soff, unsigned nrec) {
    extern __shared__ uint4 smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    i32x4 desc;
    unsigned long long a = (unsigned long long)p;
    desc[0] = (int)(a & 0xffffffffu);
    desc[1] = (int)((a >> 32) & 0xffffu);
    desc[2] = (int)nrec;
    desc[3] = 0x00020000;
    // lane l reads 16 bytes at offset (63-l)*16 (reversed), lanes 5 and 40 out of range
    unsigned voff = (63 - threadIdx.x) * 16;
    if (threadIdx.x == 5 || threadIdx.x == 40) voff = 0x80000000u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(desc), "s"(lds0 + 256), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    char* d; unsigned* o;
    hipMalloc(&d, 16384); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, o, 4096, 16384u);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
        unsigned want = 0xdeadbeefu;
        int j = i - 64;   // dword index relative to lds0+256
        if (j >= 0 && j < 256) {
            int lane = j / 4, e = j % 4;
            if (lane == 5 || lane == 40) want = 0;
            else want = 1024 + (63 - lane) * 4 + e;   // soff 4096 bytes = dword 1024
        }
        if (r[i] != want) { if (bad < 10) printf("i=%d got %08x want %08x\n", i, r[i], want); ++bad; }
    }
    printf("buffer_load lds test: %d mismatches\n", bad);
    return bad != 0;
}

// Run ON THE GPU BOX:  hipcc --offload-arch=gfx950 -O2 tools/micro/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
// prints what v_permlane16_swap_b32 / v_permlane32_swap_b32 leave in the two operands (x = lane, y = 100 + lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* o) {
    const unsigned lane = threadIdx.x;
    u2 r = __builtin_amdgcn_permlane16_swap(lane, 100u + lane, false, false);
    o[lane] = r[0]; o[64 + lane] = r[1];
    u2 q = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    o[128 + lane] = q[0]; o[192 + lane] = q[1];
}
int main() {
    unsigned* d; unsigned h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[4] = {"permlane16_swap first ", "permlane16_swap second", "permlane32_swap first ", "permlane32_swap second"};
    for (int k = 0; k < 4; ++k) {
        printf("%s:", names[k]);
        for (int i = 0; i < 64; i += 4) printf(" %3u", h[k * 64 + i]);
        printf("   (every 4th lane)\n");
    }
    return 0;
}

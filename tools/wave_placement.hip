// tools/wave_placement.hip -- which SIMD do the waves of one workgroup land on?  (HW_REG_HW_ID: bits 5:4 = SIMD, 3:0 = wave slot)
// Build: hipcc --offload-arch=gfx950 -O2 tools/wave_placement.hip -o tools/wave_placement ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
    for (int threads : {256, 768, 1024}) {
        unsigned *d, h[4 * 16] = {};
        hipMalloc(&d, sizeof h);
        hipMemset(d, 0, sizeof h);
        hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; b++) {
            printf("threads %4d block %d: SIMD of waves 0..:", threads, b);
            for (int w = 0; w < threads / 64; w++) printf(" %u", (h[b * 16 + w] >> 4) & 3);
            printf("   (cu %u)\n", (h[b * 16] >> 8) & 15);
        }
        hipFree(d);
    }
    return 0;
}

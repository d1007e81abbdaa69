// wpattern.hip -- does the SHAPE of the write stream matter for a write-dominated 9 : 32 mix (decompress)?  (tuning aid)
//   pattern 0: every wave owns a 16 KiB output chunk and writes it as 16 x 1 KiB (decompress_kernel's shape)
//   pattern 1: the four waves of a block share a 64 KiB chunk: in step s wave w writes KiB (4 s + w) -- 4 KiB contiguous
//              per block and step, the block's write window advances together
//   pattern 2: as 0, but the wave's 16 stores are issued back to back after all its loads (burst)
// reads: 4.5 KiB per 16 KiB written (c = 9), nt 16-byte loads; stores nt.  Also a pure fill in the three shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wpattern.hip -o tools/wpattern ; run: tools/wpattern
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PATTERN, bool READ> __global__ __launch_bounds__(256) void wp_kernel(const u32x4 *src, u32x4 *dst, uint64_t nchunks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a chunk = 16 KiB of output (1024 x 16 B) + 4.5 KiB of input (288 x 16 B)
    if (PATTERN == 1) {
        for (uint64_t g = blockIdx.x; g * 4 < nchunks; g += gridDim.x) { // group of 4 chunks = 64 KiB out, 18 KiB in
            u32x4 acc = {0, 0, 0, 0};
            if (READ) {
                const u32x4 *p = src + g * 4 * 288 + wave * 288 + lane;
#pragma unroll
                for (int r = 0; r < 4; r++) acc ^= __builtin_nontemporal_load(p + r * 64);
                if (lane < 32) acc ^= __builtin_nontemporal_load(p + 4 * 64);
            }
            u32x4 *q = dst + g * 4 * 1024;
#pragma unroll
            for (int s = 0; s < 16; s++) {
                u32x4 v = acc; v.x += s;
                __builtin_nontemporal_store(v, q + (s * 4 + wave) * 64 + lane);
            }
        }
    } else {
        const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, stride = (uint64_t)gridDim.x * 4;
        for (uint64_t ch = w0; ch < nchunks; ch += stride) {
            u32x4 acc = {0, 0, 0, 0};
            if (READ) {
                const u32x4 *p = src + ch * 288 + lane;
#pragma unroll
                for (int r = 0; r < 4; r++) acc ^= __builtin_nontemporal_load(p + r * 64);
                if (lane < 32) acc ^= __builtin_nontemporal_load(p + 4 * 64);
            }
            u32x4 *q = dst + ch * 1024 + lane;
            if (PATTERN == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < 16; s++) {
                u32x4 v = acc; v.x += s;
                __builtin_nontemporal_store(v, q + s * 64);
            }
        }
    }
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint64_t nchunks = 4000000000ull / 16384 / 4 * 4; // ~4 GB out
    u32x4 *src, *dst;
    CK(hipMalloc(&src, nchunks * 288 * 16 + 4096)); CK(hipMalloc(&dst, nchunks * 16384));
    CK(hipMemset(src, 1, nchunks * 288 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char *name; void (*k)(const u32x4 *, u32x4 *, uint64_t); bool read; };
    const V vs[] = {{"fill  pattern 0 (wave owns 16 KiB)", wp_kernel<0, false>, false}, {"fill  pattern 1 (block 4 KiB/step)", wp_kernel<1, false>, false},
                    {"mix   pattern 0", wp_kernel<0, true>, true}, {"mix   pattern 1", wp_kernel<1, true>, true}, {"mix   pattern 2 (burst after loads)", wp_kernel<2, true>, true}};
    printf("%-40s %4s %10s %10s\n", "variant", "bpc", "median ms", "GB/s");
    for (const V &v : vs)
        for (int bpc : {1, 2, 4}) {
            std::vector<float> t;
            for (int rep = 0; rep < 5; rep++) {
                for (int i = 0; i < 2; i++) hipLaunchKernelGGL(v.k, dim3(bpc * cus), dim3(256), 0, 0, src, dst, nchunks);
                CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; i++) hipLaunchKernelGGL(v.k, dim3(bpc * cus), dim3(256), 0, 0, src, dst, nchunks);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 20);
            }
            std::sort(t.begin(), t.end());
            const double bytes = nchunks * (16384.0 + (v.read ? 4608.0 : 0.0));
            printf("%-40s %4d %10.4f %10.1f\n", v.name, bpc, t[2], bytes / t[2] / 1e6);
        }
    return 0;
}

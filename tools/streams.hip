// streams.hip -- how fast can P output streams be written when every wave writes one PIECE of each stream per step?
// (tuning aid for the per-predicate shared scan at large P.)  Wave w of the persistent grid handles steps
// w, w + W, ...; in step t it writes PIECE bytes at stream[p] + t * PIECE for every p < P -- the store pattern of
// shared_wide_kernel (PIECE = 512: one 8-byte store per lane and key) and of variants with larger pieces.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/streams.hip -o tools/streams ; run: tools/streams [GB=8]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// PIECE bytes per (wave, stream, step): 512 -> u32x2 per lane, 1024 -> u32x4, 2048 / 4096 -> 2 / 4 x u32x4
template <int PIECE, int NT> __global__ __launch_bounds__(256) void streams_kernel(uint8_t *out, uint64_t stream_bytes, uint32_t P, uint64_t nsteps)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t W = (uint64_t)gridDim.x * 4;
    for (uint64_t t = wave; t < nsteps; t += W) {
        for (uint32_t p = 0; p < P; p++) {
            uint8_t *dst = out + (uint64_t)p * stream_bytes + t * PIECE;
            if constexpr (PIECE == 512) {
                u32x2 v = {(uint32_t)t, p};
                if (NT) __builtin_nontemporal_store(v, (u32x2 *)dst + lane); else ((u32x2 *)dst)[lane] = v;
            } else {
#pragma unroll
                for (int j = 0; j < PIECE / 1024; j++) {
                    u32x4 v = {(uint32_t)t, p, (uint32_t)j, 0};
                    if (NT) __builtin_nontemporal_store(v, (u32x4 *)dst + j * 64 + lane); else ((u32x4 *)dst)[j * 64 + lane] = v;
                }
            }
        }
    }
}

template <int PIECE, int NT> float run(uint8_t *out, uint64_t total, uint32_t P, int bpc, int cus)
{
    const uint64_t stream_bytes = total / P / 4096 * 4096;
    const uint64_t nsteps = stream_bytes / PIECE;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((streams_kernel<PIECE, NT>), dim3(bpc * cus), dim3(256), 0, 0, out, stream_bytes, P, nsteps);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL((streams_kernel<PIECE, NT>), dim3(bpc * cus), dim3(256), 0, 0, out, stream_bytes, P, nsteps);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms / 3 < best ? ms / 3 : best;
    }
    printf("P=%4u piece=%4d nt=%d bpc=%d  %8.3f ms  %7.1f GB/s\n", P, PIECE, NT, bpc, best, stream_bytes * (double)P / best / 1e6);
    fflush(stdout);
    return best;
}

int main(int argc, char **argv)
{
    const uint64_t total = (uint64_t)(argc > 1 ? atoi(argv[1]) : 8) << 30;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint8_t *out;
    CK(hipMalloc(&out, total));
    for (uint32_t P : {8u, 64u, 512u})
        for (int bpc : {1, 2}) {
            run<512, 0>(out, total, P, bpc, cus);
            run<512, 1>(out, total, P, bpc, cus);
            run<1024, 1>(out, total, P, bpc, cus);
            run<2048, 1>(out, total, P, bpc, cus);
            run<4096, 1>(out, total, P, bpc, cus);
        }
    return 0;
}

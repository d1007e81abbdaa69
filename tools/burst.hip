// burst.hip -- does the LENGTH of the write bursts matter for a 9 : 1 read : write stream on MI355X?  (tuning aid)
//
// The equality scan at 1e9 x 9 bit reads 9 KiB and writes 1 KiB per wave tile and runs at 99 % of a trivial kernel with
// that mix (tools/ceilings.hip), but the 10 % of bitmap bytes cost ~15 % of the time next to a read-only stream.  This
// tool varies the one thing round 1 did not: how many KiB a wave writes back to back at consecutive addresses.
//   burst_kernel<R, K, NTS>: a wave owns chunks of K consecutive steps; per chunk it reads K x R KiB (nt 16-byte
//   loads), keeps one 16-byte result per step in registers and then stores the K results = K KiB contiguous per wave
//   (the four waves of a block own adjacent chunks: 4K KiB contiguous per block).  K = 1 is the scan's current shape.
//   NTS: 0 plain, 1 non-temporal, 2 write-through (sc1) stores.
//   DEFER = 1: the stores of chunk i are issued after the loads of chunk i+1 (the scan kernel's deferral).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/burst.hip -o tools/burst
// Run:   tools/burst [rows=1e9] [launches per burst=50]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

template <int NTS> __device__ __forceinline__ void store16(u32x4 *p, u32x4 v)
{
    if constexpr (NTS == 2)
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (NTS == 1)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <int R, int K, int NTS, int DEFER>
__global__ __launch_bounds__(256) void burst_kernel(const u32x4 *src, u32x4 *dst, uint64_t nchunks)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    u32x4 held[K];
    uint64_t held_chunk = ~0ull;
    for (uint64_t ch = wave; ch < nchunks; ch += stride) {
        u32x4 acc[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const u32x4 *p = src + (ch * K + k) * (uint64_t)(R * 64) + lane;
            u32x4 a = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < R; r++) a ^= __builtin_nontemporal_load(p + r * 64);
            acc[k] = a;
        }
        if constexpr (DEFER) {
            if (held_chunk != ~0ull) {
                u32x4 *q = dst + held_chunk * K * 64 + lane;
#pragma unroll
                for (int k = 0; k < K; k++) store16<NTS>(q + k * 64, held[k]);
            }
#pragma unroll
            for (int k = 0; k < K; k++) held[k] = acc[k];
            held_chunk = ch;
        } else {
            u32x4 *q = dst + ch * K * 64 + lane;
#pragma unroll
            for (int k = 0; k < K; k++) store16<NTS>(q + k * 64, acc[k]);
        }
    }
    if constexpr (DEFER) {
        if (held_chunk != ~0ull) {
            u32x4 *q = dst + held_chunk * K * 64 + lane;
#pragma unroll
            for (int k = 0; k < K; k++) store16<NTS>(q + k * 64, held[k]);
        }
    }
}

// read-only and write-only references on the same buffers
template <int R> __global__ __launch_bounds__(256) void read_kernel(const u32x4 *src, u32x4 *dst, uint64_t nsteps)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    u32x4 a = {0, 0, 0, 0};
    for (uint64_t s = wave; s < nsteps; s += stride) {
        const u32x4 *p = src + s * (uint64_t)(R * 64) + lane;
#pragma unroll
        for (int r = 0; r < R; r++) a ^= __builtin_nontemporal_load(p + r * 64);
    }
    if (a.x == 0x12345678u) dst[lane] = a;
}

struct Variant {
    std::string name;
    std::function<void(int bpc, hipStream_t)> launch;
};

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const int BURST = argc > 2 ? atoi(argv[2]) : 50;
    constexpr int R = 9;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint64_t nsteps = n * R / 8 / (R * 1024) / 64 * 64; // whole multiples of every K below
    const double bytes = nsteps * 1024.0 * (R + 1), rbytes = nsteps * 1024.0 * R;
    printf("device %s, %d CUs, %llu steps of %d KiB read + 1 KiB written (%.3f GB + %.3f GB)\n", prop.gcnArchName, cus,
           (unsigned long long)nsteps, R, rbytes / 1e9, nsteps * 1024.0 / 1e9);
    u32x4 *src, *dst;
    CK(hipMalloc(&src, nsteps * R * 1024 + 4096));
    CK(hipMalloc(&dst, nsteps * 1024 + 4096));
    CK(hipMemset(src, 0x5a, nsteps * R * 1024));
    CK(hipMemset(dst, 0, nsteps * 1024));

    std::vector<Variant> vs;
    vs.push_back({"read only", [=](int bpc, hipStream_t s) {
                      hipLaunchKernelGGL((read_kernel<R>), dim3(bpc * cus), dim3(256), 0, s, src, dst, nsteps);
                  }});
#define ADD(K, NTS, DEFER)                                                                                               \
    vs.push_back({std::string("K=") + #K + (NTS == 2 ? " sc1" : NTS == 1 ? " nt " : " pl ") + (DEFER ? " deferred" : ""), \
                  [=](int bpc, hipStream_t s) {                                                                          \
                      hipLaunchKernelGGL((burst_kernel<R, K, NTS, DEFER>), dim3(bpc * cus), dim3(256), 0, s, src, dst,    \
                                         nsteps / K);                                                                    \
                  }})
    ADD(1, 0, 0); ADD(1, 1, 0); ADD(1, 2, 0); ADD(1, 2, 1);
    ADD(2, 0, 0); ADD(2, 1, 0); ADD(2, 2, 0); ADD(2, 2, 1);
    ADD(4, 0, 0); ADD(4, 1, 0); ADD(4, 2, 0); ADD(4, 2, 1);
    ADD(8, 0, 0); ADD(8, 1, 0); ADD(8, 2, 0); ADD(8, 2, 1);
    ADD(16, 0, 0); ADD(16, 1, 0); ADD(16, 2, 0); ADD(16, 2, 1);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-22s %4s %10s %10s %10s   (bursts of %d launches back to back; median and best of 5 rounds, variants interleaved)\n",
           "variant", "bpc", "median ms", "best ms", "read GB/s", BURST);
    const int bpcs[] = {1, 2, 4};
    std::vector<std::vector<float>> times(vs.size() * 3);
    for (int round = 0; round < 5; round++)
        for (size_t i = 0; i < vs.size(); i++)
            for (int b = 0; b < 3; b++) {
                for (int w = 0; w < 3; w++) vs[i].launch(bpcs[b], 0);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int w = 0; w < BURST; w++) vs[i].launch(bpcs[b], 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                times[i * 3 + b].push_back(ms / BURST);
            }
    for (size_t i = 0; i < vs.size(); i++)
        for (int b = 0; b < 3; b++) {
            auto &t = times[i * 3 + b];
            std::sort(t.begin(), t.end());
            printf("%-22s %4d %10.4f %10.4f %10.1f\n", vs[i].name.c_str(), bpcs[b], t[t.size() / 2], t[0], rbytes / t[t.size() / 2] / 1e6);
        }
    (void)bytes;
    return 0;
}

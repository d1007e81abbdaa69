// ceilings.hip -- what can a TRIVIAL kernel reach with the same byte mix as each product kernel?
// (tuning aid, not part of the product.)  For every operation at 1e9 x C bit it times, in the same process and
// interleaved, the product kernel and `mix_kernel<R, W>`: a persistent grid whose waves read R KiB and write W KiB per
// step with 16-byte accesses and no arithmetic -- R : W = the read : write bytes of the operation
// (eq / range scan C : 1, shared scan P=8 C : 8, decompress C : 32).  The ratio product / mix is the part of the
// gap to the HBM roofline that the kernel itself (and not the memory system's read/write mix behaviour) owns.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTUNE_C=9] tools/ceilings.hip -o tools/ceilings
// Run:   tools/ceilings [rows=1e9] [burst=50]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../shared_simd_scan_amd/csrc/kernels.hpp"
#include "scan_kernel_r1.hpp"

using namespace mi355;

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

// one wave-step: R x 1 KiB read (nt), W x 1 KiB written (NTS: 1 nt stores, 2 write-through sc1 stores); steps strided over the persistent grid
template <int R, int W, int NTS> __global__ __launch_bounds__(256) void mix_kernel(const u32x4 *src, u32x4 *dst, uint64_t nsteps)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    for (uint64_t s = wave; s < nsteps; s += stride) {
        u32x4 acc = {0, 0, 0, 0};
        const u32x4 *p = src + s * (R * 64) + lane;
#pragma unroll
        for (int r = 0; r < R; r++) acc ^= __builtin_nontemporal_load(p + r * 64);
        u32x4 *q = dst + s * (W * 64) + lane;
#pragma unroll
        for (int w = 0; w < W; w++) {
            u32x4 v = acc;
            v.x += w;
            if (NTS == 2)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q + w * 64), "v"(v) : "memory");
            else if (NTS == 1)
                __builtin_nontemporal_store(v, q + w * 64);
            else
                q[w * 64] = v;
        }
    }
}

struct Variant {
    std::string name;
    std::function<void(int bpc, hipStream_t)> launch;
    std::vector<int> bpcs;
    double bytes;
};

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const int BURST = argc > 2 ? atoi(argv[2]) : 50;
#ifndef TUNE_C
#define TUNE_C 9
#endif
    constexpr int C = TUNE_C;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, n=%llu, c=%d\n", prop.gcnArchName, cus, (unsigned long long)n, C);

    const size_t pbytes = (n * C + 7) / 8 + 256;
    uint8_t *packed, *out;
    unsigned long long *hits, *scratch;
    CK(hipMalloc(&packed, pbytes + 65536));
    CK(hipMalloc(&out, 4 * n + 65536)); // decompress target; the bitmaps use its head
    CK(hipMalloc(&hits, 64));
    CK(hipMalloc(&scratch, kScratchWords * 8));
    CK(hipMemset(scratch, 0, kScratchWords * 8));
    PackArgs pa{};
    pa.n = n;
    pa.param = 42;
    pa.out = (uint32_t *)packed;
    pa.out_dwords = pbytes / 4;
    pa.c = C;
    hipLaunchKernelGGL(pack_kernel<kSrcSplitmix>, dim3(cus * 8), dim3(256), 0, 0, pa);
    CK(hipDeviceSynchronize());

    ScanArgs sa{};
    sa.packed = packed;
    sa.n = n;
    sa.out = out;
    sa.hits = hits;
    sa.scratch = scratch;
    sa.key[0] = 3;
    sa.nkeys = 1;
    ScanArgs sh = sa;
    const uint64_t stride8 = ((n / 8 + 4096 + 15) / 16) * 16;
    sh.out_stride = stride8;
    sh.nkeys = 8;
    sh.layout = 0;
    for (int q = 0; q < 8; q++) sh.key[q] = q * 37 + 3;
    DecompArgs da{packed, n, (int32_t *)out};

    const double read_bytes = n * C / 8.0;
    std::vector<Variant> vs;
    auto grid_of = [=](uint64_t ntiles, int bpc) {
        return (unsigned)std::min<uint64_t>((ntiles + kWavesPerBlock - 1) / kWavesPerBlock, (uint64_t)bpc * cus);
    };
    {
        constexpr int VPL = scan_vpl(C, kModeEq);
        using G = ScanGeom<C, VPL>;
        const uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        vs.push_back({"scan_eq (round-1 kernel, plain st)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((scan_kernel<C, kModeEq, 2, VPL>), dim3(grid_of(ntiles, bpc)), dim3(kBlockThreads), 0, s, sa);
                      }, {1, 2}, read_bytes + n / 8.0});
        vs.push_back({"scan_eq (round-1 kernel, sc1 st)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((scan_kernel<C, kModeEq, 34, VPL>), dim3(grid_of(ntiles, bpc)), dim3(kBlockThreads), 0, s, sa);
                      }, {1, 2}, read_bytes + n / 8.0});
        // the shipped kernel: 4 tiles per store burst (K = 1 at the widths where burst_k() says so: see width_group.hip)
        constexpr int KB = (C == 5 || C == 6 || (C >= 9 && C <= 16)) ? 4 : 1;
        vs.push_back({"scan_eq (product: burst, sc1 st)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((scan_burst_kernel<C, kModeEq, 34, VPL, KB>), dim3(grid_of((ntiles + KB - 1) / KB, bpc)), dim3(kBlockThreads), 0, s, sa);
                      }, {1, 2}, read_bytes + n / 8.0});
        ScanArgs sc = sa;
        sc.out = nullptr;
        vs.push_back({"scan_eq count only (product)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((scan_burst_kernel<C, kModeEq, 34, VPL, KB>), dim3(grid_of((ntiles + KB - 1) / KB, bpc)), dim3(kBlockThreads), 0, s, sc);
                      }, {1, 2}, read_bytes});
    }
    {
        constexpr int VPL = scan_vpl(C, kModeRange);
        using G = ScanGeom<C, VPL>;
        const uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        ScanArgs sr = sa;
        sr.key[0] = (1u << C) / 4;                          // lo
        sr.key[1] = (uint32_t)((1ull << C) / 2 - (1u << C) / 4); // hi - lo   (BASELINE config 3: [2^c/4, 2^c/2])
        vs.push_back({"scan_range (product, sc1 st)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((scan_kernel<C, kModeRange, 34, VPL>), dim3(grid_of(ntiles, bpc)), dim3(kBlockThreads), 0, s, sr);
                      }, {1, 2}, read_bytes + n / 8.0});
    }
    {
        const uint64_t nsteps = (uint64_t)(read_bytes / (C * 1024));
        vs.push_back({"mix C:1", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((mix_kernel<C, 1, 0>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, (u32x4 *)out, nsteps);
                      }, {1, 2, 4}, nsteps * 1024.0 * (C + 1)});
        vs.push_back({"mix C:1 sc1 stores", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((mix_kernel<C, 1, 2>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, (u32x4 *)out, nsteps);
                      }, {1, 2, 4}, nsteps * 1024.0 * (C + 1)});
    }
    {
        using G = ScanGeom<C, 64>;
        const uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        vs.push_back({"shared_scan P=8 (product, nt st)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((shared_lut_kernel<C, 18, 64, 0, false>), dim3(grid_of(ntiles, bpc)), dim3(kBlockThreads), 0, s, sh);
                      }, {1, 2}, read_bytes + n});
        ScanArgs shl = sh;
        shl.layout = 1;
        vs.push_back({"shared_scan P=8 linear (nt st)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((shared_lut_kernel<C, 18, 64, 1, false>), dim3(grid_of(ntiles, bpc)), dim3(kBlockThreads), 0, s, shl);
                      }, {1, 2}, read_bytes + n});
    }
    {
        const uint64_t nsteps = (uint64_t)(read_bytes / (C * 1024));
        vs.push_back({"mix C:8", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((mix_kernel<C, 8, 0>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, (u32x4 *)out, nsteps);
                      }, {1, 2, 4}, nsteps * 1024.0 * (C + 8)});
        vs.push_back({"mix C:8 nt stores", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((mix_kernel<C, 8, 1>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, (u32x4 *)out, nsteps);
                      }, {1, 2, 4}, nsteps * 1024.0 * (C + 8)});
    }
    {
        const uint64_t ntiles = (n + DecompGeom<C>::TILE_VALUES - 1) / DecompGeom<C>::TILE_VALUES;
        vs.push_back({"decompress (product)", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((decompress_kernel<C, 18>), dim3(grid_of(ntiles, bpc)), dim3(kBlockThreads), 0, s, da);
                      }, {1, 2, 4}, read_bytes + 4.0 * n});
    }
    {
        const uint64_t nsteps = (uint64_t)(read_bytes / (C * 1024));
        vs.push_back({"mix C:32", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((mix_kernel<C, 32, 0>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, (u32x4 *)out, nsteps);
                      }, {1, 2, 4}, nsteps * 1024.0 * (C + 32)});
        vs.push_back({"mix C:32 nt stores", [=](int bpc, hipStream_t s) {
                          hipLaunchKernelGGL((mix_kernel<C, 32, 1>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, (u32x4 *)out, nsteps);
                      }, {1, 2, 4}, nsteps * 1024.0 * (C + 32)});
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-28s %4s %12s %10s   (bursts of %d launches back to back, best of 3)\n", "kernel", "bpc", "ms/launch", "GB/s", BURST);
    for (auto &v : vs)
        for (int bpc : v.bpcs) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                for (int i = 0; i < 3; i++) v.launch(bpc, 0);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < BURST; i++) v.launch(bpc, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / BURST);
            }
            printf("%-28s %4d %12.4f %10.1f\n", v.name.c_str(), bpc, best, v.bytes / best / 1e6);
            fflush(stdout);
        }
    return 0;
}

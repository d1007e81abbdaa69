// tune_scan.hip -- standalone tuning / ablation harness for the scan kernels (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_scan.hip -o tools/tune_scan
// Run on the GPU box: tools/tune_scan [rows] [bits=9 only] [column: 0 mod5 | 1 random]
// Every variant is timed with hipEvents per launch (median / min of REPS), interleaved in rounds as
// the CDNA guide asks (rule 24), and checked against the hit count of the first variant.
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

// plain streaming-read reference: 16 B / lane loads, XOR-reduced, nothing written (read ceiling)
template <int UNROLL, int NT> __global__ __launch_bounds__(256) void read_kernel(const u32x4 *p, uint64_t nvec, uint32_t *sink)
{
    u32x4 acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (NT)
                v[u] = __builtin_nontemporal_load(p + i + u * stride);
            else
                v[u] = p[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u];
    }
    for (; i < nvec; i += stride) acc ^= p[i];
    uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345678u) *sink = x;
}

// streaming-write and copy references (write ceiling / copy ceiling of the part)
template <int NT> __global__ __launch_bounds__(256) void fill_kernel(u32x4 *p, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        if (NT)
            __builtin_nontemporal_store(v, p + i);
        else
            p[i] = v;
    }
}
template <int NT> __global__ __launch_bounds__(256) void copy_kernel(const u32x4 *src, u32x4 *dst, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        if (NT)
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else
            dst[i] = src[i];
    }
}

struct Variant {
    std::string name;
    std::function<void(int bpc, hipStream_t)> launch;
    int max_bpc;
    double bytes; // algorithmic bytes per launch
    bool counts_hits;
};

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const int column = argc > 3 ? atoi(argv[3]) : 0;
    const int REPS = argc > 4 ? atoi(argv[4]) : 15;
    const int NOHITS = argc > 5 ? atoi(argv[5]) : 0;
#ifndef TUNE_C
#define TUNE_C 9
#endif
    constexpr int C = TUNE_C;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, n=%llu, c=%d, column=%s\n", prop.gcnArchName, cus, (unsigned long long)n, C,
           column ? "random" : "mod5");

    const size_t pbytes = (n * C + 7) / 8 + 256;
    uint8_t *packed, *bitmap;
    unsigned long long *hits;
    uint32_t *sink;
    CK(hipMalloc(&packed, pbytes));
    CK(hipMalloc(&bitmap, n / 8 + 4096));
    CK(hipMalloc(&hits, 64));
    unsigned long long *scratch;
    CK(hipMalloc(&scratch, kScratchWords * 8));
    CK(hipMemset(scratch, 0, kScratchWords * 8));
    CK(hipMalloc(&sink, 64));
    PackArgs pa{};
    pa.n = n;
    pa.first_row = 0;
    pa.param = column ? 42 : 5;
    pa.out = (uint32_t *)packed;
    pa.out_dwords = pbytes / 4;
    pa.c = C;
    if (column)
        hipLaunchKernelGGL(pack_kernel<kSrcSplitmix>, dim3(cus * 8), dim3(256), 0, 0, pa);
    else
        hipLaunchKernelGGL(pack_kernel<kSrcMod>, dim3(cus * 8), dim3(256), 0, 0, pa);
    CK(hipDeviceSynchronize());

    ScanArgs sa{};
    sa.packed = packed;
    sa.n = n;
    sa.out = bitmap;
    sa.out_stride = 0;
    sa.hits = NOHITS ? nullptr : hits;
    sa.scratch = scratch;
    sa.key[0] = 3;
    sa.nkeys = 1;

    std::vector<Variant> vs;
    const double scan_bytes = n * C / 8.0 + n / 8.0;
    const double read_bytes = n * C / 8.0;
#define SCAN_VARIANT(NAME, VPL, AUX, ABL, BYTES)                                                                        \
    vs.push_back({NAME,                                                                                                \
                  [=](int bpc, hipStream_t s) {                                                                        \
                      using G = ScanGeom<C, VPL>;                                                                      \
                      uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;                                     \
                      uint64_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;                                  \
                      unsigned grid = (unsigned)std::min<uint64_t>(want, (uint64_t)bpc * cus);                         \
                      hipLaunchKernelGGL((scan_kernel<C, kModeEq, AUX, VPL, ABL>), dim3(grid), dim3(kBlockThreads), 0, \
                                         s, sa);                                                                       \
                  },                                                                                                   \
                  ScanGeom<C, VPL>::OCC, BYTES, (ABL) == 0 || (ABL) == 3 || (ABL) == 5})
    SCAN_VARIANT("vpl128 aux0", 128, 0, 0, scan_bytes);
    SCAN_VARIANT("vpl128 nt", 128, 2, 0, scan_bytes);
    SCAN_VARIANT("vpl128 sc0", 128, 1, 0, scan_bytes);
    SCAN_VARIANT("vpl128 sc0+nt", 128, 3, 0, scan_bytes);
    SCAN_VARIANT("vpl128 nt xcd-contig", 128, 2, 5, scan_bytes);
#define SCAN_VARIANT_D2(NAME, VPL, AUX, BYTES)                                                                          \
    vs.push_back({NAME,                                                                                                \
                  [=](int bpc, hipStream_t s) {                                                                        \
                      using G = ScanGeom<C, VPL>;                                                                      \
                      uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;                                     \
                      uint64_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;                                  \
                      unsigned grid = (unsigned)std::min<uint64_t>(want, (uint64_t)bpc * cus);                         \
                      hipLaunchKernelGGL((scan_kernel<C, kModeEq, AUX, VPL, 0, 2>), dim3(grid), dim3(kBlockThreads), 0, \
                                         s, sa);                                                                       \
                  },                                                                                                   \
                  2, BYTES, true})
    SCAN_VARIANT_D2("vpl128 nt depth2", 128, 2, scan_bytes);
    SCAN_VARIANT_D2("vpl64 nt depth2", 64, 2, scan_bytes);
    SCAN_VARIANT("vpl128 nt sc1store", 128, 34, 0, scan_bytes);
    SCAN_VARIANT("vpl128 nt ntstore", 128, 18, 0, scan_bytes);
    SCAN_VARIANT("vpl128 aux0 ntstore", 128, 16, 0, scan_bytes);
    SCAN_VARIANT("vpl64 nt ntstore", 64, 18, 0, scan_bytes);
    SCAN_VARIANT("vpl64 aux0", 64, 0, 0, scan_bytes);
    SCAN_VARIANT("vpl64 nt", 64, 2, 0, scan_bytes);
    SCAN_VARIANT("vpl32 nt", 32, 2, 0, scan_bytes);
    SCAN_VARIANT("vpl128 nt ABL1 dma-only", 128, 2, 1, read_bytes);
    SCAN_VARIANT("vpl128 nt ABL2 dma+lds", 128, 2, 2, read_bytes);
    SCAN_VARIANT("vpl128 nt ABL3 no-store", 128, 2, 3, read_bytes);
    SCAN_VARIANT("vpl64 nt ABL1 dma-only", 64, 2, 1, read_bytes);
    SCAN_VARIANT("vpl64 nt ABL3 no-store", 64, 2, 3, read_bytes);
    // shared scan, 8 keys, per-predicate bitmaps (needs 8 bitmaps: reuse one big buffer)
    uint8_t *bitmap8;
    const uint64_t stride8 = ((n / 8 + 4096 + 15) / 16) * 16;
    CK(hipMalloc(&bitmap8, stride8 * 8));
    unsigned long long *dbg;
    CK(hipMalloc(&dbg, 8192 * 32));
    CK(hipMemset(dbg, 0, 8192 * 32));
    sa.keys_dev = (const int32_t *)dbg;
    ScanArgs sh = sa;
    sh.out = bitmap8;
    sh.out_stride = stride8;
    sh.nkeys = 8;
    sh.layout = 0;
    for (int q = 0; q < 8; q++) sh.key[q] = q;
    const double shared_bytes = n * C / 8.0 + n;
#define SHARED_VARIANT(NAME, VPL, ABL, BYTES)                                                                           \
    vs.push_back({NAME,                                                                                                \
                  [=](int bpc, hipStream_t s) {                                                                        \
                      using G = ScanGeom<C, VPL>;                                                                      \
                      uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;                                     \
                      uint64_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;                                  \
                      unsigned grid = (unsigned)std::min<uint64_t>(want, (uint64_t)bpc * cus);                         \
                      hipLaunchKernelGGL((scan_kernel<C, kModeShared, 2, VPL, ABL>), dim3(grid), dim3(kBlockThreads),  \
                                         0, s, sh);                                                                    \
                  },                                                                                                   \
                  (scan_occ<C, VPL, kModeShared>()), BYTES, false})
#define LUT_VARIANT(NAME, VPL, LAYOUT)                                                                                   \
    vs.push_back({NAME,                                                                                                \
                  [=](int bpc, hipStream_t s) {                                                                        \
                      using G = ScanGeom<C, VPL>;                                                                      \
                      uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;                                     \
                      uint64_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;                                  \
                      unsigned grid = (unsigned)std::min<uint64_t>(want, (uint64_t)bpc * cus);                         \
                      ScanArgs x = sh;                                                                                 \
                      x.layout = LAYOUT;                                                                               \
                      hipLaunchKernelGGL((shared_lut_kernel<C, 2, VPL, LAYOUT, false>), dim3(grid),                    \
                                         dim3(kBlockThreads), 0, s, x);                                                \
                  },                                                                                                   \
                  3, shared_bytes, false})
    LUT_VARIANT("lut8 vpl64 per-pred", 64, 0);
    LUT_VARIANT("lut8 vpl64 linear", 64, 1);
    LUT_VARIANT("lut8 vpl128 per-pred", 128, 0);
    SHARED_VARIANT("shared8 vpl64", 64, 0, shared_bytes);
    SHARED_VARIANT("shared8 vpl64 ABL1 dma-only", 64, 1, read_bytes);
    SHARED_VARIANT("shared8 vpl64 ABL2 dma+lds", 64, 2, read_bytes);
    SHARED_VARIANT("shared8 vpl64 ABL3 no-store", 64, 3, read_bytes);
    SHARED_VARIANT("shared8 vpl64 ABL4 stamps", 64, 4, shared_bytes);
    SCAN_VARIANT("vpl128 nt ABL4 stamps", 128, 2, 4, scan_bytes);
    SHARED_VARIANT("shared8 vpl128", 128, 0, shared_bytes);
    SHARED_VARIANT("shared8 vpl32", 32, 0, shared_bytes);
    SHARED_VARIANT("shared8 vpl32 ABL3 no-store", 32, 3, read_bytes);
    const uint64_t nvec = (n * C / 8) / 16;
    vs.push_back({"read x4 unroll4", [=](int bpc, hipStream_t s) {
                      hipLaunchKernelGGL((read_kernel<4, 0>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, nvec, sink);
                  }, 8, read_bytes, false});
    vs.push_back({"read x4 unroll4 nt", [=](int bpc, hipStream_t s) {
                      hipLaunchKernelGGL((read_kernel<4, 1>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, nvec, sink);
                  }, 8, read_bytes, false});
    vs.push_back({"read x4 unroll8 nt", [=](int bpc, hipStream_t s) {
                      hipLaunchKernelGGL((read_kernel<8, 1>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, nvec, sink);
                  }, 8, read_bytes, false});

    {
        const uint64_t wvec = (stride8 * 8) / 16; // the 8-bitmap buffer (~1 GB) as a write target
        u32x4 *wbuf = (u32x4 *)bitmap8;
        vs.push_back({"fill x4", [=](int bpc, hipStream_t s) { hipLaunchKernelGGL((fill_kernel<0>), dim3(bpc * cus), dim3(256), 0, s, wbuf, wvec); }, 8, wvec * 16.0, false});
        vs.push_back({"fill x4 nt", [=](int bpc, hipStream_t s) { hipLaunchKernelGGL((fill_kernel<1>), dim3(bpc * cus), dim3(256), 0, s, wbuf, wvec); }, 8, wvec * 16.0, false});
        const uint64_t cvec = std::min<uint64_t>(nvec, wvec);
        vs.push_back({"copy x4", [=](int bpc, hipStream_t s) { hipLaunchKernelGGL((copy_kernel<0>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, wbuf, cvec); }, 8, cvec * 32.0, false});
        vs.push_back({"copy x4 nt", [=](int bpc, hipStream_t s) { hipLaunchKernelGGL((copy_kernel<1>), dim3(bpc * cus), dim3(256), 0, s, (const u32x4 *)packed, wbuf, cvec); }, 8, cvec * 32.0, false});
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Cfg {
        int v, bpc;
        std::vector<float> ms;
        unsigned long long hits;
    };
    std::vector<Cfg> cfgs;
    for (int v = 0; v < (int)vs.size(); v++)
        for (int bpc : {1, 2, 3, 4, 6, 8})
            if (bpc <= vs[v].max_bpc) cfgs.push_back({v, bpc, {}, 0});
    // warm-up + correctness
    for (auto &c : cfgs) {
        CK(hipMemsetAsync(hits, 0, 8, 0));
        vs[c.v].launch(c.bpc, 0);
        CK(hipMemcpy(&c.hits, hits, 8, hipMemcpyDeviceToHost));
        CK(hipGetLastError());
        if (vs[c.v].name.find("ABL4") != std::string::npos) {
            unsigned long long st[2];
            CK(hipMemcpy(st, scratch + kScratchDone + 2, 16, hipMemcpyDeviceToHost));
            {
                std::vector<unsigned long long> h(8192 * 4);
                CK(hipMemcpy(h.data(), dbg, 8192 * 32, hipMemcpyDeviceToHost));
                int nb = c.bpc * cus;
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int b = 0; b < nb; b++) { t0 = std::min(t0, h[b * 4]); t1 = std::max(t1, h[b * 4 + 1]); }
                // histogram of blocks per (xcc, se, sh, cu) and of start times
                std::vector<int> percu(8 * 8 * 2 * 16, 0);
                int late = 0;
                double sumdur = 0;
                for (int b = 0; b < nb; b++) {
                    unsigned hw = (unsigned)h[b * 4 + 2], xcc = (unsigned)h[b * 4 + 3] & 15;
                    unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    percu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
                    if (h[b * 4] - t0 > (t1 - t0) / 10) late++;
                    sumdur += (double)(h[b * 4 + 1] - h[b * 4]);
                }
                int used = 0, mx = 0;
                for (int v : percu) { used += v > 0; mx = std::max(mx, v); }
                printf("[blocks] %-26s bpc=%d: %d blocks on %d distinct CUs (max %d per CU); %d started later than 10%% into the kernel; span %.1f us, mean block lifetime %.1f us\n",
                       vs[c.v].name.c_str(), c.bpc, nb, used, mx, late, (t1 - t0) * 0.01, sumdur / nb * 0.01);
            }
            printf("[clock] %-28s bpc=%d: %llu shader cycles / %llu ticks of 100 MHz = %.3f GHz\n", vs[c.v].name.c_str(), c.bpc, st[0], st[1], st[0] / (double)st[1] * 0.1);
        }
    }
    for (int r = 0; r < REPS; r++) {
        for (auto &c : cfgs) {
            CK(hipMemsetAsync(hits, 0, 8, 0));
            CK(hipEventRecord(e0, 0));
            vs[c.v].launch(c.bpc, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            c.ms.push_back(ms);
        }
    }
    const unsigned long long ref_hits = cfgs[0].hits;
    printf("%-28s %4s %9s %9s %10s %10s  %s\n", "variant", "bpc", "med ms", "min ms", "GB/s(med)", "GB/s(min)", "hits");
    for (auto &c : cfgs) {
        std::sort(c.ms.begin(), c.ms.end());
        float med = c.ms[c.ms.size() / 2], mn = c.ms[0];
        const Variant &v = vs[c.v];
        printf("%-28s %4d %9.4f %9.4f %10.1f %10.1f  %s\n", v.name.c_str(), c.bpc, med, mn, v.bytes / med / 1e6,
               v.bytes / mn / 1e6, !v.counts_hits ? "-" : (c.hits == ref_hits ? "ok" : "MISMATCH"));
    }
    printf("reference hits = %llu\n", ref_hits);
    // sustained mode: BURST launches back to back, no host sync in between (what bench.py times)
    const int BURST = 100;
    printf("\n%-28s %4s %12s %10s   (burst of %d launches back to back)\n", "variant", "bpc", "ms/launch", "GB/s", BURST);
    for (auto &c : cfgs) {
        const Variant &v = vs[c.v];
        if (v.name.find("ABL") != std::string::npos && v.name.find("ABL1") == std::string::npos) continue;
        for (int i = 0; i < 5; i++) v.launch(c.bpc, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < BURST; i++) v.launch(c.bpc, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= BURST;
        printf("%-28s %4d %12.4f %10.1f\n", v.name.c_str(), c.bpc, ms, v.bytes / ms / 1e6);
    }
    return 0;
}

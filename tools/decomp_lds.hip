// decomp_lds.hip -- does decompress_kernel's LDS access pattern matter?  (tuning aid; the product kernel is untouched)
// decompress_kernel reads every value's two dwords on their own (8 dword reads per lane and step of 256 values; half of
// its LDS cycles are bank conflicts, PMC SQ_LDS_BANK_CONFLICT).  The variant here reads the lane's SPAN once -- values
// 4l .. 4l+3 are 4C contiguous bits: ceil((31 + 4C) / 32) dwords, 3 at c = 9 instead of 8 -- and picks each value's two
// dwords with selects (the dword a value starts in is one of two candidates per lane, fixed for the whole launch).
// Same tile geometry, same stores, same grid; both kernels timed in the same process, launches back to back, interleaved.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTUNE_C=9 tools/decomp_lds.hip -o tools/decomp_lds_c9 ; run: tools/decomp_lds_c9 [rows]
#include "../shared_simd_scan_amd/csrc/kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef TUNE_C
#define TUNE_C 9
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace mi355;

template <int C> __global__ __launch_bounds__(kBlockThreads) void decompress_span_kernel(DecompArgs a)
{
    using G = DecompGeom<C>;
    constexpr int NDW = (31 + 4 * C + 31) / 32; // dwords that hold 4C bits starting at any bit of a dword
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES + 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const uint32_t *lds32 = (const uint32_t *)lds_wave;
    const uint64_t n = a.n;
    const uint64_t nfull = n / G::TILE_VALUES; // (full tiles only: the ragged tail is not what is measured here)
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    constexpr uint32_t mask = C == 32 ? 0xffffffffu : ((1u << C) - 1u);
    const uint32_t bit0 = 4u * lane * C, d0 = bit0 >> 5, p = bit0 & 31u;
    bool up[4];       // value j starts one dword later than floor(jC / 32) says
    uint32_t sft[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        up[j] = ((p + j * C) >> 5) != (uint32_t)((j * C) >> 5);
        sft[j] = (p + j * C) & 31u;
    }
    for (uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave; tile < nfull; tile += stride) {
        const uint8_t *src = a.packed + tile * G::TILE_BYTES;
#pragma unroll
        for (int j = 0; j < G::DMA_INSTRS; j++) {
            uint32_t o = j * 1024 + lane * 16;
            if (o < G::TILE_BYTES) __builtin_amdgcn_global_load_lds(MI355_GPTR(src + o), MI355_LPTR(lds_wave + j * 1024), 16, 0, 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int32_t *dst = a.out + tile * G::TILE_VALUES + lane * 4;
#pragma unroll
        for (int s = 0; s < G::STEPS; s++) {
            uint32_t w[NDW + 1];
#pragma unroll
            for (int i = 0; i < NDW; i++) w[i] = lds32[s * 8 * C + d0 + i];
            w[NDW] = 0;
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                constexpr int dummy = 0;
                (void)dummy;
                const int k = (j * C) >> 5;
                const uint32_t lo = up[j] ? w[k + 1 < NDW ? k + 1 : NDW] : w[k];
                const uint32_t hi = up[j] ? w[k + 2 < NDW ? k + 2 : NDW] : w[k + 1 < NDW ? k + 1 : NDW];
                v[j] = __builtin_amdgcn_alignbit(hi, lo, sft[j]) & mask;
            }
            __builtin_nontemporal_store(v, (u32x4 *)(dst + s * 256));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int main(int argc, char **argv)
{
    constexpr int C = TUNE_C;
    const uint64_t n = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull) / 4096 * 4096;
    const size_t packed_bytes = n * C / 8 + 256;
    uint8_t *packed;
    int32_t *out, *out2;
    CK(hipMalloc(&packed, packed_bytes));
    CK(hipMalloc(&out, n * 4 + 64));
    CK(hipMalloc(&out2, n * 4 + 64));
    std::vector<uint32_t> h(packed_bytes / 4 + 1);
    uint64_t x = 88172645463325252ull;
    for (auto &w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)x; }
    CK(hipMemcpy(packed, h.data(), packed_bytes, hipMemcpyHostToDevice));
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    DecompArgs a{packed, n, out}, b{packed, n, out2};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // both kernels produce the same values
    hipLaunchKernelGGL((decompress_kernel<C, 18>), dim3(cus * 2), dim3(kBlockThreads), 0, 0, a);
    hipLaunchKernelGGL((decompress_span_kernel<C>), dim3(cus * 2), dim3(kBlockThreads), 0, 0, b);
    CK(hipDeviceSynchronize());
    {
        std::vector<int32_t> x1(1 << 20), x2(1 << 20);
        for (uint64_t off : {(uint64_t)0, n / 2 / 4096 * 4096, n - (1 << 20)}) {
            CK(hipMemcpy(x1.data(), out + off, x1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(x2.data(), out2 + off, x2.size() * 4, hipMemcpyDeviceToHost));
            if (x1 != x2) { printf("MISMATCH at %llu\n", (unsigned long long)off); return 1; }
        }
    }
    for (int bpc : {1, 2, 4}) {
        std::vector<float> t0, t1;
        for (int round = 0; round < 5; round++) {
            for (int which = 0; which < 2; which++) {
                for (int i = 0; i < 3; i++) {
                    if (which == 0) hipLaunchKernelGGL((decompress_kernel<C, 18>), dim3(cus * bpc), dim3(kBlockThreads), 0, 0, a);
                    else hipLaunchKernelGGL((decompress_span_kernel<C>), dim3(cus * bpc), dim3(kBlockThreads), 0, 0, b);
                }
                CK(hipEventRecord(e0));
                for (int i = 0; i < 20; i++) {
                    if (which == 0) hipLaunchKernelGGL((decompress_kernel<C, 18>), dim3(cus * bpc), dim3(kBlockThreads), 0, 0, a);
                    else hipLaunchKernelGGL((decompress_span_kernel<C>), dim3(cus * bpc), dim3(kBlockThreads), 0, 0, b);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                (which == 0 ? t0 : t1).push_back(ms / 20);
            }
        }
        std::sort(t0.begin(), t0.end());
        std::sort(t1.begin(), t1.end());
        printf("c=%2d n=%.0e blocks/CU=%d  decompress_kernel (8 dword reads per lane-step) median %.4f ms   span variant (%d reads) median %.4f ms   ratio %.3f\n",
               C, (double)n, bpc, t0[t0.size() / 2], (31 + 4 * C + 31) / 32, t1[t1.size() / 2], t1[t1.size() / 2] / t0[t0.size() / 2]);
    }
    return 0;
}

// extras/histogram.hpp -- how often every value occurs in a packed column (SELECT v, count(*) ... GROUP BY v -- with the
// c-bit value a dictionary index, the usual aggregation over a dictionary-coded column), optionally only over the rows of a
// bitmap.  Widths up to 14 bits: the 2^c counters live in LDS, one LDS atomic per value, one global atomic per non-empty
// counter and block at the end.  Built from the scan's tile pipeline; included by capi.hip only.
#pragma once

#include "../kernels.hpp"

namespace mi355 {

constexpr int kHistogramMaxBits = 14; // 2^14 counters = 64 KiB of LDS next to the block's tiles

struct HistArgs {
    const uint8_t *packed; // 16 B aligned
    uint64_t n;
    const uint8_t *mask;     // rows that count (ceil(n/8) bytes, 4 B aligned) or null = every row
    unsigned long long *out; // 2^C counters, zeroed in front of the launch
};

template <int C, int VPL>
__global__ __launch_bounds__(kBlockThreads) void histogram_kernel(HistArgs a)
{
    static_assert(C <= kHistogramMaxBits, "the counters must fit in LDS");
    using G = ScanGeom<C, VPL>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = 2; // the column is streamed once: non-temporal DMA
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ uint32_t hist[1 << C]; // a block sees at most n / gridDim.x values: 32 bits are plenty
    for (uint32_t k = threadIdx.x; k < (1u << C); k += kBlockThreads) hist[k] = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint64_t nbytes = (a.n + 7) / 8;

    auto load_mask = [&](uint64_t t, uint32_t (&m)[WORDS]) {
#pragma unroll
        for (int j = 0; j < WORDS; j++) m[j] = 0xffffffffu;
        if (!a.mask) return;
        const uint64_t at = t * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4);
        if (t < tc.nfull) {
#pragma unroll
            for (int j = 0; j < WORDS; j++) m[j] = ((const uint32_t *)(a.mask + at))[j];
        } else { // ragged end: only the bytes the bitmap is guaranteed to hold
#pragma unroll
            for (int j = 0; j < WORDS; j++) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (at + 4 * j + b < nbytes) v |= (uint32_t)a.mask[at + 4 * j + b] << (8 * b);
                m[j] = v;
            }
        }
    };

    uint32_t mnext[WORDS];
    if (tile < tc.ntiles) {
        tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
        load_mask(tile, mnext);
    }
    __syncthreads(); // the counters are zero
    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        uint32_t m[WORDS];
#pragma unroll
        for (int j = 0; j < WORDS; j++) m[j] = mnext[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) {
            tc.template issue<AUX>(a.packed, next, lds_wave, lane);
            load_mask(next, mnext);
        }
        const bool full = tile < tc.nfull;
        if (!full) { // rows behind the column count for nothing
            const int64_t left = (int64_t)(a.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
            const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
#pragma unroll
            for (int j = 0; j < WORDS; j++) m[j] &= tail_mask(valid, j);
        }
        uint32_t xs[VPL];
        extract_all<C, VPL, 0, G::LANE_DWORDS>(w, xs);
        if (!a.mask && full) {
#pragma unroll
            for (int v = 0; v < VPL; v++) __hip_atomic_fetch_add(&hist[xs[v]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
#pragma unroll
            for (int v = 0; v < VPL; v++)
                if ((m[v >> 5] >> (v & 31)) & 1u) __hip_atomic_fetch_add(&hist[xs[v]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        tile = next;
    }
    __syncthreads(); // every wave's adds are done
    for (uint32_t k = threadIdx.x; k < (1u << C); k += kBlockThreads) {
        const uint32_t v = hist[k];
        if (v) __hip_atomic_fetch_add(a.out + k, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int C> inline void launch_histogram(const HistArgs &a, int num_cus, hipStream_t stream)
{
    constexpr int VPL = 128;
    using G = ScanGeom<C, VPL>;
    const uint64_t ntiles = (a.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
    // two blocks per CU while tiles + counters of two blocks fit in the CU's 160 KiB (the LDS atomics want the second wave per SIMD)
    const bool two = 2 * (4 * (size_t)G::LDS_BYTES + (4u << C)) + 1024 <= 160 * 1024;
    const uint64_t blocks_wanted = (uint64_t)num_cus * (two ? 2 : 1);
    const uint64_t blocks_needed = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned grid = (unsigned)(blocks_needed < blocks_wanted ? (blocks_needed ? blocks_needed : 1) : blocks_wanted);
    hipLaunchKernelGGL((histogram_kernel<C, VPL>), dim3(grid), dim3(kBlockThreads), 0, stream, a);
}

inline bool launch_histogram_width(unsigned c, const HistArgs &a, int num_cus, hipStream_t stream)
{
    switch (c) {
#define MI355_HIST_CASE(W) case W: launch_histogram<W>(a, num_cus, stream); return true;
        MI355_HIST_CASE(1) MI355_HIST_CASE(2) MI355_HIST_CASE(3) MI355_HIST_CASE(4) MI355_HIST_CASE(5) MI355_HIST_CASE(6) MI355_HIST_CASE(7)
        MI355_HIST_CASE(8) MI355_HIST_CASE(9) MI355_HIST_CASE(10) MI355_HIST_CASE(11) MI355_HIST_CASE(12) MI355_HIST_CASE(13) MI355_HIST_CASE(14)
#undef MI355_HIST_CASE
    default: return false;
    }
}

} // namespace mi355

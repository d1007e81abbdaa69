// extras/aggregate.hpp -- sum / min / max / count of a packed column, optionally only over the rows of a bitmap
// (SELECT sum(b), min(b), max(b), count(*) WHERE <bitmap>): the consumer that ends most predicate chains.  Built from the
// scan's pipeline (kernels/tile.hpp, kernels/scan.hpp: LDS-DMA tiles, lane owns a run, compile-time bit offsets) but not
// one of the profiled hot-path kernels; included by capi.hip only.
#pragma once

#include "../kernels.hpp"

namespace mi355 {

struct AggArgs {
    const uint8_t *packed; // 16 B aligned
    uint64_t n;
    const uint8_t *mask;   // bitmap of the rows that count (ceil(n/8) bytes, 4 B aligned) or null = every row
    unsigned long long *out; // [0] sum, [1] count, [2] min, [3] max -- initialised to 0, 0, ~0, 0 in front of the launch
};

static __global__ void aggregate_init_kernel(unsigned long long *out)
{
    out[0] = 0;
    out[1] = 0;
    out[2] = ~0ull;
    out[3] = 0;
}

// One pass over the column: per value an extraction, the row's bitmap bit spread over a word (v_bfe_i32), AND, add, max, and
// for the minimum a subtraction and a max (see the loop).  The lane's sum of a tile stays in 32 bits while 128 * 2^C fits
// (C <= 24), else 64.  ~7.5 VALU operations per value with a mask, ~4.5 without: at c = 9 about the time of the stream.
template <int C, int VPL>
__global__ __launch_bounds__(kBlockThreads) void aggregate_kernel(AggArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = 2; // the column is streamed once: non-temporal DMA
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint64_t nbytes = (a.n + 7) / 8;

    // the lane's mask words of a tile (ragged end: only the bytes the bitmap is guaranteed to hold)
    auto load_mask = [&](uint64_t t, uint32_t (&m)[WORDS]) {
#pragma unroll
        for (int j = 0; j < WORDS; j++) m[j] = 0xffffffffu;
        if (!a.mask) return;
        const uint64_t at = t * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4);
        if (t < tc.nfull) {
#pragma unroll
            for (int j = 0; j < WORDS; j++) m[j] = ((const uint32_t *)(a.mask + at))[j];
        } else {
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

    unsigned long long sum = 0, cnt = 0;
    uint32_t nmn = 0, mx = 0; // nmn: maximum of ~value over the rows that count (the minimum, complemented)
    uint32_t mnext[WORDS];
    if (tile < tc.ntiles) {
        tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
        load_mask(tile, mnext);
    }
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
        if (tile >= tc.nfull) { // rows behind the column count for nothing
            const int64_t left = (int64_t)(a.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
            const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
#pragma unroll
            for (int j = 0; j < WORDS; j++) m[j] &= tail_mask(valid, j);
        }
        uint32_t xs[VPL];
        extract_all<C, VPL, 0, G::LANE_DWORDS>(w, xs);
        uint32_t s32 = 0;
        unsigned long long s64 = 0;
        if (!a.mask && tile < tc.nfull) {
            // every row counts: add, max, min per value
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                const uint32_t x = xs[v];
                if constexpr (C <= 24)
                    s32 += x;
                else
                    s64 += x;
                mx = x > mx ? x : mx;
                nmn = ~x > nmn ? ~x : nmn;
            }
        } else {
            // sel = the row's bitmap bit spread over the word (one v_bfe_i32); excluded rows contribute 0 to the sum and the
            // maximum; the minimum is kept as the maximum of sel - x (= ~x for a row that counts, 0 for one that does not)
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)m[v >> 5], v & 31, 1);
                const uint32_t x = xs[v] & sel;
                if constexpr (C <= 24)
                    s32 += x;
                else
                    s64 += x;
                mx = x > mx ? x : mx;
                const uint32_t z = sel - x;
                nmn = z > nmn ? z : nmn;
            }
        }
        sum += C <= 24 ? (unsigned long long)s32 : s64;
#pragma unroll
        for (int j = 0; j < WORDS; j++) cnt += __builtin_popcount(m[j]);
        tile = next;
    }
    // wave totals (once per wave): 64-bit sums by 24-bit limbs through the DPP scan, min / max by a butterfly
    const unsigned long long lo = wave_sum((uint32_t)(sum & 0xffffffull)), mid = wave_sum((uint32_t)((sum >> 24) & 0xffffffull)),
                             hi = wave_sum((uint32_t)((sum >> 48) & 0xffffull));
    // (a lane's sum is below 2^58 -- 2^32 values x 2^26 lanes' worth is far beyond any column -- so 16 bits of hi suffice)
    const unsigned long long wsum = lo + (mid << 24) + (hi << 48);
    const unsigned long long wcnt = (unsigned long long)wave_sum((uint32_t)(cnt & 0xffffffull)) + ((unsigned long long)wave_sum((uint32_t)(cnt >> 24)) << 24);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t omn = (uint32_t)__shfl_xor((int)nmn, o, 64), omx = (uint32_t)__shfl_xor((int)mx, o, 64);
        nmn = omn > nmn ? omn : nmn;
        mx = omx > mx ? omx : mx;
    }
    const uint32_t mn = ~nmn;
    if (lane == 0 && wcnt) {
        __hip_atomic_fetch_add(a.out + 0, wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.out + 1, wcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_min(a.out + 2, (unsigned long long)mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(a.out + 3, (unsigned long long)mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int C> inline void launch_aggregate(const AggArgs &a, int num_cus, hipStream_t stream)
{
    constexpr int VPL = scan_vpl(C, kModeEq);
    using G = ScanGeom<C, VPL>;
    const uint64_t ntiles = (a.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
    // one 4-wave block per CU keeps ~36-48 KiB of DMA in flight at c = 9, as the scans do; small tiles take two
    const uint64_t blocks_wanted = (uint64_t)num_cus * (G::TILE_BYTES < 6144 ? 2 : 1);
    const uint64_t blocks_needed = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned grid = (unsigned)(blocks_needed < blocks_wanted ? (blocks_needed ? blocks_needed : 1) : blocks_wanted);
    hipLaunchKernelGGL(aggregate_init_kernel, dim3(1), dim3(1), 0, stream, a.out);
    hipLaunchKernelGGL((aggregate_kernel<C, VPL>), dim3(grid), dim3(kBlockThreads), 0, stream, a);
}

inline bool launch_aggregate_width(unsigned c, const AggArgs &a, int num_cus, hipStream_t stream)
{
    switch (c) {
#define MI355_AGG_CASE(W) case W: launch_aggregate<W>(a, num_cus, stream); return true;
        MI355_AGG_CASE(1) MI355_AGG_CASE(2) MI355_AGG_CASE(3) MI355_AGG_CASE(4) MI355_AGG_CASE(5) MI355_AGG_CASE(6) MI355_AGG_CASE(7) MI355_AGG_CASE(8)
        MI355_AGG_CASE(9) MI355_AGG_CASE(10) MI355_AGG_CASE(11) MI355_AGG_CASE(12) MI355_AGG_CASE(13) MI355_AGG_CASE(14) MI355_AGG_CASE(15) MI355_AGG_CASE(16)
        MI355_AGG_CASE(17) MI355_AGG_CASE(18) MI355_AGG_CASE(19) MI355_AGG_CASE(20) MI355_AGG_CASE(21) MI355_AGG_CASE(22) MI355_AGG_CASE(23) MI355_AGG_CASE(24)
        MI355_AGG_CASE(25) MI355_AGG_CASE(26) MI355_AGG_CASE(27) MI355_AGG_CASE(28) MI355_AGG_CASE(29) MI355_AGG_CASE(30) MI355_AGG_CASE(31) MI355_AGG_CASE(32)
#undef MI355_AGG_CASE
    default: return false;
    }
}

} // namespace mi355

// extras/gather.hpp -- values of a packed column at given row ids (the step after a selection vector: "take").
// Not one of the profiled hot-path kernels (kernels/*.hpp): a consumer either side of the scan, included by
// capi.hip only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi355 {

struct GatherArgs {
    const uint8_t *packed; // n values of c bits, LSB first (src/simd_scan_compression.cpp:53-104), readable 8 bytes past any value
    uint64_t n;
    uint32_t c;
    uint64_t first_row;         // row id of value 0
    const uint64_t *rowids;     // ids to fetch (any order, duplicates allowed)
    const uint64_t *count_dev;  // number of ids, read on the device (the count a selection left there) ...
    uint64_t capacity;          // ... capped by this
    int32_t *out;               // out[i] = value of row rowids[i]; -1 for an id outside [first_row, first_row + n)
};

// One id per lane: the value's two dwords (bit offset (id - first_row) * c; a value never spans more than two), one
// v_alignbit_b32, one mask.  Ascending ids -- what a selection vector holds -- make neighbouring lanes read neighbouring
// dwords; nothing is staged.
static __global__ __launch_bounds__(256) void gather_kernel(GatherArgs g)
{
    uint64_t count = *g.count_dev;
    if (count > g.capacity) count = g.capacity;
    const uint32_t mask = g.c >= 32 ? 0xffffffffu : ((1u << g.c) - 1u);
    const uint32_t *const words = (const uint32_t *)g.packed;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = g.rowids[i] - g.first_row; // (an id below first_row wraps to a huge r: rejected below)
        int32_t v = -1;
        if (r < g.n) {
            const uint64_t bit = r * g.c;
            const uint64_t w = bit >> 5;
            const uint32_t lo = words[w], hi = words[w + 1];
            v = (int32_t)(__builtin_amdgcn_alignbit(hi, lo, (uint32_t)bit & 31u) & mask);
        }
        g.out[i] = v;
    }
}

} // namespace mi355

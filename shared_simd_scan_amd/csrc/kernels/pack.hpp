// kernels/pack.hpp -- pack_kernel: packer and synthetic column generators.  Part of kernels.hpp (gfx950 only).
#pragma once

#include "tile.hpp"

namespace mi355 {

// ---- packer / synthetic column generator ----------------------------------------------------
// One thread per OUTPUT dword D.  32 values occupy exactly C dwords, so with G = D / C, r = D % C
// the dword holds bits [32r, 32r+32) of group G: values k = floor(32r/C) .. floor((32r+31)/C) of the
// group (k <= 31), each shifted to its place.  Values are masked to C bits.
enum PackSource { kSrcU16 = 0, kSrcU32 = 1, kSrcMod = 2, kSrcSplitmix = 3, kSrcIndex = 4 };

struct PackArgs {
    const void *values; // kSrcU16 / kSrcU32
    uint64_t n;
    uint64_t first_row; // generators
    uint64_t param;     // modulus or seed
    uint32_t *out;
    uint64_t out_dwords; // ceil(compressed_buffer_size / 4): payload + zero pad
    uint32_t c;
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// MAXK > 0 (packing from an array): an output dword draws on at most MAXK = floor(31/c) + 2 values; their loads are
// issued together (predicated) instead of one per loop trip -- the trip-by-trip form was bound by load latency.
template <int SRC, int MAXK = 0> __global__ __launch_bounds__(256) void pack_kernel(PackArgs a)
{
    const uint32_t c = a.c;
    const uint32_t mask = c == 32 ? 0xffffffffu : ((1u << c) - 1u);
    const uint64_t gstride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t D = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (D >= a.out_dwords) return;
    // (group, dword-in-group) of the thread's first dword by one division; later dwords advance incrementally.
    // n / c for n < 1024 and c <= 32 is exactly (n * M) >> 16 with M = ceil(2^16 / c): no division in the loop.
    uint64_t grp = D / c;
    uint32_t r = (uint32_t)(D - grp * c);
    const uint64_t step_grp = gstride / c;
    const uint32_t step_r = (uint32_t)(gstride - step_grp * c);
    const uint32_t M = (65536u + c - 1u) / c;
    auto advance = [&](uint64_t &g_, uint32_t &r_) {
        g_ += step_grp;
        r_ += step_r;
        if (r_ >= c) {
            r_ -= c;
            g_++;
        }
    };
    if constexpr (MAXK > 0) {
        static_assert(SRC == kSrcU16 || SRC == kSrcU32, "batched loads are for array sources");
        if (a.n == 0) { // only the pad: nothing to read
            for (; D < a.out_dwords; D += gstride) a.out[D] = 0;
            return;
        }
        // dword r_ of group g_: the (predicated) loads of its <= MAXK values are issued together
        auto word_of = [&](uint64_t g_, uint32_t r_) {
            const uint32_t lo_bit = 32 * r_;
            const uint32_t k0 = (lo_bit * M) >> 16;
            uint32_t k1 = ((lo_bit + 31) * M) >> 16;
            k1 = k1 < 31 ? k1 : 31;
            uint32_t vals[MAXK];
#pragma unroll
            for (int j = 0; j < MAXK; j++) {
                const uint32_t k = k0 + j;
                const uint64_t i = g_ * 32 + k;
                uint32_t v = 0;
                if (k <= k1 && i < a.n) v = SRC == kSrcU16 ? (uint32_t)((const uint16_t *)a.values)[i] : ((const uint32_t *)a.values)[i];
                vals[j] = v & mask;
            }
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < MAXK; j++) {
                const int32_t pos = (int32_t)((k0 + j) * c) - (int32_t)lo_bit; // bit position inside this dword
                word |= pos >= 0 ? (vals[j] << (pos & 31)) : (vals[j] >> ((-pos) & 31)); // vals[j] == 0 when unused
            }
            return word;
        };
        // (two dwords per iteration with unconditional clamped loads was measured: 1.59 ms against 1.28 ms per 1e9 values)
        for (; D < a.out_dwords; D += gstride) {
            a.out[D] = word_of(grp, r);
            advance(grp, r);
        }
    } else {
        for (; D < a.out_dwords; D += gstride) {
            const uint32_t lo_bit = 32 * r;
            const uint32_t k0 = (lo_bit * M) >> 16;
            uint32_t k1 = ((lo_bit + 31) * M) >> 16;
            k1 = k1 < 31 ? k1 : 31;
            uint32_t word = 0;
            for (uint32_t k = k0; k <= k1; k++) {
                const uint64_t i = grp * 32 + k;
                if (i >= a.n) break;
                uint32_t v;
                if constexpr (SRC == kSrcU16)
                    v = ((const uint16_t *)a.values)[i];
                else if constexpr (SRC == kSrcU32)
                    v = ((const uint32_t *)a.values)[i];
                else if constexpr (SRC == kSrcMod)
                    v = (uint32_t)((a.first_row + i) % a.param);
                else if constexpr (SRC == kSrcSplitmix)
                    v = (uint32_t)splitmix64(a.param, a.first_row + i);
                else
                    v = (uint32_t)(a.first_row + i);
                v &= mask;
                const int32_t pos = (int32_t)(k * c) - (int32_t)lo_bit; // bit position inside this dword
                word |= pos >= 0 ? (v << pos) : (v >> (-pos));
            }
            a.out[D] = word;
            advance(grp, r);
        }
    }
}

} // namespace mi355

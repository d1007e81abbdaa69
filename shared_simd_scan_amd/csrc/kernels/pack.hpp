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

template <int SRC> __global__ __launch_bounds__(256) void pack_kernel(PackArgs a)
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

// ---- tiled packer for array sources -------------------------------------------------------------------
// pack_kernel reads each source value straight from global memory: the ~32/c + 1 values of an output dword are
// fetched by ~32/c + 1 separate wave-loads whose lanes sit 32/c values apart (c = 9: five loads spanning the same
// eight cache lines).  Here a block stages a tile of 8192 values in LDS with fully coalesced 16-byte loads (masked to
// c bits, widened to 32), then every thread assembles output dwords from LDS; a tile is 256*c output dwords, so tiles
// start on a dword (and group) boundary.  The last tile also writes the zero pad behind the payload.
constexpr int kPackTile = 8192;

template <int SRC, int MAXK> __global__ __launch_bounds__(256) void pack_tiled_kernel(PackArgs a)
{
    static_assert(SRC == kSrcU16 || SRC == kSrcU32, "array sources only");
    __shared__ __attribute__((aligned(16))) uint32_t vals[kPackTile];
    const uint32_t c = a.c;
    const uint32_t mask = c == 32 ? 0xffffffffu : ((1u << c) - 1u);
    const uint32_t M = (uint32_t)((0x100000000ull + c - 1) / c); // n / c == umulhi(n, M) for n < 2^23 (n <= 32 * 8192 here)
    const uint64_t ntiles = a.n ? (a.n + kPackTile - 1) / kPackTile : 1;
    const uint64_t tile_dwords = 256ull * c;
    const bool aligned16 = ((uintptr_t)a.values & 15) == 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = tile * kPackTile;
        const uint32_t cnt = a.n - base < (uint64_t)kPackTile ? (uint32_t)(a.n - base) : (uint32_t)kPackTile;
        if (cnt == (uint32_t)kPackTile && aligned16) {
            if constexpr (SRC == kSrcU32) {
                const u32x4 *src = (const u32x4 *)((const uint32_t *)a.values + base);
#pragma unroll
                for (int j = 0; j < kPackTile / 4 / 256; j++) {
                    u32x4 v = __builtin_nontemporal_load(src + j * 256 + threadIdx.x);
                    v.x &= mask; v.y &= mask; v.z &= mask; v.w &= mask;
                    ((u32x4 *)vals)[j * 256 + threadIdx.x] = v;
                }
            } else {
                const u32x4 *src = (const u32x4 *)((const uint16_t *)a.values + base);
#pragma unroll
                for (int j = 0; j < kPackTile / 8 / 256; j++) {
                    const u32x4 v = __builtin_nontemporal_load(src + j * 256 + threadIdx.x); // 8 x u16
                    u32x4 lo = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
                    u32x4 hi = {v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16};
                    lo.x &= mask; lo.y &= mask; lo.z &= mask; lo.w &= mask;
                    hi.x &= mask; hi.y &= mask; hi.z &= mask; hi.w &= mask;
                    ((u32x4 *)vals)[(j * 256 + threadIdx.x) * 2] = lo;
                    ((u32x4 *)vals)[(j * 256 + threadIdx.x) * 2 + 1] = hi;
                }
            }
        } else {
            for (uint32_t i = threadIdx.x; i < (uint32_t)kPackTile; i += 256) {
                uint32_t v = 0;
                if (i < cnt) v = SRC == kSrcU16 ? (uint32_t)((const uint16_t *)a.values)[base + i] : ((const uint32_t *)a.values)[base + i];
                vals[i] = v & mask;
            }
        }
        __syncthreads();
        // this tile's output dwords; the last tile continues into the pad (everything behind value n is zero)
        const uint64_t first = tile * tile_dwords;
        const uint64_t nd = (tile + 1 == ntiles) ? a.out_dwords - first : tile_dwords;
        for (uint64_t D = threadIdx.x; D < nd; D += 256) {
            uint32_t word = 0;
            if (D < tile_dwords) {
                const uint32_t lo_bit = 32 * (uint32_t)D;
                const uint32_t k0 = c == 1 ? lo_bit : __umulhi(lo_bit, M); // (c == 1: M = 2^32 does not fit)
                uint32_t k1 = c == 1 ? lo_bit + 31 : __umulhi(lo_bit + 31, M);
                k1 = k1 < (uint32_t)kPackTile - 1 ? k1 : (uint32_t)kPackTile - 1;
                uint32_t v[MAXK];
#pragma unroll
                for (int j = 0; j < MAXK; j++) {
                    const uint32_t k = k0 + j;
                    v[j] = k <= k1 ? vals[k] : 0u;
                }
#pragma unroll
                for (int j = 0; j < MAXK; j++) {
                    const int32_t pos = (int32_t)((k0 + j) * c) - (int32_t)lo_bit; // bit position inside this dword
                    word |= pos >= 0 ? (v[j] << (pos & 31)) : (v[j] >> ((-pos) & 31)); // v[j] == 0 when unused
                }
            }
            a.out[first + D] = word;
        }
        __syncthreads();
    }
}

} // namespace mi355

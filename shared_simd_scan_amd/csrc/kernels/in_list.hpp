// kernels/in_list.hpp -- in_kernel: IN-list scan.  Part of kernels.hpp (gfx950 only).
#pragma once

#include "scan.hpp"

namespace mi355 {

// ---- IN-list scan: bitmap[i] = (value_i in {keys}) ----------------------------------------------------
// One result bitmap for a set of keys (the OR-reduction of a shared scan).
//   C <= 16: the set is a 2^C-bit bitset in LDS (<= 8 KiB) built by the block; one byte lookup + bit extract per value.
//   C  > 16: compare chain over the key list (device array, padded to 8): O(P) half-rate compares per value.
// Same tile / DMA / deferred-store skeleton as scan_burst_kernel with K = 1 (VPL from scan_vpl(C, kModeEq)); and_mask / invert apply.
template <int C, int AUX_, int VPL>
__global__ __launch_bounds__(kBlockThreads) void in_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // bitmap stores: 1 non-temporal, 2 write-through (sc1)
    constexpr bool BITSET = C <= 16;
    constexpr int SET_BYTES = BITSET ? ((1 << (C < 16 ? C : 16)) + 7) / 8 : 16;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t set_words[(SET_BYTES + 3) / 4];
    __shared__ __attribute__((aligned(16))) uint8_t mlds[kWavesPerBlock][1024]; // the tile's AND-mask bytes (LDS-DMA, with the tile)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    uint8_t *mlds_wave = mlds[wave];
    // the mask of a fused conjunction travels with the tile (in front of it, so one wait covers both): a plain load issued
    // where the mask is needed stalled every tile for a memory round trip (see scan_burst_kernel)
    auto issue_tile = [&](uint64_t t) {
        if (a.and_mask && t < tc.nfull) {
            if (lane * 16 < G::BITMAP_BYTES)
                __builtin_amdgcn_global_load_lds(MI355_GPTR(a.and_mask + t * G::BITMAP_BYTES + lane * 16), MI355_LPTR(mlds_wave), 16, 0, 0);
        }
        tc.template issue<AUX>(a.packed, t, lds_wave, lane);
    };

    if (tile < tc.ntiles) issue_tile(tile);
    if constexpr (BITSET) {
        for (uint32_t i = threadIdx.x; i < (SET_BYTES + 3) / 4; i += kBlockThreads) set_words[i] = 0;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
            const uint32_t key = (uint32_t)a.keys_dev[k];
            if ((key >> C) == 0) atomicOr(&set_words[key >> 5], 1u << (key & 31)); // out-of-range keys match nothing
        }
        __syncthreads();
    }
    const uint8_t *set_bytes = (const uint8_t *)set_words;

    uint32_t hits = 0;
    uint32_t res[WORDS];
    uint64_t prev = ~0ull;
    uint8_t *const out_lane = a.out + lane * (WORDS * 4);
    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        uint32_t mcur[WORDS];
        if (a.and_mask && tile < tc.nfull) {
#pragma unroll
            for (int j = 0; j < WORDS; j++) mcur[j] = ((const uint32_t *)(mlds_wave + lane * (WORDS * 4)))[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (prev != ~0ull) store_words<WORDS, NTS>(out_lane + prev * G::BITMAP_BYTES, res);
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) issue_tile(next);

        uint32_t xs[VPL];
        extract_all<C, VPL, 0, G::LANE_DWORDS>(w, xs);
        if constexpr (BITSET) {
#pragma unroll
            for (int j = 0; j < WORDS; j++) {
                uint32_t acc = 0;
#pragma unroll
                for (int k = 31; k >= 0; k--) { // value 32j+0 ends in bit 0
                    const uint32_t x = xs[32 * j + k];
                    const uint32_t bit = (set_bytes[x >> 3] >> (x & 7)) & 1u;
                    acc = (acc << 1) | bit;
                }
                res[j] = acc;
            }
        } else {
            uint32_t m[WORDS];
#pragma unroll
            for (int j = 0; j < WORDS; j++) m[j] = 0;
            for (uint32_t k = 0; k < P; k++) {
                const uint32_t key = __builtin_amdgcn_readfirstlane((uint32_t)a.keys_dev[k]);
#pragma unroll
                for (int j = 0; j < WORDS; j++) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int i = 31; i >= 0; i--) acc = (acc << 1) | (xs[32 * j + i] == key ? 1u : 0u);
                    m[j] |= acc;
                }
            }
#pragma unroll
            for (int j = 0; j < WORDS; j++) res[j] = m[j];
        }
        const uint32_t inv = a.invert;
#pragma unroll
        for (int j = 0; j < WORDS; j++) res[j] ^= inv;
        if (tile < tc.nfull) {
            if (a.and_mask) {
#pragma unroll
                for (int j = 0; j < WORDS; j++) res[j] &= mcur[j];
            }
#pragma unroll
            for (int j = 0; j < WORDS; j++) hits += __builtin_popcount(res[j]);
            prev = tile;
        } else {
            const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
            const int nbytes = left <= 0 ? 0 : (int)((left >= VPL ? VPL : left) + 7) / 8;
            if (a.and_mask) {
                const uint8_t *mp = a.and_mask + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
#pragma unroll
                for (int j = 0; j < WORDS; j++) {
                    uint32_t mm = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (4 * j + b < nbytes) mm |= (uint32_t)mp[4 * j + b] << (8 * b);
                    res[j] &= mm;
                }
            }
            hits += tc.finish_tail(tile, res, out_lane + tile * G::BITMAP_BYTES, 1, lane);
            prev = ~0ull;
        }
        tile = next;
    }
    if (prev != ~0ull) store_words<WORDS, NTS>(out_lane + prev * G::BITMAP_BYTES, res);
    if (a.hits) hits_add(a, 0, wave_sum(hits), lane);
    hits_finalize(a, 1, lane);
}

} // namespace mi355

// kernels/decompress.hpp -- decompress_kernel: packed -> int32.  Part of kernels.hpp (gfx950 only).
#pragma once

#include "tile.hpp"

namespace mi355 {

// ---- decompression to int32 -------------------------------------------------------------------
// Lane-per-value: in a step of 256 values lane l owns values 4l..4l+3, so its output is one
// 16-byte store and a wave-instruction writes 1 KiB contiguous.  The bit position of value
// (step*256 + 4l + j) is step*256C + (4l+j)C: the per-lane part is step-invariant (the same
// periodicity the reference exploits every 8 values, src/simd_scan_commons.hpp:5-16), so each lane
// keeps 4 (dword index, shift) pairs; a step costs 4 x { two-dword LDS read, v_alignbit_b32, v_and }.
template <int C> struct DecompGeom {
    static constexpr int TILE_VALUES = 4096;            // 16 steps of 256
    static constexpr int TILE_BYTES = TILE_VALUES * C / 8; // 512C
    static constexpr int DMA_INSTRS = (TILE_BYTES + 1023) / 1024;
    static constexpr int LDS_BYTES = DMA_INSTRS * 1024 + 16; // +16: the hi dword of the last value
    static constexpr int STEPS = TILE_VALUES / 256;
};

struct DecompArgs {
    const uint8_t *packed;
    uint64_t n;
    int32_t *out;
};

template <int C, int AUX_>
__global__ __launch_bounds__(kBlockThreads) void decompress_kernel(DecompArgs a)
{
    using G = DecompGeom<C>;
    constexpr int AUX = AUX_ & 15;          // cache policy of the DMA loads
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // stores of the int32 output: 1 non-temporal, 2 write-through (sc1)
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const uint32_t *lds32 = (const uint32_t *)lds_wave;

    const uint64_t n = a.n;
    const uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;
    const uint64_t nfull = n / G::TILE_VALUES;
    const uint64_t data_bytes = (n * C + 7) / 8;
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    constexpr uint32_t mask = C == 32 ? 0xffffffffu : ((1u << C) - 1u);

    uint32_t didx[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t bit = (4 * lane + j) * C;
        didx[j] = bit >> 5;
        sh[j] = bit & 31;
    }
    if (lane == 0) *(uint32_t *)(lds_wave + G::DMA_INSTRS * 1024) = 0; // hi dword past the tile

    for (uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += stride) {
        const uint8_t *src = a.packed + tile * G::TILE_BYTES;
        const uint64_t bytes_left = data_bytes - tile * G::TILE_BYTES;
        // WAR: the previous tile's LDS reads are complete (their results were stored)
#pragma unroll
        for (int j = 0; j < G::DMA_INSTRS; j++) {
            uint32_t o = j * 1024 + lane * 16;
            if (o < G::TILE_BYTES && o < bytes_left)
                __builtin_amdgcn_global_load_lds(MI355_GPTR(src + o), MI355_LPTR(lds_wave + j * 1024), 16, 0, AUX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        int32_t *dst = a.out + tile * G::TILE_VALUES + lane * 4;
        if (tile < nfull) {
#pragma unroll
            for (int s = 0; s < G::STEPS; s++) {
                u32x4 v;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t lo = lds32[s * 8 * C + didx[j]];
                    uint32_t hi = lds32[s * 8 * C + didx[j] + 1];
                    v[j] = __builtin_amdgcn_alignbit(hi, lo, sh[j]) & mask;
                }
                if constexpr (NTS == 2)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + s * 256), "v"(v) : "memory");
                else if constexpr (NTS == 1)
                    __builtin_nontemporal_store(v, (u32x4 *)(dst + s * 256));
                else
                    *(u32x4 *)(dst + s * 256) = v;
            }
        } else {
            const uint64_t base = tile * G::TILE_VALUES;
            for (int s = 0; s < G::STEPS; s++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint64_t i = base + s * 256 + lane * 4 + j;
                    if (i < n) {
                        uint32_t lo = lds32[s * 8 * C + didx[j]];
                        uint32_t hi = lds32[s * 8 * C + didx[j] + 1];
                        a.out[i] = (int32_t)(__builtin_amdgcn_alignbit(hi, lo, sh[j]) & mask);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

} // namespace mi355

// tools/scan_kernel_r1.hpp -- round 1's scan kernel with its ABL ablation / diagnostic template knobs and the DEPTH = 2
// software pipeline.  TOOL-ONLY: instantiated by tools/tune_scan.hip and tools/ceilings.hip for ablations and A/B
// experiments (DESIGN.md 3.1 "Measured choices"); libmi355scan.so launches scan_burst_kernel (kernels/scan.hpp) and
// nothing in shared_simd_scan_amd/ includes this file.  Include after the product's kernels.hpp.
#pragma once

namespace mi355 {

// MODE kModeEq / kModeRange: one bitmap.  MODE kModeShared: up to 8 keys, one bitmap per key at out + k*out_stride
// (one decode, 8 compares per value).
// AUX_: bits 0-3 = cache policy of the DMA loads (0 default, 2 nt); bit 4 = non-temporal bitmap stores.
// ABL: 1 = DMA only, 2 = DMA + LDS reads, 3 = no bitmap stores, 4 = normal + clock / placement stamps (written through
// a.keys_dev: a debug buffer of 4 x gridDim.x uint64), 5 = XCD-contiguous tile mapping.
// DEPTH: tiles of DMA in flight per wave ahead of the one being decoded (1: one LDS buffer per wave; 2: two buffers,
// the wait for tile t is `vmcnt(DMA_INSTRS)` = everything older than the DMA of tile t+1).
template <int C, int MODE, int AUX_, int VPL, int ABL = 0, int DEPTH = 1>
__global__ __launch_bounds__(kBlockThreads, (DEPTH == 1 ? scan_occ<C, VPL, MODE>() : 1)) void scan_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int NK = (MODE == kModeShared) ? kMaxKeysPerPass : 1;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0);
    static_assert(DEPTH == 1 || DEPTH == 2, "DEPTH");
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][DEPTH][G::LDS_BYTES];
    constexpr int LK = (MODE != kModeShared && ABL == 0) ? narrow_k<C>() : 0; // values per table lookup (0: compare chain)
    __shared__ __attribute__((aligned(16))) uint8_t nlut[LK ? (1 << (LK * C)) : 16];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave][0];
    TileCtx<C, VPL> tc(a.n);
    uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    if constexpr (ABL == 5) {
        // experiment (tools/tune_scan.hip): XCD-contiguous mapping -- blocks are dealt round-robin over the 8 XCDs,
        // so give XCD x the x-th eighth of the tiles instead of every 8th block-row
        const uint64_t per = (tc.ntiles + 7) / 8;
        const uint64_t x = blockIdx.x & 7;
        stride = (uint64_t)(gridDim.x / 8) * kWavesPerBlock;
        tile = x * per + (uint64_t)(blockIdx.x / 8) * kWavesPerBlock + wave;
        const uint64_t lim = (x + 1) * per < tc.ntiles ? (x + 1) * per : tc.ntiles;
        tc.ntiles = tile < lim ? lim : tile; // this wave's range ends at its XCD's slice
    }
    const uint32_t P = (MODE == kModeShared) ? a.nkeys : 1;

    unsigned long long stamp_c0 = 0, stamp_r0 = 0;
    if constexpr (ABL == 4) { // diagnostic build: shader clock = d(memtime)/d(memrealtime) x 100 MHz
        stamp_c0 = __builtin_amdgcn_s_memtime();
        stamp_r0 = __builtin_amdgcn_s_memrealtime();
    }

    uint32_t key[kMaxKeysPerPass];
#pragma unroll
    for (int q = 0; q < kMaxKeysPerPass; q++) key[q] = a.key[q];
    uint32_t hits[NK];
#pragma unroll
    for (int q = 0; q < NK; q++) hits[q] = 0;

    // lane's byte offset inside a tile's bitmap, and the per-key bitmap bases
    uint8_t *const out_lane = a.out + lane * (WORDS * 4);
    const uint64_t kstride = a.out_stride;

    uint32_t res[NK][WORDS];
    uint64_t prev = ~0ull; // tile whose results sit in `res`, not yet stored
    const bool store = a.out != nullptr; // null: count-only scan (the read stream alone)
    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    if constexpr (DEPTH == 2) {
        if (tile + stride < tc.ntiles) tc.template issue<AUX>(a.packed, tile + stride, lds_wave + G::LDS_BYTES, lane);
    }
    if constexpr (LK > 0) {
        // predicate table: bit j of entry e = predicate(field j of e); the same formulas as push1/2/4
        constexpr uint32_t fmask = (1u << C) - 1u;
        for (uint32_t e = threadIdx.x; e < (1u << (LK * C)); e += kBlockThreads) {
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < LK; j++) {
                const uint32_t f = (e >> (j * C)) & fmask;
                const bool hit = (MODE == kModeRange) ? (f - key[0]) <= key[1] : f == key[0];
                m |= (hit ? 1u : 0u) << j;
            }
            nlut[e] = (uint8_t)m;
        }
        __syncthreads();
    }
    uint32_t parity = 0; // DEPTH 2: which of the wave's two LDS buffers holds the current tile
    while (tile < tc.ntiles) {
        uint8_t *cur = lds_wave;
        if constexpr (DEPTH == 2) {
            cur = lds_wave + parity * G::LDS_BYTES;
            // tile t+1's DMA (DMA_INSTRS instructions, all issued: it is a full tile) may stay in flight
            if (tile + stride < tc.nfull)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::DMA_INSTRS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint32_t w[G::LANE_DWORDS];
        if constexpr (ABL != 1) read_lane_data<C, VPL>(cur, lane, w);
        // the LDS tile must be fully read before the next DMA may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (prev != ~0ull && store) { // every tile but a wave's last is a full tile
            if constexpr (ABL == 0 || ABL >= 4) {
                uint8_t *dst = out_lane + prev * G::BITMAP_BYTES;
#pragma unroll
                for (int q = 0; q < NK; q++) {
                    if ((uint32_t)q < P) store_words<WORDS, NTS>(dst, res[q]);
                    dst += kstride;
                }
            }
        }
        const uint64_t next = tile + stride;
        if constexpr (DEPTH == 2) {
            if (next + stride < tc.ntiles) tc.template issue<AUX>(a.packed, next + stride, cur, lane);
            parity ^= 1;
        } else {
            if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        }

        if constexpr (ABL == 1) {
#pragma unroll
            for (int q = 0; q < NK; q++)
#pragma unroll
                for (int j = 0; j < WORDS; j++) res[q][j] = 0;
        } else if constexpr (ABL == 2) {
            uint32_t x = 0;
#pragma unroll
            for (int i = 0; i < G::LANE_DWORDS; i++) x ^= w[i];
#pragma unroll
            for (int q = 0; q < NK; q++)
#pragma unroll
                for (int j = 0; j < WORDS; j++) res[q][j] = x;
        } else if constexpr (LK > 0) {
            decode_words_narrow<C, VPL, LK, 0, G::LANE_DWORDS>(w, res, nlut);
        } else {
            decode_words<C, VPL, 0, NK, MODE, G::LANE_DWORDS>(w, res, key);
        }
        if constexpr (MODE != kModeShared) {
            // negation (!=, NOT BETWEEN) and conjunction with an earlier predicate's bitmap, fused into the scan
            const uint32_t inv = a.invert;
#pragma unroll
            for (int j = 0; j < WORDS; j++) res[0][j] ^= inv;
            if (a.and_mask) {
                const uint8_t *mp = a.and_mask + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                const uint32_t mop = a.mask_op;
                auto combine = [mop](uint32_t r, uint32_t m) -> uint32_t {
                    return mop == 0 ? (r & m) : mop == 1 ? (r | m) : mop == 2 ? (r ^ m) : (m & ~r);
                };
                if (tile < tc.nfull) {
#pragma unroll
                    for (int j = 0; j < WORDS; j++) res[0][j] = combine(res[0][j], ((const uint32_t *)mp)[j]);
                } else { // tail tile: read only the bytes the mask is guaranteed to hold (ceil(n/8))
                    const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
                    const int nbytes = left <= 0 ? 0 : (int)((left >= VPL ? VPL : left) + 7) / 8;
#pragma unroll
                    for (int j = 0; j < WORDS; j++) {
                        uint32_t m = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            if (4 * j + b < nbytes) m |= (uint32_t)mp[4 * j + b] << (8 * b);
                        res[0][j] = combine(res[0][j], m);
                    }
                }
            }
        }
        if (tile < tc.nfull) {
#pragma unroll
            for (int q = 0; q < NK; q++)
#pragma unroll
                for (int j = 0; j < WORDS; j++) hits[q] += __builtin_popcount(res[q][j]);
            prev = tile;
        } else {
            uint8_t *dst = out_lane + tile * G::BITMAP_BYTES;
#pragma unroll
            for (int q = 0; q < NK; q++) {
                if ((uint32_t)q < P) hits[q] += tc.finish_tail(tile, res[q], dst, 1, lane, store);
                dst += kstride;
            }
            prev = ~0ull;
        }
        tile = next;
    }
    if (prev != ~0ull && store) {
        if constexpr (ABL == 0 || ABL >= 4) {
            uint8_t *dst = out_lane + prev * G::BITMAP_BYTES;
#pragma unroll
            for (int q = 0; q < NK; q++) {
                if ((uint32_t)q < P) store_words<WORDS, NTS>(dst, res[q]);
                dst += kstride;
            }
        } else if (res[0][0] == 0x12345678u) { // keep the ablated pipeline alive
            a.out[lane] = 1;
        }
    }
    if (a.hits) {
#pragma unroll
        for (int q = 0; q < NK; q++) {
            uint32_t s = wave_sum(hits[q]);
            if ((uint32_t)q < P) hits_add(a, q, s, lane);
        }
    }
    hits_finalize(a, P, lane);

    if constexpr (ABL == 4) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.scratch[kScratchDone + 2] = __builtin_amdgcn_s_memtime() - stamp_c0;
            a.scratch[kScratchDone + 3] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        }
        if (threadIdx.x == 0 && a.keys_dev) { // per-block record: start, end (100 MHz ticks), HW_ID, XCC_ID
            unsigned long long *dbg = (unsigned long long *)a.keys_dev + (uint64_t)blockIdx.x * 4;
            dbg[0] = stamp_r0;
            dbg[1] = __builtin_amdgcn_s_memrealtime();
            dbg[2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
            dbg[3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        }
    }
}

} // namespace mi355

// kernels/scan.hpp -- scan_burst_kernel / scan2_kernel: equality / range scan (+ negation, + fused mask).  Part of kernels.hpp (gfx950 only).
#pragma once

#include "tile.hpp"

namespace mi355 {

// ---- tile bookkeeping shared by every kernel that walks wave tiles ---------------------------------------------
// Per wave, per tile:  wait for the tile's DMA -> ds_read the lane's run into VGPRs -> (LDS is free)
// store EARLIER results, then issue the NEXT tile's DMA -> decode/compare in registers.
// Stores are issued before the DMA that the next iteration waits for, so a plain vmcnt(0) never waits
// on a store younger than the data it needs, whatever the number of stores per tile is; the DMA of
// tile t+1 is in flight during the whole compute phase of tile t.
// AUX_: bits 0-3 = cache policy of the DMA loads (0 default, 2 nt); bit 4 = non-temporal bitmap stores, bit 5 = write-through.
template <int C, int VPL> struct TileCtx {
    using G = ScanGeom<C, VPL>;
    uint64_t n, ntiles, nfull, data_bytes;
    __device__ __forceinline__ TileCtx(uint64_t n_) : n(n_)
    {
        ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        nfull = n / G::TILE_VALUES;
        data_bytes = (n * C + 7) / 8;
    }
    template <int AUX> __device__ __forceinline__ void issue(const uint8_t *packed, uint64_t t, uint8_t *lds_wave, int lane) const
    {
        const uint8_t *src = packed + t * G::TILE_BYTES;
        if (t < nfull)
            dma_tile_full<G::TILE_BYTES, AUX>(src, lds_wave, lane);
        else
            dma_tile_partial<G::TILE_BYTES, AUX>(src, data_bytes - t * G::TILE_BYTES, lds_wave, lane);
    }
    // tail tile: zero bits >= n, write exactly ceil(n/8) bytes of the tile's bitmap; returns the lane's hit count
    __device__ __forceinline__ uint32_t finish_tail(uint64_t t, uint32_t (&v)[VPL / 32], uint8_t *dst, uint64_t byte_stride, int lane,
                                                    bool store = true) const
    {
        const int64_t left = (int64_t)(n - t * G::TILE_VALUES) - (int64_t)lane * VPL;
        const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
        const int nbytes = (valid + 7) / 8;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < VPL / 32; j++) {
            v[j] &= tail_mask(valid, j);
            cnt += __builtin_popcount(v[j]);
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (store && 4 * j + b < nbytes) dst[(uint64_t)(4 * j + b) * byte_stride] = (uint8_t)(v[j] >> (8 * b));
        }
        return cnt;
    }
};

// ---- the scan kernel, burst form ----------------------------------------------------------------------------
// Tile pipeline (DMA of the next tile in flight while the current one is decoded; result stores
// issued after the LDS read and BEFORE the next DMA, so the vmcnt(0) that guards a DMA never waits for a younger store),
// with two differences:
//   * a wave owns CHUNKS of K consecutive tiles and keeps their K x WORDS bitmap words in registers: the chunk's
//     results leave as one burst of K store instructions = K KiB of contiguous bitmap per wave (4K KiB per block),
//     issued when the first tile of the wave's next chunk has been read from LDS.  K = 1: a store per tile (round 1's kernel, now tools/scan_kernel_r1.hpp).
//     (tools/burst.hip: a 9 : 1 read : write stream gains up to 6 % from 2-4 KiB bursts at one block per CU.)
//   * the mask bitmap of a fused conjunction / disjunction is fetched for the whole chunk at its start (K KiB
//     contiguous per wave), IN FRONT of the DMA of the next tile: the loads are older than every later DMA, so the
//     counted / top-of-loop waits cover them and no tile stalls for a memory round trip (a plain load issued where the
//     mask is needed cost 0.249 ms against 0.185 for the unmasked scan at 1e9 x 9 bit).  The mask travels by LDS-DMA
//     into a per-wave LDS image like the column itself, NOT into registers: a register load the compiler can see makes
//     it insert its own `s_waitcnt vmcnt(0)` in front of the first use -- right behind the result stores, draining them
//     every tile (+6 % on the UNMASKED scan) -- and one it cannot see (inline asm) may be copied to another register
//     before the data has arrived (seen at c <= 3: wrong bitmaps in 29 of 30 launches).
// a.out == nullptr: count-only scan.  MODE kModeEq / kModeRange only (shared scans have their own kernels).
// waves per SIMD the LDS footprint of scan_burst_kernel admits (tiles + mask image + narrow table), capped as scan_occ
template <int C, int VPL, int K> constexpr int burst_occ()
{
    using G = ScanGeom<C, VPL>;
    constexpr int mask_bytes = (K * G::BITMAP_BYTES + 1023) / 1024 * 1024;
    constexpr int table = narrow_k<C>() ? (1 << (narrow_k<C>() * C)) : 16;
    constexpr int lds = (160 * 1024) / (4 * (G::LDS_BYTES + mask_bytes) + table + 64);
    constexpr int cap = narrow_k<C>() ? 4 : 8;
    return lds > cap ? cap : (lds < 1 ? 1 : lds);
}

template <int C, int MODE, int AUX_, int VPL, int K>
__global__ __launch_bounds__(kBlockThreads, (burst_occ<C, VPL, K>())) void scan_burst_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    static_assert(MODE == kModeEq || MODE == kModeRange, "MODE");
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0);
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    constexpr int LK = narrow_k<C>(); // values per table lookup (0: compare chain)
    __shared__ __attribute__((aligned(16))) uint8_t nlut[LK ? (1 << (LK * C)) : 16];
    constexpr int MASK_DMA = (K * G::BITMAP_BYTES + 1023) / 1024; // LDS-DMA instructions per chunk of mask
    __shared__ __attribute__((aligned(16))) uint8_t mlds[kWavesPerBlock][MASK_DMA * 1024];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    uint8_t *mlds_wave = mlds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t nchunks = (tc.ntiles + K - 1) / K;
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t chunk = (uint64_t)blockIdx.x * kWavesPerBlock + wave;

    uint32_t key[2] = {a.key[0], a.key[1]};
    uint32_t hits = 0;
    uint8_t *const out_lane = a.out + lane * (WORDS * 4);
    const bool store = a.out != nullptr;
    const uint8_t *const mask = a.and_mask;
    const uint32_t mop = a.mask_op, inv = a.invert;
    auto combine = [mop](uint32_t r, uint32_t m) -> uint32_t {
        return mop == 0 ? (r & m) : mop == 1 ? (r | m) : mop == 2 ? (r ^ m) : (m & ~r);
    };

    uint32_t res[K][WORDS];   // results of the chunk being decoded; until its first tile is decoded: of the previous chunk
    uint64_t pend_chunk = 0;
    int pend_n = 0;           // full tiles of `pend_chunk` whose words still sit in `res`

    if (chunk < nchunks) tc.template issue<AUX>(a.packed, chunk * K, lds_wave, lane);
    if constexpr (LK > 0) {
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

    while (chunk < nchunks) {
        const uint64_t tfirst = chunk * K;
        int nfull_here = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint64_t tile = tfirst + k;
            if (tile < tc.ntiles) { // wave-uniform
                // the tile's DMA has landed (and, from the chunk's second tile on, all of its mask words)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t w[G::LANE_DWORDS];
                read_lane_data<C, VPL>(lds_wave, lane, w);
                uint32_t mcur[WORDS]; // the tile's mask words (k > 0: landed with the wait above; k == 0: read below)
                if (k > 0 && mask) {
#pragma unroll
                    for (int j = 0; j < WORDS; j++) mcur[j] = ((const uint32_t *)(mlds_wave + k * G::BITMAP_BYTES + lane * (WORDS * 4)))[j];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS tile fully read: the next DMA may overwrite it
                if (k == 0) {
                    if (pend_n && store) {
                        uint8_t *dst = out_lane + pend_chunk * K * G::BITMAP_BYTES;
#pragma unroll
                        for (int q = 0; q < K; q++)
                            if (q < pend_n) store_words<WORDS, NTS>(dst + q * G::BITMAP_BYTES, res[q]);
                    }
                    pend_n = 0;
                    if (mask) { // the chunk's mask bytes, full tiles only (the ragged tile reads bytes, below): one linear image
                        const uint64_t mbytes = (tc.nfull > tfirst ? (tc.nfull - tfirst < K ? tc.nfull - tfirst : K) : 0) * G::BITMAP_BYTES;
                        const uint8_t *src = mask + tfirst * G::BITMAP_BYTES;
#pragma unroll
                        for (int q = 0; q < MASK_DMA; q++) {
                            const uint32_t o = q * 1024 + lane * 16;
                            if (o < mbytes) __builtin_amdgcn_global_load_lds(MI355_GPTR(src + o), MI355_LPTR(mlds_wave + q * 1024), 16, 0, 0);
                        }
                    }
                }
                // next tile of this wave: the chunk's next tile, else the first tile of the wave's next chunk
                const uint64_t next = (k + 1 < K && tile + 1 < tc.ntiles) ? tile + 1 : (chunk + stride) * K;
                const bool have_next = next < tc.ntiles && (k + 1 < K || chunk + stride < nchunks);
                if (have_next) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
                uint32_t r1[1][WORDS];
                if constexpr (LK > 0) {
                    decode_words_narrow<C, VPL, LK, 0, G::LANE_DWORDS>(w, r1, nlut);
                } else {
                    const uint32_t key8[kMaxKeysPerPass] = {key[0], key[1], 0, 0, 0, 0, 0, 0};
                    decode_words<C, VPL, 0, 1, MODE, G::LANE_DWORDS>(w, r1, key8);
                }
#pragma unroll
                for (int j = 0; j < WORDS; j++) r1[0][j] ^= inv;
                if (tile < tc.nfull) {
                    if (mask) {
                        if (k == 0) {
                            // the chunk's mask DMAs are older than the tile DMA issued behind them: a full next tile is
                            // DMA_INSTRS instructions, anything else drains
                            if (have_next && next < tc.nfull)
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::DMA_INSTRS) : "memory");
                            else
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int j = 0; j < WORDS; j++) mcur[j] = ((const uint32_t *)(mlds_wave + lane * (WORDS * 4)))[j];
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
#pragma unroll
                        for (int j = 0; j < WORDS; j++) r1[0][j] = combine(r1[0][j], mcur[j]);
                    }
#pragma unroll
                    for (int j = 0; j < WORDS; j++) {
                        hits += __builtin_popcount(r1[0][j]);
                        res[k][j] = r1[0][j];
                    }
                    nfull_here = k + 1; // full tiles are a prefix of the chunk (the ragged tile is the column's last)
                } else {
                    if (mask) { // tail tile: read only the bytes the mask is guaranteed to hold (ceil(n/8))
                        const uint8_t *mp = mask + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                        const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
                        const int nbytes = left <= 0 ? 0 : (int)((left >= VPL ? VPL : left) + 7) / 8;
#pragma unroll
                        for (int j = 0; j < WORDS; j++) {
                            uint32_t m = 0;
#pragma unroll
                            for (int b = 0; b < 4; b++)
                                if (4 * j + b < nbytes) m |= (uint32_t)mp[4 * j + b] << (8 * b);
                            r1[0][j] = combine(r1[0][j], m);
                        }
                    }
                    hits += tc.finish_tail(tile, r1[0], out_lane + tile * G::BITMAP_BYTES, 1, lane, store);
                }
            }
        }
        pend_chunk = chunk;
        pend_n = nfull_here;
        chunk += stride;
    }
    if (pend_n && store) {
        uint8_t *dst = out_lane + pend_chunk * K * G::BITMAP_BYTES;
#pragma unroll
        for (int q = 0; q < K; q++)
            if (q < pend_n) store_words<WORDS, NTS>(dst + q * G::BITMAP_BYTES, res[q]);
    }
    if (a.hits) hits_add(a, 0, wave_sum(hits), lane);
    hits_finalize(a, 1, lane);
}

// ---- two columns, one launch: bitmap = COMBINE(p1(column 1), p2(column 2)) ------------------------------------------
// The conjunction / disjunction of predicates over two columns of the same width without the first predicate's bitmap
// ever going to HBM (the intent of src/simd_scan.hpp:76-84).  Per wave tile both columns' tiles are in
// flight together (two LDS buffers per wave), both are decoded in registers, the result words are combined and leave
// as in scan_burst_kernel with K = 1 (stores deferred one tile; out == nullptr: count only).  Both predicates are
// inclusive ranges with a negation word (every comparison of mi355_scan_where_dev is one).
template <int C, int AUX_, int VPL>
__global__ __launch_bounds__(kBlockThreads, (burst_occ<C, VPL, 1>() > 1 ? burst_occ<C, VPL, 1>() / 2 : 1)) void scan2_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0);
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][2][G::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds1 = lds[wave][0], *lds2 = lds[wave][1];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t k1[kMaxKeysPerPass] = {a.key[0], a.key[1], 0, 0, 0, 0, 0, 0};
    const uint32_t k2[kMaxKeysPerPass] = {a.key2[0], a.key2[1], 0, 0, 0, 0, 0, 0};
    const uint32_t inv1 = a.invert, inv2 = a.invert2, mop = a.mask_op;
    auto combine = [mop](uint32_t p2, uint32_t p1) -> uint32_t { // (as the mask forms: p1 plays the earlier bitmap)
        return mop == 0 ? (p2 & p1) : mop == 1 ? (p2 | p1) : mop == 2 ? (p2 ^ p1) : (p1 & ~p2);
    };
    uint32_t hits = 0;
    uint8_t *const out_lane = a.out + lane * (WORDS * 4);
    const bool store = a.out != nullptr;
    uint32_t res[WORDS];
    uint64_t prev = ~0ull;

    auto issue = [&](uint64_t t) {
        tc.template issue<AUX>(a.packed, t, lds1, lane);
        tc.template issue<AUX>(a.packed2, t, lds2, lane);
    };
    if (tile < tc.ntiles) issue(tile);
    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w1[G::LANE_DWORDS], w2[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds1, lane, w1);
        read_lane_data<C, VPL>(lds2, lane, w2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (prev != ~0ull && store) store_words<WORDS, NTS>(out_lane + prev * G::BITMAP_BYTES, res);
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) issue(next);
        uint32_t r1[1][WORDS], r2[1][WORDS];
        decode_words<C, VPL, 0, 1, kModeRange, G::LANE_DWORDS>(w1, r1, k1);
        decode_words<C, VPL, 0, 1, kModeRange, G::LANE_DWORDS>(w2, r2, k2);
#pragma unroll
        for (int j = 0; j < WORDS; j++) res[j] = combine(r2[0][j] ^ inv2, r1[0][j] ^ inv1);
        if (tile < tc.nfull) {
#pragma unroll
            for (int j = 0; j < WORDS; j++) hits += __builtin_popcount(res[j]);
            prev = tile;
        } else {
            hits += tc.finish_tail(tile, res, out_lane + tile * G::BITMAP_BYTES, 1, lane, store);
            prev = ~0ull;
        }
        tile = next;
    }
    if (prev != ~0ull && store) store_words<WORDS, NTS>(out_lane + prev * G::BITMAP_BYTES, res);
    if (a.hits) hits_add(a, 0, wave_sum(hits), lane);
    hits_finalize(a, 1, lane);
}

// 4x4 byte transpose: c[j] byte i = r[i] byte j   (v_perm_b32: selector 0-3 = bytes of the 2nd operand, 4-7 = 1st)
__device__ __forceinline__ void transpose4x4_bytes(const uint32_t (&r)[4], uint32_t (&c)[4])
{
    const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u); // r0.b0 r1.b0 r0.b1 r1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u); // r0.b2 r1.b2 r0.b3 r1.b3
    const uint32_t t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u);
    const uint32_t t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    c[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); // t0.b0 t0.b1 t2.b0 t2.b1
    c[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u); // t0.b2 t0.b3 t2.b2 t2.b3
    c[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    c[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

} // namespace mi355

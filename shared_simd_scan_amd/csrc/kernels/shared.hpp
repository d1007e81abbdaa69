// kernels/shared.hpp -- shared multi-predicate scans: compare chain, byte-entry LDS tables, dword-entry LDS tables.  Part of kernels.hpp (gfx950 only).
#pragma once

#include "scan.hpp"

namespace mi355 {

// ---- general shared scan: any P <= 1024 (ceil(P/8) passes of 8 keys over the lane's registers per tile),
// per-predicate or linear output (byte of 8-value group g and key k at g*P + k,
// src/simd_scan_shared_linear.cpp:57).  The tile's DMA is prefetched as above; results are stored pass by pass.
template <int C, int AUX_, int VPL>
__global__ __launch_bounds__(kBlockThreads) void shared_general_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int NK = kMaxKeysPerPass;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ uint32_t s_hits[kMaxKeys];
    for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxKeys; k += kBlockThreads) s_hits[k] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const bool keys_in_args = P <= (uint32_t)kMaxKeysPerPass;
    const uint32_t npass = (P + kMaxKeysPerPass - 1) / kMaxKeysPerPass;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        const bool full = tile < tc.nfull;

        for (uint32_t pass = 0; pass < npass; pass++) {
            uint32_t key[kMaxKeysPerPass];
            if (keys_in_args) {
#pragma unroll
                for (int q = 0; q < kMaxKeysPerPass; q++) key[q] = a.key[q];
            } else {
#pragma unroll
                for (int q = 0; q < kMaxKeysPerPass; q++)
                    key[q] = __builtin_amdgcn_readfirstlane((uint32_t)a.keys_dev[pass * kMaxKeysPerPass + q]);
            }
            uint32_t res[NK][WORDS];
            decode_words<C, VPL, 0, NK, kModeShared, G::LANE_DWORDS>(w, res, key);
            uint32_t cnts[8];
#pragma unroll
            for (int q = 0; q < NK; q++) {
                const uint32_t k = pass * kMaxKeysPerPass + q;
                cnts[q] = 0;
                if (k < P) {
                    uint32_t cnt = 0;
                    if (a.layout == 0) {
                        uint8_t *dst = a.out + (uint64_t)k * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                        if (full) {
#pragma unroll
                            for (int j = 0; j < WORDS; j++) cnt += __builtin_popcount(res[q][j]);
                            store_words<WORDS>(dst, res[q]);
                        } else {
                            cnt = tc.finish_tail(tile, res[q], dst, 1, lane);
                        }
                    } else if (!full) {
                        uint8_t *dst = a.out + (tile * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4)) * P + k;
                        cnt = tc.finish_tail(tile, res[q], dst, P, lane);
                    } else {
#pragma unroll
                        for (int j = 0; j < WORDS; j++) cnt += __builtin_popcount(res[q][j]);
                    }
                    cnts[q] = cnt;
                }
            }
            if (a.hits) block_hits_add8(s_hits, pass * kMaxKeysPerPass, P, cnts, lane);
            if (a.layout != 0 && full) {
                // linear: the 8 keys of this pass are 8 consecutive bytes of every 8-value group: gather them
                // with 4x4 byte transposes (key-major words -> group-major key bytes) and store 8 bytes per group
                const uint32_t nk = (P - pass * 8) < 8 ? (P - pass * 8) : 8;
                const uint64_t g0 = tile * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4);
#pragma unroll
                for (int j = 0; j < WORDS; j++) {
                    const uint32_t r0[4] = {res[0][j], res[1][j], res[2][j], res[3][j]};
                    const uint32_t r1[4] = {res[4][j], res[5][j], res[6][j], res[7][j]};
                    uint32_t c0[4], c1[4]; // c0[b] = bytes of keys 0..3 for group 4j+b; c1[b] = keys 4..7
                    transpose4x4_bytes(r0, c0);
                    transpose4x4_bytes(r1, c1);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        uint8_t *dst = a.out + (g0 + 4 * j + b) * P + pass * 8;
                        if (nk == 8) {
                            store8_unaligned(dst, c0[b], c1[b]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; q++)
                                if ((uint32_t)q < nk) dst[q] = (uint8_t)((q < 4 ? c0[b] : c1[b]) >> (8 * (q & 3)));
                        }
                    }
                }
            }
        }
        tile = next;
    }
    if (a.hits) block_hits_flush(a, s_hits, P);
    hits_finalize(a, P, lane);
}

// ---- shared scan, exactly two keys: compares, not lookups -------------------------------------------
// The one-pass LUT kernel below does the work of eight keys whatever P is (a byte lookup per value and the full 8 x 8
// bit transposes): P = 2, 3, 4 all take the time of P = 4 (2.5e8 x 9 bit: 0.079 / 0.076 / 0.077 ms -- P = 2 moves
// 4.3 TB/s where P = 4 moves 5.3).  Two keys cost the equality scan's decode twice (v_cmp + v_addc per value and key,
// the extraction shared): VALU 0.04 ms per 2.5e8 values, below the column stream.  The scan's geometry (128 values per lane
// for c <= 16: one 16-byte store per key and lane), the next tile's DMA in flight, a counted vmcnt past the result stores.
// Linear layout: a row is (key 0 byte, key 1 byte), the lane's rows 2 x WORDS x 4 contiguous bytes (v_perm_b32 interleave).
template <int C, int AUX_, int VPL>
__global__ __launch_bounds__(kBlockThreads) void shared_pair_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // result stores: 1 non-temporal, 2 write-through (sc1)
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t k0 = a.key[0], k1 = a.key[1]; // (a key outside [0, 2^C) equals no value: nothing to check)
    const bool linear = a.layout != 0;
    uint32_t hits0 = 0, hits1 = 0;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    int behind = 0; // result stores issued after the DMA the next wait is for
    while (tile < tc.ntiles) {
        if (behind == 2)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (behind == 1)
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        uint32_t r0[1][WORDS], r1[1][WORDS];
#pragma unroll
        for (int j = 0; j < WORDS; j++) r0[0][j] = r1[0][j] = 0;
        decode_step1<C, VPL, 31, kModeEq, G::LANE_DWORDS>(w, r0, k0, 0u);
        decode_step1<C, VPL, 31, kModeEq, G::LANE_DWORDS>(w, r1, k1, 0u);
        if (tile < tc.nfull) {
#pragma unroll
            for (int j = 0; j < WORDS; j++) {
                hits0 += __builtin_popcount(r0[0][j]);
                hits1 += __builtin_popcount(r1[0][j]);
            }
            if (!linear) {
                uint8_t *dst = a.out + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                store_words<WORDS, NTS>(dst, r0[0]);
                store_words<WORDS, NTS>(dst + a.out_stride, r1[0]);
                behind = 2;
            } else {
                uint32_t y[2 * WORDS]; // rows 4j .. 4j+3 of the lane: (key 0, key 1) byte pairs
#pragma unroll
                for (int j = 0; j < WORDS; j++) {
                    y[2 * j] = __builtin_amdgcn_perm(r1[0][j], r0[0][j], 0x05010400u);
                    y[2 * j + 1] = __builtin_amdgcn_perm(r1[0][j], r0[0][j], 0x07030602u);
                }
                uint8_t *dst = a.out + (tile * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4)) * 2;
#pragma unroll
                for (int i = 0; i < 2 * WORDS; i += 4) {
                    const uint32_t v[4] = {y[i], y[i + 1], y[i + 2], y[i + 3]};
                    store_words<4, NTS>(dst + 4 * i, v);
                }
                behind = WORDS / 2;
            }
        } else {
            uint8_t *dst = linear ? a.out + (tile * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4)) * 2
                                  : a.out + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
            hits0 += tc.finish_tail(tile, r0[0], dst, linear ? 2 : 1, lane);
            hits1 += tc.finish_tail(tile, r1[0], linear ? dst + 1 : dst + a.out_stride, linear ? 2 : 1, lane);
            behind = 0;
        }
        tile = next;
    }
    if (a.hits) {
        hits_add(a, 0, wave_sum(hits0), lane);
        hits_add(a, 1, wave_sum(hits1), lane);
    }
    hits_finalize(a, 2, lane);
}

// ---- shared scan through an LDS lookup table ------------------------------------------------------
// v_cmp / v_addc_co run at HALF rate on gfx950 (4.1 cycles per wave-instruction per SIMD against 2.3 for a plain
// VOP2 op; tools/ubench_valu.hip), so the compare chain above costs 8 keys x 2 x 4.1 = 66 cycles per value and
// is VALU-bound at ~1.5e12 values/s whatever the occupancy.  For P predicates the LUT form does ONE LDS byte
// lookup per value and pass of 8 keys instead: entry[v] has bit q set iff v == key[q].  Eight consecutive
// values give eight bytes = an 8x8 bit matrix (value x key); an in-register 8x8 bit transpose (3 masked
// shift/xor rounds on a dword pair) turns it into the eight bitmap bytes (key x value).  That is ~1.1
// cycles per (value, key) result instead of 8.2.
//   one table of 2^C entries per pass while that is affordable: C <= 16 for a single pass (P <= 8: 64 KiB at most, one
//             ds_read_u8 per value), C <= 10 for the multi-pass kernel (the tables of all passes share the LDS);
//   else      ND digit tables (single pass: 2 digits for C <= 24, 3 beyond; multi-pass: byte digits);
//             entry_d[digit_d(v)] has bit q set iff digit_d(key[q]) == digit_d(v); the AND over the digits is exact
//             equality.
// The block zeroes its tables and scatters the keys into them with LDS atomic ORs (O(table/4 + P) per block).
// Keys outside [0, 2^C) get no bit anywhere (they match nothing, as in the reference).  The multi-pass form takes as
// many passes as its tables fit in LDS beside the tiles (checked by the launcher).

struct __attribute__((packed, aligned(1))) Unaligned128 { uint32_t w[4]; };

extern __shared__ __attribute__((aligned(16))) uint8_t mi355_dyn_lds[]; // lookup tables of the multi-pass LUT kernel (size set at launch)

template <int C, bool MULTI> struct LutGeom {
    // digits per value: single pass 1 / 2 / 3 for C <= 16 / 24 / 32 (tables of <= 64 KiB, 2 x 4 KiB, 3 x 2 KiB);
    // multi-pass 1 for C <= 10, else byte digits (small tables, so that many passes fit in LDS)
    static constexpr int ND = MULTI ? (C <= 10 ? 1 : (C + 7) / 8) : (C <= 16 ? 1 : (C <= 24 ? 2 : 3));
    static constexpr bool SINGLE = ND == 1;
    static constexpr int DIGIT_BITS = SINGLE ? C : (MULTI ? 8 : (C + ND - 1) / ND);
    static constexpr int ENTRIES = 1 << DIGIT_BITS;
    static constexpr int TABLE_BYTES = ND * ENTRIES; // per pass of 8 keys
    // digit d of a value or key below 2^C
    static __device__ __forceinline__ uint32_t digit(uint32_t x, int d)
    {
        return (d == ND - 1) ? (x >> (DIGIT_BITS * d)) : ((x >> (DIGIT_BITS * d)) & (uint32_t)(ENTRIES - 1));
    }
};

// 8x8 bit transpose of the 64-bit matrix (hi:lo): bit (8r + c) <-> bit (8c + r)
__device__ __forceinline__ void transpose8x8(uint32_t &lo, uint32_t &hi)
{
    uint32_t t;
    t = (lo ^ (lo >> 7)) & 0x00AA00AAu;  lo ^= t ^ (t << 7);
    t = (hi ^ (hi >> 7)) & 0x00AA00AAu;  hi ^= t ^ (t << 7);
    t = (lo ^ (lo >> 14)) & 0x0000CCCCu; lo ^= t ^ (t << 14);
    t = (hi ^ (hi >> 14)) & 0x0000CCCCu; hi ^= t ^ (t << 14);
    // 64-bit round: t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0 ; x ^= t ^ (t << 28)
    t = (lo ^ ((lo >> 28) | (hi << 4))) & 0xF0F0F0F0u;
    lo ^= t;
    hi ^= t >> 4;
}

template <int C, bool MULTI> __device__ __forceinline__ uint32_t lut_lookup(const uint8_t *table, uint32_t x)
{
    using L = LutGeom<C, MULTI>;
    if constexpr (L::SINGLE) {
        return table[x];
    } else {
        uint32_t m = table[L::digit(x, 0)];
#pragma unroll
        for (int d = 1; d < L::ND; d++) m &= table[d * L::ENTRIES + L::digit(x, d)];
        return m;
    }
}

// Y[g] = (lo, hi): byte q of the pair = bitmap byte of key q for the lane's 8-value group g;
// x[] = the lane's values, extracted once per tile (the passes of a multi-pass scan only differ in the table)
template <int C, int VPL, bool TAIL, bool MULTI>
__device__ __forceinline__ void lut_groups_x(const uint32_t (&x)[VPL], const uint8_t *table, int valid, uint32_t (&Y)[VPL / 8][2])
{
#pragma unroll
    for (int g = 0; g < VPL / 8; g++) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t m = lut_lookup<C, MULTI>(table, x[8 * g + i]);
            if constexpr (TAIL) m = (8 * g + i < valid) ? m : 0u;
            if (i < 4)
                lo |= m << (8 * i);
            else
                hi |= m << (8 * (i - 4));
        }
        transpose8x8(lo, hi);
        Y[g][0] = lo;
        Y[g][1] = hi;
    }
}

template <int C, int VPL, int K, int NW> __device__ __forceinline__ void extract_all(const uint32_t (&w)[NW], uint32_t (&x)[VPL])
{
    x[K] = extract<C, K, NW>(w);
    if constexpr (K + 1 < VPL) extract_all<C, VPL, K + 1, NW>(w, x);
}

// ---- (value x key) -> (key x value) for a whole 32-value bitmap word ------------------------------------
// R[i] byte L holds the 8-key match byte of value 8L + i of the word.  Each byte lane of the 8 registers is then an
// 8x8 bit matrix (register x bit), and the classic three rounds of masked swaps BETWEEN registers (4 / 2 / 1 apart,
// masks 0x0F.. / 0x33.. / 0x55..) transpose all four lanes at once: afterwards R[q] bit 8L + i = match of value
// 8L + i against key q, i.e. R[q] IS the bitmap word of key q.  72 VOP2 ops per 32 values x 8 keys, against 4 x 30
// for four 8x8 transposes inside register pairs plus 16 v_perm_b32 to gather the keys' bytes.
__device__ __forceinline__ void transpose_bits_8regs(uint32_t (&R)[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t t = ((R[i] >> 4) ^ R[i + 4]) & 0x0F0F0F0Fu;
        R[i + 4] ^= t;
        R[i] ^= t << 4;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = (k & 1) + 4 * (k >> 1); // 0, 1, 4, 5
        const uint32_t t = ((R[i] >> 2) ^ R[i + 2]) & 0x33333333u;
        R[i + 2] ^= t;
        R[i] ^= t << 2;
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const uint32_t t = ((R[i] >> 1) ^ R[i + 1]) & 0x55555555u;
        R[i + 1] ^= t;
        R[i] ^= t << 1;
    }
}

// out[q][j] = bitmap word j of the lane for key q (one table lookup per value, byte entries); TAIL: values >= valid
// contribute nothing
template <int C, int VPL, bool TAIL, bool MULTI>
__device__ __forceinline__ void lut_words(const uint32_t (&x)[VPL], const uint8_t *table, int valid, uint32_t (&out)[8][VPL / 32])
{
#pragma unroll
    for (int j = 0; j < VPL / 32; j++) {
        uint32_t R[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t r = 0;
#pragma unroll
            for (int L = 0; L < 4; L++) {
                uint32_t m = lut_lookup<C, MULTI>(table, x[32 * j + 8 * L + i]);
                if constexpr (TAIL) m = (32 * j + 8 * L + i < valid) ? m : 0u;
                r |= m << (8 * L);
            }
            R[i] = r;
        }
        transpose_bits_8regs(R);
#pragma unroll
        for (int q = 0; q < 8; q++) out[q][j] = R[q];
    }
}

// linear layout: Y[g] = (lo, hi) with byte q = bitmap byte of key q for the lane's 8-value group g = 4j + L, i.e.
// bytes L of out[0..3][j] and of out[4..7][j]
template <int VPL> __device__ __forceinline__ void words_to_groups(const uint32_t (&out)[8][VPL / 32], uint32_t (&Y)[VPL / 8][2])
{
#pragma unroll
    for (int j = 0; j < VPL / 32; j++) {
        const uint32_t r0[4] = {out[0][j], out[1][j], out[2][j], out[3][j]};
        const uint32_t r1[4] = {out[4][j], out[5][j], out[6][j], out[7][j]};
        uint32_t lo[4], hi[4];
        transpose4x4_bytes(r0, lo);
        transpose4x4_bytes(r1, hi);
#pragma unroll
        for (int L = 0; L < 4; L++) {
            Y[4 * j + L][0] = lo[L];
            Y[4 * j + L][1] = hi[L];
        }
    }
}

// per-key bitmap words of the lane: out[q][j] = bytes q of Y[4j..4j+3]
template <int VPL> __device__ __forceinline__ void lut_gather_keys(const uint32_t (&Y)[VPL / 8][2], uint32_t (&out)[8][VPL / 32])
{
#pragma unroll
    for (int j = 0; j < VPL / 32; j++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t r[4] = {Y[4 * j + 0][h], Y[4 * j + 1][h], Y[4 * j + 2][h], Y[4 * j + 3][h]};
            uint32_t c[4];
            transpose4x4_bytes(r, c);
#pragma unroll
            for (int q = 0; q < 4; q++) out[4 * h + q][j] = c[q];
        }
    }
}

struct __attribute__((packed, aligned(1))) Unaligned16 {
    uint32_t a, b, c, d;
};
// Linear layout, P = 3, 5, 6, 7 keys (one pass): the lane's GROUPS rows of P bytes are GROUPS*P CONTIGUOUS output bytes
// (8-byte aligned), the wave's tile 64 x that.  Y[g] = (lo, hi) holds row g with key q in byte q: the rows are packed
// back to back in registers -- every shift and byte select is a compile-time constant once P is one -- and leave as
// 16 / 8-byte stores.  (Byte stores straight from Y, the first version, ran at 0.6-2.0 TB/s: 8 P store instructions per lane
// and tile; 2.5e8 x 9 bit, P = 7: 0.875 ms.)
template <int P, int GROUPS, int NRES> __device__ __forceinline__ void store_linear_rows_packed(uint8_t *dst, const uint32_t (&res)[NRES])
{
    constexpr int NDW = GROUPS * P / 4; // GROUPS is a multiple of 8
    uint32_t d[NDW];
#pragma unroll
    for (int i = 0; i < NDW; i++) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int b = 4 * i + k, g = b / P, q = b % P;
            v |= ((res[2 * g + (q >> 2)] >> (8 * (q & 3))) & 0xffu) << (8 * k);
        }
        d[i] = v;
    }
#pragma unroll
    for (int i = 0; i + 4 <= NDW; i += 4) *(Unaligned16 *)(dst + 4 * i) = Unaligned16{d[i], d[i + 1], d[i + 2], d[i + 3]};
    if constexpr (NDW % 4 >= 2) store8_unaligned(dst + 4 * (NDW & ~3), d[NDW & ~3], d[(NDW & ~3) + 1]);
    static_assert(NDW % 2 == 0, "GROUPS * P is a multiple of 8 bytes");
}

// LAYOUT 0: per-predicate bitmaps at out + k*out_stride; 1: linear (byte of 8-value group g and key k at
// g*P + k, src/simd_scan_shared_linear.cpp:57).  MULTI false: P <= 8, one pass, stores deferred by one tile
// (as in the equality scan); true: ceil(P/8) passes per tile, stored pass by pass.
template <int C, int AUX_, int VPL, int LAYOUT, bool MULTI>
__global__ __launch_bounds__(kBlockThreads) void shared_lut_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    using L = LutGeom<C, MULTI>;
    constexpr int WORDS = G::WORDS;
    constexpr int GROUPS = VPL / 8;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // result stores: 1 non-temporal, 2 write-through (sc1)
    constexpr int NRES = LAYOUT == 0 ? 8 * WORDS : GROUPS * 2; // result dwords per lane, tile and pass
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t lut_static[(MULTI || L::TABLE_BYTES < 16) ? 16 : L::TABLE_BYTES];
    uint8_t *const lut = MULTI ? mi355_dyn_lds : lut_static; // MULTI: npass * TABLE_BYTES dynamic bytes
    __shared__ uint32_t s_hits[MULTI ? kMaxKeys : 1];          // MULTI: per-block hit counters (block_hits_add8)
    __shared__ __attribute__((aligned(16))) uint8_t stage[(LAYOUT == 1 && !MULTI) ? kWavesPerBlock : 1][(LAYOUT == 1 && !MULTI) ? GROUPS * 8 * 64 : 16];
    if constexpr (MULTI)
        for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxKeys; k += kBlockThreads) s_hits[k] = 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t npass = MULTI ? (P + 7) / 8 : 1;

    // the tile's DMA does not depend on the tables: get it going first
    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);

    // tables: zero, then OR bit (k % 8) into the entry (or, digit tables: the ND entries) of every in-range key k
    {
        uint32_t *const lut32 = (uint32_t *)lut;
        const uint32_t ndw = (npass * L::TABLE_BYTES + 3) / 4;
        for (uint32_t i = threadIdx.x; i < ndw; i += kBlockThreads) lut32[i] = 0;
        __syncthreads();
        auto scatter = [&](uint32_t k, uint32_t key) {
            const bool in_range = C == 32 || (key >> (C & 31)) == 0;
            if (!in_range) return;
            const uint32_t base = (k >> 3) * L::TABLE_BYTES;
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                const uint32_t idx = base + d * L::ENTRIES + e;
                __hip_atomic_fetch_or(lut32 + (idx >> 2), (1u << (k & 7)) << (8 * (idx & 3)), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        if constexpr (MULTI) {
            for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) scatter(k, (uint32_t)a.keys_dev[k]);
        } else {
            if (threadIdx.x == 0) {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if ((uint32_t)q < P) scatter(q, a.key[q]);
            }
        }
    }
    __syncthreads();

    uint32_t hits[8];
#pragma unroll
    for (int q = 0; q < 8; q++) hits[q] = 0;

    // full-tile store of one pass: LAYOUT 0: res = out[q][j] (q-major); LAYOUT 1: res = Y[g][0..1]
    auto store_full = [&](uint64_t t, uint32_t pass, const uint32_t (&res)[NRES]) {
        if constexpr (LAYOUT == 0) {
            uint8_t *dst = a.out + (uint64_t)(pass * 8) * a.out_stride + t * G::BITMAP_BYTES + lane * (WORDS * 4);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (pass * 8 + q < P) {
                    uint32_t v[WORDS];
#pragma unroll
                    for (int j = 0; j < WORDS; j++) v[j] = res[q * WORDS + j];
                    store_words<WORDS, NTS>(dst, v);
                }
                dst += a.out_stride;
            }
        } else {
            const uint64_t g0 = t * G::BITMAP_BYTES + (uint64_t)lane * GROUPS;
            if (!MULTI && P == 4) { // (8 keys: see stage_put / stage_store below)
                // 4 keys: a group's row is the dword of keys 0..3; the lane's GROUPS rows are contiguous
                u32x4 *dst = (u32x4 *)(a.out + g0 * 4);
#pragma unroll
                for (int g = 0; g < GROUPS; g += 4) {
                    u32x4 v = {res[2 * g], res[2 * g + 2], res[2 * g + 4], res[2 * g + 6]};
                    dst[g / 4] = v;
                }
            } else if (!MULTI && P == 2) {
                // 2 keys: a row is 2 bytes; rows of two groups share a dword (v_perm_b32: bytes 0,1 of each source)
                u32x4 *dst = (u32x4 *)(a.out + g0 * 2);
#pragma unroll
                for (int g = 0; g < GROUPS; g += 8) {
                    u32x4 v = {__builtin_amdgcn_perm(res[2 * g + 2], res[2 * g], 0x05040100u),
                               __builtin_amdgcn_perm(res[2 * g + 6], res[2 * g + 4], 0x05040100u),
                               __builtin_amdgcn_perm(res[2 * g + 10], res[2 * g + 8], 0x05040100u),
                               __builtin_amdgcn_perm(res[2 * g + 14], res[2 * g + 12], 0x05040100u)};
                    dst[g / 8] = v;
                }
            } else if (!MULTI && P == 3) {
                store_linear_rows_packed<3, GROUPS, NRES>(a.out + g0 * 3, res);
            } else if (!MULTI && P == 5) {
                store_linear_rows_packed<5, GROUPS, NRES>(a.out + g0 * 5, res);
            } else if (!MULTI && P == 6) {
                store_linear_rows_packed<6, GROUPS, NRES>(a.out + g0 * 6, res);
            } else if (!MULTI && P == 7) {
                store_linear_rows_packed<7, GROUPS, NRES>(a.out + g0 * 7, res);
            } else {
                const uint32_t nk = (P - pass * 8) < 8 ? (P - pass * 8) : 8;
#pragma unroll
                for (int g = 0; g < GROUPS; g++) {
                    uint8_t *dst = a.out + (g0 + g) * P + pass * 8;
                    if (nk == 8) {
                        store8_unaligned(dst, res[2 * g], res[2 * g + 1]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; q++)
                            if ((uint32_t)q < nk) dst[q] = (uint8_t)(res[2 * g + (q >> 2)] >> (8 * (q & 3)));
                    }
                }
            }
        }
    };

    // Linear layout, 8 keys: the lane's GROUPS x 8 result bytes are 8*GROUPS contiguous output bytes and the wave's tile
    // 64 x that.  They are transposed through a per-wave LDS stage so that every store instruction writes 1 KiB
    // contiguous (instead of 64 x 16 B at a 64-B stride): written to the stage as soon as they are computed, read back
    // at the top of the next iteration together with the next tile's data (one LDS wait for both), stored, and only
    // then is the following DMA issued -- the stage round trip never sits between a DMA wait and the next DMA issue.
    const bool lin8 = LAYOUT == 1 && !MULTI && P == 8;
    auto stage_put = [&](const uint32_t (&res)[NRES]) {
        u32x4 *st = (u32x4 *)stage[wave];
#pragma unroll
        for (int g = 0; g < GROUPS; g += 2) {
            u32x4 v = {res[2 * g], res[2 * g + 1], res[2 * g + 2], res[2 * g + 3]};
            st[lane * (GROUPS / 2) + g / 2] = v;
        }
    };
    auto stage_get = [&](u32x4 (&r)[GROUPS / 2]) {
        const u32x4 *st = (const u32x4 *)stage[wave];
#pragma unroll
        for (int j = 0; j < GROUPS / 2; j++) r[j] = st[j * 64 + lane];
    };
    auto stage_store = [&](uint64_t t, const u32x4 (&r)[GROUPS / 2]) {
        u32x4 *dst = (u32x4 *)(a.out + (t * G::BITMAP_BYTES) * 8);
#pragma unroll
        for (int j = 0; j < GROUPS / 2; j++) {
            if constexpr (NTS == 2)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + j * 64 + lane), "v"(r[j]) : "memory");
            else if constexpr (NTS == 1)
                __builtin_nontemporal_store(r[j], dst + j * 64 + lane);
            else
                dst[j * 64 + lane] = r[j];
        }
    };

    uint32_t resp[NRES]; // !MULTI: results of the previous tile, not yet stored (lin8: they wait in the LDS stage)
    uint64_t prev = ~0ull;

    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        u32x4 staged[GROUPS / 2];
        if (lin8 && prev != ~0ull) stage_get(staged);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!MULTI) {
            if (prev != ~0ull) {
                if (lin8)
                    stage_store(prev, staged);
                else
                    store_full(prev, 0, resp);
            }
            prev = ~0ull;
        }
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        const bool full = tile < tc.nfull;
        uint32_t xs[VPL];
        extract_all<C, VPL, 0, G::LANE_DWORDS>(w, xs);

        if constexpr (MULTI && LAYOUT == 1) {
            // Linear layout, many keys: row g (8-value group) holds P bytes, and a lane's 8 rows are contiguous.
            // Walk the rows in order and, inside a row, the passes in order, so every row is written start to end
            // in one go (the pass-major order revisits each 128-B line npass times: 10x slower at P = 512).
            if (full) {
                const uint64_t g0 = tile * G::BITMAP_BYTES + (uint64_t)lane * GROUPS;
#pragma unroll
                for (int g = 0; g < GROUPS; g++) {
                    uint8_t *row = a.out + (g0 + g) * P;
                    uint32_t xg[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) xg[i] = xs[8 * g + i];
                    auto pass8 = [&](uint32_t pass, uint32_t &lo, uint32_t &hi) {
                        const uint8_t *table = lut + pass * L::TABLE_BYTES;
                        lo = 0;
                        hi = 0;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const uint32_t m = lut_lookup<C, MULTI>(table, xg[i]);
                            if (i < 4)
                                lo |= m << (8 * i);
                            else
                                hi |= m << (8 * (i - 4));
                        }
                        transpose8x8(lo, hi);
                    };
                    uint32_t pass = 0;
                    // two passes = 16 keys = one 16-byte store while whole pairs remain
                    for (; pass + 2 <= P / 8; pass += 2) {
                        uint32_t l0, h0, l1, h1;
                        pass8(pass, l0, h0);
                        pass8(pass + 1, l1, h1);
                        Unaligned128 v;
                        v.w[0] = l0; v.w[1] = h0; v.w[2] = l1; v.w[3] = h1;
                        *(Unaligned128 *)(row + pass * 8) = v;
                    }
                    for (; pass < npass; pass++) {
                        uint32_t lo, hi;
                        pass8(pass, lo, hi);
                        const uint32_t nk = (P - pass * 8) < 8 ? (P - pass * 8) : 8;
                        if (nk == 8) {
                            store8_unaligned(row + pass * 8, lo, hi);
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; q++)
                                if ((uint32_t)q < nk) row[pass * 8 + q] = (uint8_t)((q < 4 ? lo : hi) >> (8 * (q & 3)));
                        }
                    }
                }
            }
        }

        // pass-major loop: per-predicate stores, hit counts, tail tiles (and everything for one-pass scans)
        for (uint32_t pass = 0; pass < npass && (!(MULTI && LAYOUT == 1) || !full || a.hits); pass++) {
            const uint8_t *table = lut + pass * L::TABLE_BYTES;
            uint32_t Y[GROUPS][2];
            uint32_t out[8][WORDS];
            if (full) {
                // register-wise transposition of whole bitmap words (lut_words).  Same-box A/B against the earlier
                // per-group 8x8 transposes + byte gather, 1e9 x 9 bit, launches back to back: linear layout P = 8
                // 0.374 against 0.394 ms (the hit counts no longer need their own gather), per-predicate P = 8 equal
                // (0.379: the stream, not the VALU, bounds it), P = 2 0.300 against 0.338 ms.
                lut_words<C, VPL, false, MULTI>(xs, table, VPL, out);
                if constexpr (LAYOUT == 1) words_to_groups<VPL>(out, Y);
                if (a.hits) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
#pragma unroll
                        for (int j = 0; j < WORDS; j++) hits[q] += __builtin_popcount(out[q][j]);
                }
                uint32_t res[NRES];
                if constexpr (LAYOUT == 0) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
#pragma unroll
                        for (int j = 0; j < WORDS; j++) res[q * WORDS + j] = out[q][j];
                } else {
#pragma unroll
                    for (int g = 0; g < GROUPS; g++) { res[2 * g] = Y[g][0]; res[2 * g + 1] = Y[g][1]; }
                }
                if constexpr (!MULTI) {
                    if (lin8) {
                        stage_put(res);
                    } else {
#pragma unroll
                        for (int i = 0; i < NRES; i++) resp[i] = res[i];
                    }
                    prev = tile;
                } else {
                    if (a.hits) {
                        block_hits_add8(s_hits, pass * 8, P, hits, lane);
#pragma unroll
                        for (int q = 0; q < 8; q++) hits[q] = 0;
                    }
                    // LAYOUT 1: the rows are written group by group below (each row's P bytes back to back)
                    if constexpr (LAYOUT == 0) store_full(tile, pass, res);
                }
            } else {
                // tail tile: lookups of values >= n are zeroed; the bitmap is written byte-exact
                const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
                const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
                const int nbytes = (valid + 7) / 8;
                if constexpr (LAYOUT == 0) { // (the per-group form here: with lut_words in this cold branch the compiler's
                                             // schedule of the hot full-tile path came out 8 % slower)
                    lut_groups_x<C, VPL, true, MULTI>(xs, table, valid, Y);
                    lut_gather_keys<VPL>(Y, out);
                } else {
                    lut_words<C, VPL, true, MULTI>(xs, table, valid, out);
                }
                uint32_t tcnt[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t k = pass * 8 + q;
                    tcnt[q] = 0;
                    if (k < P) {
                        uint32_t cnt = 0;
#pragma unroll
                        for (int j = 0; j < WORDS; j++) cnt += __builtin_popcount(out[q][j]);
                        if constexpr (!MULTI)
                            hits[q] += cnt;
                        else
                            tcnt[q] = cnt;
                        uint8_t *dst = LAYOUT == 0
                                           ? a.out + (uint64_t)k * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4)
                                           : a.out + (tile * G::BITMAP_BYTES + (uint64_t)lane * GROUPS) * P + k;
                        const uint64_t bstride = LAYOUT == 0 ? 1 : P;
#pragma unroll
                        for (int b = 0; b < WORDS * 4; b++)
                            if (b < nbytes) dst[(uint64_t)b * bstride] = (uint8_t)(out[q][b >> 2] >> (8 * (b & 3)));
                    }
                }
                if constexpr (MULTI) {
                    if (a.hits) block_hits_add8(s_hits, pass * 8, P, tcnt, lane);
                }
            }
        }
        tile = next;
    }
    if constexpr (MULTI) {
        if (a.hits) block_hits_flush(a, s_hits, P);
    }
    if constexpr (!MULTI) {
        if (prev != ~0ull) {
            if (lin8) {
                u32x4 staged[GROUPS / 2];
                stage_get(staged);
                stage_store(prev, staged);
            } else {
                store_full(prev, 0, resp);
            }
        }
        if (a.hits) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t s = wave_sum(hits[q]);
                if ((uint32_t)q < P) hits_add(a, q, s, lane);
            }
        }
    }
    hits_finalize(a, P, lane);
}

// ---- shared scan, P > 8: 32 predicates per LDS lookup -----------------------------------------------
// With byte entries a scan over P keys does P/8 lookups per value, and at P >= 32 those lookups (random
// ds_read_u8, ~5 LDS cycles each per CU) bound the kernel at ~3.7e12 lookups/s.  Here the tables hold DWORD
// entries -- bit (k % 32) of entry[v] of table k / 32 is set iff v == key[k] -- so one ds_read_b32 answers 32
// predicates: the 8 dwords of an 8-value group are split into their 4 key-bytes with two 4x4 byte transposes
// (v_perm_b32), and each key-byte's (value x key) 8x8 bit matrix is transposed as in shared_lut_kernel.
//   C <= 10: one table of 2^C dwords per 32 keys; C > 10: ceil(C/8) byte-digit tables of 256 dwords, ANDed.
// LAYOUT 0: per-predicate bitmaps, pass-major (a wave-store is 512 B contiguous per key).
// LAYOUT 1: linear; every 8-value group's row of P bytes is written start to end, 32 bytes per lookup round.
// BIG: fewer, wider digits where byte digits need three or four -- c = 17 .. 20: two digits of 9 / 10 bits (4 / 8 KiB per 32
// keys instead of 3 KiB), c = 25 .. 30: three of 9 / 10 bits (6 / 12 KiB instead of 4 KiB): one lookup, one AND and a digit
// extraction less per value and round, for key counts whose tables still fit (the launcher decides).
template <int C, bool BIG = false> struct WideLutGeom {
    static constexpr int ND = C <= 10 ? 1 : (BIG && C >= 17 && C <= 20) ? 2 : (BIG && C >= 25 && C <= 30) ? 3 : (C + 7) / 8;
    static constexpr bool SINGLE = ND == 1;
    static constexpr int DIGIT_BITS = SINGLE ? C : (BIG ? (C + ND - 1) / ND : 8);
    static constexpr int ENTRIES = 1 << DIGIT_BITS;
    static constexpr int TABLE_DWORDS = ND * ENTRIES; // per pass of 32 keys
    static constexpr int TABLE_BYTES = TABLE_DWORDS * 4;
    static __device__ __forceinline__ uint32_t digit(uint32_t x, int d)
    {
        return (d == ND - 1) ? (x >> (DIGIT_BITS * d)) : ((x >> (DIGIT_BITS * d)) & (uint32_t)(ENTRIES - 1));
    }
    static __device__ __forceinline__ uint32_t lookup(const uint32_t *table, uint32_t x)
    {
        if constexpr (SINGLE) {
            return table[x];
        } else {
            uint32_t m = table[digit(x, 0)];
#pragma unroll
            for (int d = 1; d < ND; d++) m &= table[d * ENTRIES + digit(x, d)];
            return m;
        }
    }
};

// The next tile's DMA is issued BEFORE the tile's result stores, and vmcnt retires in issue order: once at most
// `nstores` operations are outstanding, where `nstores` stores were issued after that DMA, the DMA has landed.  So the
// top of the loop waits for a COUNT instead of draining the stores of the previous tile (a drain per tile left the
// store queue empty during every lookup / transpose phase: stores in bursts of P between compute phases).
__device__ __forceinline__ void wait_dma_behind_stores(uint32_t nstores)
{
    if (nstores >= 48)
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (nstores >= 32)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (nstores >= 16)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (nstores >= 8)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C, int AUX_, int VPL, int LAYOUT>
__global__ __launch_bounds__(kBlockThreads) void shared_wide_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C>;
    constexpr int WORDS = G::WORDS;
    constexpr int GROUPS = VPL / 8;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // result stores: 1 non-temporal, 2 write-through (sc1)
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ uint32_t s_hits[kMaxKeys]; // per-block hit counters (block_hits_add8)
    uint32_t *const lut = (uint32_t *)mi355_dyn_lds; // ceil(P/32) * TABLE_BYTES dynamic bytes
    for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxKeys; k += kBlockThreads) s_hits[k] = 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t npass32 = (P + 31) / 32;

    // the tile's DMA does not depend on the tables: get it going first
    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);

    for (uint32_t i = threadIdx.x; i < npass32 * L::TABLE_DWORDS; i += kBlockThreads) lut[i] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
        const uint32_t key = (uint32_t)a.keys_dev[k];
        const bool in_range = C == 32 || (key >> (C & 31)) == 0;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                __hip_atomic_fetch_or(lut + (k >> 5) * L::TABLE_DWORDS + d * L::ENTRIES + e, 1u << (k & 31), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // (value x 32 keys) dwords of one 8-value group -> for each key-byte b the transposed pair: byte q of
    // (lo, hi) = bitmap byte of key 8b + q for this group
    // (nb = key-bytes in use, 1..4: the last table of a scan over P keys may be partly empty)
    auto group32 = [&](const uint32_t *table, const uint32_t (&x)[VPL], int g, int valid, bool tail, uint32_t nb, uint32_t (&Y)[4][2]) {
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            m[i] = L::lookup(table, x[8 * g + i]);
            if (tail) m[i] = (8 * g + i < valid) ? m[i] : 0u;
        }
        const uint32_t r0[4] = {m[0], m[1], m[2], m[3]}, r1[4] = {m[4], m[5], m[6], m[7]};
        uint32_t lo4[4], hi4[4];
        transpose4x4_bytes(r0, lo4);
        transpose4x4_bytes(r1, hi4);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t lo = lo4[b], hi = hi4[b];
            if ((uint32_t)b < nb) transpose8x8(lo, hi);
            Y[b][0] = lo;
            Y[b][1] = hi;
        }
    };

    // store instructions a FULL tile issues after the next tile's DMA (a lower bound is what the wait needs):
    // per-predicate: one per key; linear: at least one per 8-value group and 32-key round
    const uint32_t stores_per_tile = LAYOUT == 0 ? P : (uint32_t)GROUPS * npass32;
    uint32_t behind = 0; // stores issued after the DMA the next wait is for
    while (tile < tc.ntiles) {
        wait_dma_behind_stores(behind);
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        const bool full = tile < tc.nfull;
        behind = (full && !(a.flags & 1u)) ? stores_per_tile : 0; // (a tail tile is a wave's last: nothing waits behind it)
        uint32_t xs[VPL];
        extract_all<C, VPL, 0, G::LANE_DWORDS>(w, xs);
        const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
        const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);

        if constexpr (LAYOUT == 1) {
            if (full) {
                const uint64_t g0 = tile * G::BITMAP_BYTES + (uint64_t)lane * GROUPS;
#pragma unroll
                for (int g = 0; g < GROUPS; g++) {
                    uint8_t *row = a.out + (g0 + g) * P;
                    for (uint32_t p32 = 0; p32 < npass32; p32++) {
                        uint32_t Y[4][2];
                        const uint32_t nk = (P - p32 * 32) < 32 ? (P - p32 * 32) : 32;
                        group32(lut + p32 * L::TABLE_DWORDS, xs, g, VPL, false, (nk + 7) / 8, Y);
                        uint8_t *dst = row + p32 * 32;
                        if (nk == 32) {
                            Unaligned128 v0, v1;
                            v0.w[0] = Y[0][0]; v0.w[1] = Y[0][1]; v0.w[2] = Y[1][0]; v0.w[3] = Y[1][1];
                            v1.w[0] = Y[2][0]; v1.w[1] = Y[2][1]; v1.w[2] = Y[3][0]; v1.w[3] = Y[3][1];
                            *(Unaligned128 *)dst = v0;
                            *(Unaligned128 *)(dst + 16) = v1;
                        } else {
#pragma unroll
                            for (int b = 0; b < 4; b++) {
                                if ((uint32_t)(8 * b + 8) <= nk) {
                                    store8_unaligned(dst + 8 * b, Y[b][0], Y[b][1]);
                                } else {
#pragma unroll
                                    for (int q = 0; q < 8; q++)
                                        if ((uint32_t)(8 * b + q) < nk) dst[8 * b + q] = (uint8_t)(Y[b][q >> 2] >> (8 * (q & 3)));
                                }
                            }
                        }
                    }
                }
            }
        }

        // pass-major: per-predicate stores, hit counts, tail tiles
        if (LAYOUT == 0 || !full || a.hits) {
            for (uint32_t p32 = 0; p32 < npass32; p32++) {
                const uint32_t *table = lut + p32 * L::TABLE_DWORDS;
                uint32_t Yb[4][GROUPS][2];
                const uint32_t nb32 = ((P - p32 * 32) < 32 ? (P - p32 * 32) + 7 : 39) / 8;
#pragma unroll
                for (int g = 0; g < GROUPS; g++) {
                    uint32_t Y[4][2];
                    group32(table, xs, g, valid, !full, nb32, Y);
#pragma unroll
                    for (int b = 0; b < 4; b++) { Yb[b][g][0] = Y[b][0]; Yb[b][g][1] = Y[b][1]; }
                }
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t pass = p32 * 4 + b;
                    if (pass * 8 < P) {
                        uint32_t out[8][WORDS];
                        lut_gather_keys<VPL>(Yb[b], out);
                        uint32_t cnt[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            cnt[q] = 0;
#pragma unroll
                            for (int j = 0; j < WORDS; j++) cnt[q] += __builtin_popcount(out[q][j]);
                        }
                        if (a.hits) block_hits_add8(s_hits, pass * 8, P, cnt, lane);
                        if (full) {
                            if constexpr (LAYOUT == 0) {
                                uint8_t *dst = a.out + (uint64_t)(pass * 8) * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
#pragma unroll
                                for (int q = 0; q < 8; q++) {
                                    if (pass * 8 + q < P) store_words<WORDS, NTS>(dst, out[q]);
                                    dst += a.out_stride;
                                }
                            }
                        } else {
                            // tail tile: lookups of values >= n were zeroed; the bitmap is written byte-exact
                            const int nbytes = (valid + 7) / 8;
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const uint32_t k = pass * 8 + q;
                                if (k < P) {
                                    uint8_t *dst = LAYOUT == 0
                                                       ? a.out + (uint64_t)k * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4)
                                                       : a.out + (tile * G::BITMAP_BYTES + (uint64_t)lane * GROUPS) * P + k;
                                    const uint64_t bstride = LAYOUT == 0 ? 1 : P;
#pragma unroll
                                    for (int bb = 0; bb < WORDS * 4; bb++)
                                        if (bb < nbytes) dst[(uint64_t)bb * bstride] = (uint8_t)(out[q][bb >> 2] >> (8 * (bb & 3)));
                                }
                            }
                        }
                    }
                }
            }
        }
        tile = next;
    }
    if (a.hits) block_hits_flush(a, s_hits, P);
    hits_finalize(a, P, lane);
}

// ---- shared scan, P > 8, per-predicate bitmaps: whole bitmap words at a time -------------------------------------
// shared_wide_kernel transposes every 8-value group on its own (two 4x4 byte transposes, four 8x8 bit transposes inside
// register pairs, then a byte gather per key): ~0.6 VALU operations per (value, key) result, most of them VOP3 (half
// rate), and ~70 % of the kernel's time on MI355X.  Here the 32 lookups of a 32-value bitmap word are regrouped by eight
// 4x4 byte transposes so that register R[b][i] holds key-byte b of values i, 8+i, 16+i, 24+i, and each key-byte takes
// the register-wise 8x8 bit transpose of the one-pass kernel (transpose_bits_8regs: 72 VOP2 operations for 32 values x
// 8 keys): afterwards R[b][q] IS the bitmap word of key 8b + q -- 0.34 operations per result, no gather.
// REGCNT: hit counts of a single 32-key round (P <= 32) in 32 registers of the lane (see below); its own instantiation
// because the registers would cost the other paths a resident wave (c = 11, 12: 248 -> 256 + spills to AGPRs).
// RC (hit counts in registers): 0 none; 1 one round of 32 keys, a 32-bit register per key (P <= 32); 2 two rounds, two
// keys per register in 16-bit halves, flushed every kPackedFlushTiles tiles (P <= 64) -- for the widths of digit tables, where
// the per-tile wave reductions + LDS atomics of block_hits_add8 cost the 32-key rounds a quarter of their time at two waves
// per SIMD (2.5e8 x 17 bit, P = 64: 0.837 ms with hit counts against 0.657 without).
constexpr uint32_t kPackedFlushTiles = 512; // a lane adds at most 64 per key and tile: 512 x 64 < 2^16

template <int C, int AUX_, int VPL, int RC = 0, bool BIG = false>
__global__ __launch_bounds__(kBlockThreads) void shared_wide2_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C, BIG>;
    constexpr bool REGCNT = RC == 1;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // result stores: 1 non-temporal, 2 write-through (sc1)
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ uint32_t s_hits[kMaxKeys]; // per-block hit counters (block_hits_add8)
    // Hit counts by HISTOGRAM while 2^C counters fit in LDS: hits[k] = number of values equal to keys[k], whatever P is --
    // one LDS atomic add per value instead of popcounts + four wave reductions per eight keys and tile (which cost 10-35 %
    // of the kernel).  The block reads its histogram at the keys once, at the end.  An LDS atomic per value costs ~0.07 ms per
    // 2.5e8 values whatever P is, the popcount way 0.05-0.07 ms per 32-key round: the histogram pays from two rounds on.
    // P <= 32 (one round of 32 keys): neither -- every key's count lives in a register of the lane for the whole launch
    // (v_bcnt_u32_b32 accumulates: two operations per key and tile), reduced over the wave once at the end.
    constexpr bool HIST = C <= 12;
    __shared__ uint32_t hist[HIST ? (1 << C) : 1];
    uint32_t *const lut = (uint32_t *)mi355_dyn_lds; // ceil(P/32) * TABLE_BYTES dynamic bytes
    for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxKeys; k += kBlockThreads) s_hits[k] = 0;
    if constexpr (HIST)
        for (uint32_t k = threadIdx.x; k < (1u << C); k += kBlockThreads) hist[k] = 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t npass32 = (P + 31) / 32;
    constexpr bool reg_counts = REGCNT; // (the launcher picks it for P <= 32 with hit counts)
    const bool use_hist = HIST && P >= 64 && RC == 0;
    uint32_t acc[REGCNT ? 4 : 1][8];
#pragma unroll
    for (int b = 0; b < (REGCNT ? 4 : 1); b++)
#pragma unroll
        for (int q = 0; q < 8; q++) acc[b][q] = 0;
    // RC == 2: acc2[round][key-byte][pair]: keys 32 p + 8 b + 2 q (low half) and + 1 (high half)
    uint32_t acc2[RC == 2 ? 2 : 1][4][4];
#pragma unroll
    for (int p = 0; p < (RC == 2 ? 2 : 1); p++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc2[p][b][q] = 0;
    uint32_t tiles_counted = 0;
    auto flush_packed = [&]() {
        if constexpr (RC == 2) {
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t lo = wave_sum(acc2[p][b][q] & 0xffffu), hi = wave_sum(acc2[p][b][q] >> 16);
                        acc2[p][b][q] = 0;
                        if (lane == 0) {
                            unsigned long long *slot = a.scratch + (blockIdx.x % kHitSlots) * kMaxKeys + 32 * p + 8 * b + 2 * q;
                            if (lo) __hip_atomic_fetch_add(slot, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (hi) __hip_atomic_fetch_add(slot + 1, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
        }
        tiles_counted = 0;
    };

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);

    for (uint32_t i = threadIdx.x; i < npass32 * L::TABLE_DWORDS; i += kBlockThreads) lut[i] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
        const uint32_t key = (uint32_t)a.keys_dev[k];
        const bool in_range = C == 32 || (key >> (C & 31)) == 0;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                __hip_atomic_fetch_or(lut + (k >> 5) * L::TABLE_DWORDS + d * L::ENTRIES + e, 1u << (k & 31), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    uint32_t behind = 0; // stores issued after the DMA the next wait is for (wait_dma_behind_stores)
    while (tile < tc.ntiles) {
        wait_dma_behind_stores(behind);
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        const bool full = tile < tc.nfull;
        behind = (full && !(a.flags & 1u)) ? P : 0;
        uint32_t xs[VPL];
        extract_all<C, VPL, 0, G::LANE_DWORDS>(w, xs);
        const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
        const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
        if constexpr (HIST) {
            if (a.hits && use_hist) {
#pragma unroll
                for (int v = 0; v < VPL; v++)
                    if (full || v < valid) __hip_atomic_fetch_add(&hist[xs[v]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }

        // (flags bit 2, experiment: every wave starts its round-robin over the 32-key rounds at a different round, so the
        // waves of the lock-stepped grid do not all write the same 32 output streams at the same time)
        const uint32_t rot = (RC != 2 && (a.flags & 4u)) ? (uint32_t)((blockIdx.x * kWavesPerBlock + wave) % npass32) : 0u;
        auto do_round = [&](const uint32_t pi) __attribute__((always_inline)) {
            const uint32_t p32 = pi + rot < npass32 ? pi + rot : pi + rot - npass32;
            const uint32_t *table = lut + p32 * L::TABLE_DWORDS;
            const uint32_t nb = ((P - p32 * 32) < 32 ? (P - p32 * 32) + 7 : 39) / 8; // key-bytes in use, 1..4
            uint32_t outw[4][8][WORDS]; // [key-byte][key in byte][bitmap word of the lane]
#pragma unroll
            for (int j = 0; j < WORDS; j++) {
                uint32_t R[4][8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint32_t r[4], c[4];
#pragma unroll
                    for (int Lg = 0; Lg < 4; Lg++) {
                        const int v = 32 * j + 8 * Lg + i;
                        uint32_t m = L::lookup(table, xs[v]);
                        if (!full) m = v < valid ? m : 0u; // ragged tile: values >= n contribute nothing
                        r[Lg] = m;
                    }
                    transpose4x4_bytes(r, c); // c[b] byte Lg = key-byte b of value 8 Lg + i
#pragma unroll
                    for (int b = 0; b < 4; b++) R[b][i] = c[b];
                }
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if ((uint32_t)b < nb) transpose_bits_8regs(R[b]);
#pragma unroll
                    for (int q = 0; q < 8; q++) outw[b][q][j] = R[b][q];
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t pass = p32 * 4 + b;
                if (pass * 8 < P) {
                    if constexpr (reg_counts) {
#pragma unroll
                        for (int q = 0; q < 8; q++)
#pragma unroll
                            for (int j = 0; j < WORDS; j++) acc[b][q] += __builtin_popcount(outw[b][q][j]);
                    } else if constexpr (RC == 2) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            uint32_t c0 = 0, c1 = 0;
#pragma unroll
                            for (int j = 0; j < WORDS; j++) {
                                c0 += __builtin_popcount(outw[b][2 * q][j]);
                                c1 += __builtin_popcount(outw[b][2 * q + 1][j]);
                            }
                            acc2[pi][b][q] += c0 | (c1 << 16);
                        }
                    } else if (a.hits && !use_hist) {
                        uint32_t cnt[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            cnt[q] = 0;
#pragma unroll
                            for (int j = 0; j < WORDS; j++) cnt[q] += __builtin_popcount(outw[b][q][j]);
                        }
                        block_hits_add8(s_hits, pass * 8, P, cnt, lane);
                    }
                    uint8_t *dst = a.out + (uint64_t)(pass * 8) * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                    if (full) {
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            if (pass * 8 + q < P) store_words<WORDS, NTS>(dst, outw[b][q]);
                            dst += a.out_stride;
                        }
                    } else {
                        const int nbytes = (valid + 7) / 8; // the bitmap is written byte-exact
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            if (pass * 8 + q < P) {
#pragma unroll
                                for (int bb = 0; bb < WORDS * 4; bb++)
                                    if (bb < nbytes) dst[bb] = (uint8_t)(outw[b][q][bb >> 2] >> (8 * (bb & 3)));
                            }
                            dst += a.out_stride;
                        }
                    }
                }
            }
        };
        if constexpr (RC == 2) { // two rounds at most, unrolled: the round indexes acc2
#pragma unroll
            for (uint32_t pi = 0; pi < 2; pi++)
                if (pi < npass32) do_round(pi);
            if (++tiles_counted == kPackedFlushTiles) flush_packed();
        } else {
            for (uint32_t pi = 0; pi < npass32; pi++) do_round(pi);
        }
        tile = next;
    }
    if constexpr (RC == 2) flush_packed();
    if constexpr (reg_counts) {
        // a lane's count is below 2^32 (it sees at most n / 64 values); the wave's sum need not be: reduce in 64 bits
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if ((uint32_t)(8 * b + q) < P) {
                    const uint32_t lo = wave_sum(acc[b][q] & 0xffffu), hi = wave_sum(acc[b][q] >> 16);
                    if (lane == 0) {
                        const unsigned long long v = (unsigned long long)lo + ((unsigned long long)hi << 16);
                        if (v)
                            __hip_atomic_fetch_add(a.scratch + (blockIdx.x % kHitSlots) * kMaxKeys + 8 * b + q, v, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
    }
    if constexpr (HIST) {
        if (a.hits && use_hist) {
            __syncthreads(); // every wave's histogram adds are done
            for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
                const uint32_t key = (uint32_t)a.keys_dev[k];
                s_hits[k] = (key >> C) == 0 ? hist[key] : 0u; // keys outside [0, 2^C) match nothing
            }
        }
    }
    if (a.hits) block_hits_flush(a, s_hits, P);
    hits_finalize(a, P, lane);
}

// ---- shared scan, linear layout, P = 9 .. 1024: lanes in memory order ------------------------------------------------
// The linear output (byte of 8-value group g and key k at g*P + k, src/simd_scan_shared_linear.cpp:57) of a tile is ONE
// contiguous block of 512 rows x P bytes.  shared_wide_kernel gives every lane 8 consecutive rows, so a store instruction
// writes 64 pieces of 16 bytes 8P bytes apart: every 128-byte line is written in 8 or more separate visits, and lines
// that leave the L2 partly written cost a read-modify-write in ECC-protected HBM (the same effect made per-predicate
// bitmaps at a non-line-multiple stride up to 2x slower).  Here the lanes of a wave own CONSECUTIVE pieces of the
// output: a row's P bytes are T = ceil(P/32) pieces of 32 bytes (the last one P - 32 (T-1) bytes), one per table of 32
// keys; lane l of a step has table l mod T and row l / T (the 64 mod T lanes behind the last whole row idle: none when
// T is a power of two) -- so a wave step writes 64 / T whole rows back to back, 2 KiB
// contiguous when P is a multiple of 32.  A lane fetches the c bytes of its row from the tile's LDS image (lanes of one row
// read the same words: broadcast), shifts them into place with byte-granular funnel shifts and decodes the 8 values at
// compile-time offsets; 8 lookups in ITS table and one 8 x 32 bit transpose (group form) give the 32 bytes.  Pieces
// start at any byte when P is not a multiple of 16: plain unaligned 16-byte stores (the hardware runs in unaligned-access
// mode), a short last piece as two overlapping stores (store_row_piece).  Hit counts: per-byte population counts of the 32 result bytes,
// summed in packed byte counters per lane (a lane keeps its table from step to step), flushed to the block's LDS counters
// every 31 steps; from P = 128 on (c <= 12) a histogram of the values instead.
// RP = rows per 32-byte piece: 1 (a piece = up to 32 keys of one row), or 2 for P = 16 (a piece = two whole rows).
struct __attribute__((packed, aligned(1))) Unaligned8 {
    uint32_t a, b;
};
struct __attribute__((packed, aligned(1))) Unaligned4 {
    uint32_t a;
};
struct __attribute__((packed, aligned(1))) Unaligned2 {
    uint16_t a;
};
// the first n (1..31) of the 32 bytes in y[], to any address: at most two store instructions from 4 bytes on -- the
// second one OVERLAPS the first and ends on the piece's last byte (same lane, same bytes: harmless), instead of a
// 16 + 8 + 4 + 2 + 1 ladder (P = 31: five stores per row, 2.8 TB/s against 4.1 at P = 32).  n is wave-uniform.
__device__ __forceinline__ void store_row_piece(uint8_t *p, const uint32_t (&y)[8], uint32_t n)
{
    if (n >= 16u) {
        *(Unaligned16 *)p = Unaligned16{y[0], y[1], y[2], y[3]};
        if (n > 16u) {
            const uint32_t o = n - 16u, k = o >> 2, sh = o & 3u; // the window starts at byte o = 4 k + sh
            uint32_t s0, s1, s2, s3, s4;
            if (k == 0) s0 = y[0], s1 = y[1], s2 = y[2], s3 = y[3], s4 = y[4];
            else if (k == 1) s0 = y[1], s1 = y[2], s2 = y[3], s3 = y[4], s4 = y[5];
            else if (k == 2) s0 = y[2], s1 = y[3], s2 = y[4], s3 = y[5], s4 = y[6];
            else s0 = y[3], s1 = y[4], s2 = y[5], s3 = y[6], s4 = y[7];
            *(Unaligned16 *)(p + o) = Unaligned16{__builtin_amdgcn_alignbyte(s1, s0, sh), __builtin_amdgcn_alignbyte(s2, s1, sh),
                                                  __builtin_amdgcn_alignbyte(s3, s2, sh), __builtin_amdgcn_alignbyte(s4, s3, sh)};
        }
    } else if (n >= 8u) {
        *(Unaligned8 *)p = Unaligned8{y[0], y[1]};
        if (n > 8u) {
            const uint32_t o = n - 8u, sh = o & 3u;
            const uint32_t s0 = o < 4u ? y[0] : y[1], s1 = o < 4u ? y[1] : y[2], s2 = o < 4u ? y[2] : y[3];
            *(Unaligned8 *)(p + o) = Unaligned8{__builtin_amdgcn_alignbyte(s1, s0, sh), __builtin_amdgcn_alignbyte(s2, s1, sh)};
        }
    } else if (n >= 4u) {
        *(Unaligned4 *)p = Unaligned4{y[0]};
        if (n > 4u) *(Unaligned4 *)(p + (n - 4u)) = Unaligned4{__builtin_amdgcn_alignbyte(y[1], y[0], n - 4u)};
    } else {
        if (n & 2u) *(Unaligned2 *)p = Unaligned2{(uint16_t)y[0]};
        if (n & 1u) p[n - 1u] = (uint8_t)(y[0] >> (8u * (n - 1u)));
    }
}

constexpr int kLinearImageBytes = 2048 + 64; // a wave-step's rows (<= 2 KiB) + the misalignment, rounded up

template <int C, int AUX_, int RP>
__global__ __launch_bounds__(kBlockThreads) void shared_linear_kernel(ScanArgs a)
{
    static_assert(RP == 1 || RP == 2, "RP");
    constexpr int VPL = 64; // tile geometry of the shared scans: 4096 values = 512 rows of 8
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C>;
    constexpr int AUX = AUX_ & 15;
    constexpr int ROWS = G::TILE_VALUES / 8;
    constexpr int RB = RP * C;                // bytes of packed data per piece
    constexpr int ROW_DW = (RB + 3) / 4;      // dwords holding them once shifted into place
    constexpr int LOAD_DW = (RB + 3 + 3) / 4; // dwords to fetch: the piece may start at any byte of a dword
    constexpr int KB = 4 / RP;                // key-bytes (8 keys each) per row inside a piece
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES + 16];
    // Rows whose length is not a multiple of 16 bytes: a wave-step's output -- (64 / T) whole rows, <= 2 KiB, contiguous in
    // memory -- is assembled in this wave-private LDS image first (the pieces written at their byte offsets: unaligned LDS
    // stores) and leaves as ALIGNED 16-byte stores; only the < 16 bytes in front of the first and behind the last aligned
    // chunk go out as single bytes.  Unaligned 16-byte global stores (every piece of such a row) cost a quarter to a third of
    // the kernel: 2.5e8 x 9 bit with hit counts, P = 63 / 64: 2.9 / 3.9 TB/s, 257 / 256: 3.5 / 4.5, 385 / 384: 2.9 / 4.0.
    // The image lives in DYNAMIC LDS behind the tables, kLinearImageBytes per wave, and only when the launcher asks for it (flags bit
    // 20): as a static array it cost every launch of this kernel 8.4 KiB and with them a resident block at many key counts
    // (aligned rows, same box: P = 128 0.96 -> 1.21 ms, P = 192 1.38 -> 1.83).
    __shared__ uint32_t s_hits[kMaxKeys];
    constexpr bool HIST = C <= 12; // hit counts by histogram of the values (see shared_wide2_kernel), else packed byte counters
    __shared__ uint32_t hist[HIST ? (1 << C) : 1];
    uint32_t *const lut = (uint32_t *)mi355_dyn_lds; // ceil(P/32) tables of TABLE_BYTES
    for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxKeys; k += kBlockThreads) s_hits[k] = 0;
    if constexpr (HIST)
        for (uint32_t k = threadIdx.x; k < (1u << C); k += kBlockThreads) hist[k] = 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t T = RP == 1 ? (P + 31) / 32 : 1;                 // tables of 32 keys per row
    const uint32_t rows_per_step = 64u / T;                         // whole rows a wave step covers; 64 mod T lanes idle
    const uint32_t last_bytes = RP == 1 ? P - 32 * (T - 1) : 32;    // bytes of a row's last piece, 1..32
    const bool aligned16 = (P & 15u) == 0;                          // every piece starts on a 16-byte boundary
    // Where it pays: two lanes per row (P = 33 .. 63: a step's image is 32 rows, up to 2 KiB -- same box, with hit counts, P = 57 / 63:
    // 2.81 / 2.86 -> 3.50 / 3.77 TB/s).  One lane per row loses (P = 9: 2.80 -> 1.88, P = 24: 3.93 -> 3.62: small images, the
    // write -> wait -> read -> store chain per step is not hidden), and so do three and more (P = 65: 2.38 -> 2.02, 257: 3.15 -> 2.58:
    // 64 mod T idle lanes, images of 1.3 KiB and more steps per tile).  (flags bit 14: never, for A/B)
    // Single-table widths only: at c = 17 the image costs the kernel a resident block (P = 47: 2.62 -> 2.32 TB/s, 63: 2.96 -> 2.82).
    // Copying the image out a step late (below) left every one of these figures where it was -- with it and flags bit 19 (the image at
    // every T) P = 9 / 17 with hit counts run at 0.67 / 0.72 x, P = 300 / 511 at 0.75 / 0.70 x of the direct stores: the image's LDS
    // traffic (unaligned ds_write_b128 + the read back), not a wait, is what it costs.
    const bool staged = RP == 1 && (a.flags & 0x100000u); // (the launcher's rule: linear_image_wanted())
    uint8_t *const ostage_wave = mi355_dyn_lds + T * (uint32_t)(L::TABLE_DWORDS * 4) + (uint32_t)wave * kLinearImageBytes;
    // The image of step s leaves during step s + 1: its LDS reads are issued in front of that step's decode (a wave's LDS
    // operations execute in order, so they see step s's pieces and are not disturbed by step s + 1's, which follow them),
    // its global stores behind it -- the read latency passes behind the lookups and transposes instead of in front of the stores.
    uint8_t *pend_gal = nullptr; // 16-byte aligned global address of image byte 0
    uint32_t pend_a0 = 0, pend_end = 0; // image bytes [a0, end) wait to be copied out (end = 0: nothing)
    auto image_load = [&](uint32_t a0, uint32_t end, u32x4 (&v)[3], uint32_t &hb, uint32_t &tb) {
        const uint8_t *img = ostage_wave;
        const uint32_t c_lo = (a0 + 15u) / 16u, c_hi = end / 16u; // whole aligned chunks [c_lo, c_hi)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t c = c_lo + (uint32_t)lane + 64u * j;
            if (c < c_hi) v[j] = *(const u32x4 *)(img + 16u * c);
        }
        // the bytes in front of the first and behind the last whole chunk (fewer than 16 each)
        const uint32_t head_end = c_lo * 16u < end ? c_lo * 16u : end;
        if (a0 + (uint32_t)lane < head_end) hb = img[a0 + lane];
        const uint32_t tail0 = c_hi * 16u > head_end ? c_hi * 16u : head_end;
        if (tail0 + (uint32_t)lane < end) tb = img[tail0 + lane];
    };
    auto image_store = [&](uint8_t *g_al, uint32_t a0, uint32_t end, const u32x4 (&v)[3], uint32_t hb, uint32_t tb) {
        const uint32_t c_lo = (a0 + 15u) / 16u, c_hi = end / 16u;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t c = c_lo + (uint32_t)lane + 64u * j;
            if (c < c_hi) *(u32x4 *)(g_al + 16u * c) = v[j];
        }
        const uint32_t head_end = c_lo * 16u < end ? c_lo * 16u : end;
        if (a0 + (uint32_t)lane < head_end) g_al[a0 + lane] = (uint8_t)hb;
        const uint32_t tail0 = c_hi * 16u > head_end ? c_hi * 16u : head_end;
        if (tail0 + (uint32_t)lane < end) g_al[tail0 + lane] = (uint8_t)tb;
    };
    const bool use_hist = HIST && P >= 128;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    for (uint32_t i = threadIdx.x; i < T * L::TABLE_DWORDS; i += kBlockThreads) lut[i] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
        const uint32_t key = (uint32_t)a.keys_dev[k];
        const bool in_range = C == 32 || (key >> (C & 31)) == 0;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                __hip_atomic_fetch_or(lut + (k >> 5) * L::TABLE_DWORDS + d * L::ENTRIES + e, 1u << (k & 31), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // lane l has table l mod T and, in step s, row l / T + s * (64 / T): its table is the same in every step, and the
    // lanes of a step cover 64 / T consecutive rows completely (the 64 mod T lanes behind them sit the steps out: none
    // when T is a power of two; rounding T up to one instead idled up to 44 % of the wave -- T = 9: 2.2 -> 3.x TB/s)
    const uint32_t quarter = (uint32_t)lane % T;
    const bool has_table = (uint32_t)lane < rows_per_step * T;
    const uint32_t row_first = ((uint32_t)lane / T) * RP;
    const uint32_t row_step = rows_per_step * RP;
    const uint32_t nsteps = ((uint32_t)ROWS + row_step - 1) / row_step;
    const uint32_t *const table = lut + (has_table ? quarter : 0u) * L::TABLE_DWORDS;
    const uint32_t piece_bytes = quarter + 1 == T ? last_bytes : 32u;
    // histogram hit counts: the T lanes of a row share its values -- value i belongs to the lane of table i mod T
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 8 * RP; i++)
        if ((uint32_t)i % T == quarter) mine |= 1u << i;
    // packed per-byte hit counters of the piece's 32 result bytes: at most 8 per step, so they are flushed to the block's
    // LDS counters every 31 steps -- counted ACROSS tiles (a tile is only 4 steps at P = 16, 8 at P = 32: flushing per
    // tile cost 32 LDS atomics per lane every 4 steps, a third of the kernel's time with hit counts)
    uint32_t cb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cb[i] = 0;
    uint32_t since_flush = 0;
    auto flush_counts = [&]() {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int bq = 0; bq < 4; bq++) {
                const uint32_t v = (cb[i] >> (8 * bq)) & 0xffu;
                // result dword i of the piece holds keys 4 (i mod 2 KB) .. +3 of the lane's table
                if (v) atomicAdd(&s_hits[32 * quarter + 4 * (i % (2 * KB)) + bq], v);
            }
            cb[i] = 0;
        }
        since_flush = 0;
    };

    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // rows of the tile that exist (the ragged tile: fewer, the last one possibly with fewer than 8 values)
        const uint64_t left = tc.n - tile * G::TILE_VALUES;
        const uint32_t vals_here = left >= (uint64_t)G::TILE_VALUES ? (uint32_t)G::TILE_VALUES : (uint32_t)left;
        const uint32_t rows_here = (vals_here + 7) / 8;
        uint8_t *const out_tile = a.out + tile * (uint64_t)ROWS * P;
#pragma unroll 1
        for (uint32_t s = 0; s < nsteps; s++) {
            const uint32_t row = row_first + s * row_step;
            // the step's output: rows [s * row_step, ...) of the tile, contiguous bytes from step_g0 on
            uint8_t *const step_g0 = out_tile + (uint64_t)(s * row_step) * P;
            const uint32_t step_a0 = (uint32_t)((uintptr_t)step_g0 & 15u);
            u32x4 iv[3];
            uint32_t ihb = 0, itb = 0;
            const bool copy_out = pend_end != 0; // wave-uniform
            if (copy_out) image_load(pend_a0, pend_end, iv, ihb, itb);
            asm volatile("" ::: "memory");
            if (row < rows_here && has_table) { // (rows beyond the column: nothing is written)
                // the piece's RB bytes start at byte row * C of the tile: fetch the dwords around them, shift into place
                const uint32_t byte0 = row * C;
                const uint32_t *src = (const uint32_t *)(lds_wave + (byte0 & ~3u));
                uint32_t d[LOAD_DW];
#pragma unroll
                for (int i = 0; i < LOAD_DW; i++) d[i] = src[i];
                const uint32_t sh = (byte0 & 3u) * 8u;
                uint32_t w[ROW_DW];
#pragma unroll
                for (int i = 0; i < ROW_DW; i++) w[i] = (i + 1 < LOAD_DW) ? __builtin_amdgcn_alignbit(d[i + 1], d[i], sh) : (d[i] >> sh);
                uint32_t x[8 * RP];
                extract_all<C, 8 * RP, 0, ROW_DW>(w, x);
                const uint32_t nvalid = vals_here - row * 8 >= 8 * RP ? 8u * RP : vals_here - row * 8; // short only at the column's end
                uint32_t y[8];
#pragma unroll
                for (int h = 0; h < RP; h++) {
                    uint32_t m[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) m[i] = L::lookup(table, x[8 * h + i]);
                    if (vals_here != (uint32_t)G::TILE_VALUES) { // wave-uniform: only the column's last tile can hold a short row
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            if ((uint32_t)(8 * h + i) >= nvalid) m[i] = 0u;
                    }
                    const uint32_t r0[4] = {m[0], m[1], m[2], m[3]}, r1[4] = {m[4], m[5], m[6], m[7]};
                    uint32_t lo4[4], hi4[4];
                    transpose4x4_bytes(r0, lo4);
                    transpose4x4_bytes(r1, hi4);
#pragma unroll
                    for (int b = 0; b < KB; b++) {
                        uint32_t lo = 0, hi = 0;
                        // (a short piece -- P = 9: 9 of 32 bytes -- skips the key-bytes behind its end)
                        if (RP == 2 || 8u * b < piece_bytes) {
                            lo = lo4[b], hi = hi4[b];
                            transpose8x8(lo, hi);
                        }
                        y[2 * KB * h + 2 * b] = lo;
                        y[2 * KB * h + 2 * b + 1] = hi;
                    }
                }
                if constexpr (RP == 2) {
                    u32x4 *dst = (u32x4 *)(out_tile + ((uint64_t)s * 64 + lane) * 32);
                    dst[0] = u32x4{y[0], y[1], y[2], y[3]};
                    // (the second row of the piece; it exists unless the column ends on the first)
                    if (row + 1 < rows_here) dst[1] = u32x4{y[4], y[5], y[6], y[7]};
                } else if (staged) {
                    // the piece at its byte offset inside the step's image (which starts at the step's global misalignment, so
                    // that aligned chunks of the image are aligned chunks of memory)
                    uint8_t *dst = ostage_wave + step_a0 + (row - s * row_step) * P + 32u * quarter;
                    if (piece_bytes == 32) {
                        *(Unaligned16 *)dst = Unaligned16{y[0], y[1], y[2], y[3]};
                        *(Unaligned16 *)(dst + 16) = Unaligned16{y[4], y[5], y[6], y[7]};
                    } else {
                        store_row_piece(dst, y, piece_bytes);
                    }
                } else {
                    uint8_t *dst = out_tile + (uint64_t)row * P + 32u * quarter;
                    if (piece_bytes == 32 && aligned16) {
                        ((u32x4 *)dst)[0] = u32x4{y[0], y[1], y[2], y[3]};
                        ((u32x4 *)dst)[1] = u32x4{y[4], y[5], y[6], y[7]};
                    } else if (piece_bytes == 32) {
                        *(Unaligned16 *)dst = Unaligned16{y[0], y[1], y[2], y[3]};
                        *(Unaligned16 *)(dst + 16) = Unaligned16{y[4], y[5], y[6], y[7]};
                    } else {
                        store_row_piece(dst, y, piece_bytes);
                    }
                }
                if (HIST && use_hist) {
                    // (T >= 8: one value per lane, picked by a select chain -- one atomic instruction per step)
                    if (a.hits) {
                        if (T >= 8) {
                            uint32_t xi = x[0];
#pragma unroll
                            for (int i = 1; i < 8; i++) xi = quarter == (uint32_t)i ? x[i] : xi;
                            if (quarter < 8 && quarter < nvalid)
                                __hip_atomic_fetch_add(&hist[xi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else if (T >= 4) {
                            // four to seven lanes per row: values `quarter` and `quarter + T` -- two atomic instructions a step
                            // instead of eight with a few lanes each
                            uint32_t xa = x[0], xb = x[0];
                            const uint32_t ib = quarter + T;
#pragma unroll
                            for (int i = 1; i < 8; i++) {
                                xa = quarter == (uint32_t)i ? x[i] : xa;
                                xb = ib == (uint32_t)i ? x[i] : xb;
                            }
                            if (quarter < nvalid) __hip_atomic_fetch_add(&hist[xa], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (ib < 8u && ib < nvalid) __hip_atomic_fetch_add(&hist[xb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
#pragma unroll
                            for (int i = 0; i < 8 * RP; i++)
                                if (((mine >> i) & 1u) && (uint32_t)i < nvalid)
                                    __hip_atomic_fetch_add(&hist[x[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                } else if (a.hits) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (RP == 2 || 4u * i < piece_bytes) { // (dwords behind the end of a short piece hold no key)
                            uint32_t v = y[i];
                            v = v - ((v >> 1) & 0x55555555u);
                            v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
                            cb[i] += (v + (v >> 4)) & 0x0F0F0F0Fu;
                        }
                    }
                }
            }
            if (!use_hist && a.hits && ++since_flush == 31) flush_counts();
            if (copy_out) image_store(pend_gal, pend_a0, pend_end, iv, ihb, itb);
            pend_end = 0;
            if (staged) {
                const uint32_t first = s * row_step;
                const uint32_t rows_in_step = first >= rows_here ? 0u : (rows_here - first < row_step ? rows_here - first : row_step);
                if (rows_in_step) {
                    pend_gal = step_g0 - step_a0;
                    pend_a0 = step_a0;
                    pend_end = step_a0 + rows_in_step * P;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the tile's LDS reads are done: the next DMA may overwrite it
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        tile = next;
    }
    if (pend_end) { // the wave's last image
        u32x4 iv[3];
        uint32_t ihb = 0, itb = 0;
        image_load(pend_a0, pend_end, iv, ihb, itb);
        image_store(pend_gal, pend_a0, pend_end, iv, ihb, itb);
    }
    if (!use_hist && a.hits) flush_counts();
    if constexpr (HIST) {
        if (a.hits && use_hist) {
            __syncthreads(); // every wave's histogram adds are done
            for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
                const uint32_t key = (uint32_t)a.keys_dev[k];
                s_hits[k] = (key >> C) == 0 ? hist[key] : 0u; // keys outside [0, 2^C) match nothing
            }
        }
    }
    if (a.hits) block_hits_flush(a, s_hits, P);
    hits_finalize(a, P, lane);
}

// ---- shared scan, linear layout, any P = 9 .. 1024: full tables in memory order, the short last table on its own ---------
// shared_linear_kernel above gives every (row, table) pair a lane, the row's last table included even when it holds one
// key: P = 33 costs two lane-steps per row where P = 32 costs one (2.7 TB/s against 4.1), P = 65 three against two.  The
// kernel is bound by those lane-steps (VALU: decode, byte regrouping, 8x8 transposes, packed hit counters), not by memory.
// Here a row is Tf = P / 32 FULL pieces plus, when R = P mod 32 > 0, one SHORT piece of R bytes, and the two kinds never
// share a step:
//   * full pieces: lanes (row, table) in memory order as before (lane's table fixed, 64 mod Tf lanes idle);
//   * the short piece costs what its R keys need -- ceil(R / 8) of the four 8x8 transposes, 3 byte-gather operations instead
//     of 16 when R <= 8, one or two stores, ceil(R / 4) counters -- and
//       - Tf = 1 (P = 33 .. 63): the lane that has just done the row's full piece does the short one too, from the values
//         it already decoded (no second fetch / decode);
//       - Tf >= 2 or Tf = 0 (P = 9 .. 31): steps of their own with a lane per row (64 rows a step), run as soon as the full
//         steps have passed their rows, so that both kinds of piece reach a 128-byte line within a few steps of each
//         other (a tile's rows are 512 P bytes: written a phase apart, the first phase's partly written lines would
//         leave the L2 before the second one fills them).
// Piece sizes are wave-uniform in every step, so no lane ever waits for another lane's longer store ladder.
template <int C, int AUX_>
__global__ __launch_bounds__(kBlockThreads) void shared_linear2_kernel(ScanArgs a)
{
    constexpr int VPL = 64; // tile geometry of the shared scans: 4096 values = 512 rows of 8
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C>;
    constexpr int AUX = AUX_ & 15;
    constexpr int ROWS = G::TILE_VALUES / 8;
    constexpr int ROW_DW = (C + 3) / 4;      // dwords holding a row's C bytes once shifted into place
    constexpr int LOAD_DW = (C + 3 + 3) / 4; // dwords to fetch: the row may start at any byte of a dword
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES + 16];
    __shared__ uint32_t s_hits[kMaxKeys];
    constexpr bool HIST = C <= 12; // hit counts by histogram of the values (see shared_wide2_kernel), else packed byte counters
    __shared__ uint32_t hist[HIST ? (1 << C) : 1];
    uint32_t *const lut = (uint32_t *)mi355_dyn_lds; // ceil(P/32) tables of TABLE_BYTES
    for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxKeys; k += kBlockThreads) s_hits[k] = 0;
    if constexpr (HIST)
        for (uint32_t k = threadIdx.x; k < (1u << C); k += kBlockThreads) hist[k] = 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t Tf = P / 32, R = P % 32;                        // full tables; keys of the short last one
    const uint32_t T = Tf + (R ? 1u : 0u);
    // the short piece rides on the lane of the row's LAST full piece (Tf = 1; behind two to five full tables when it is short
    // enough for the launcher to ask for it: flags bit 17) ...
    const bool attached = R != 0 && (Tf == 1 || (Tf >= 2 && (a.flags & 0x20000u)));
    const bool dedicated = R != 0 && !attached;                    // ... or gets steps of its own
    const bool aligned16 = (P & 15u) == 0;                         // every full piece starts on a 16-byte boundary
    const bool use_hist = HIST && P >= 128;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    for (uint32_t i = threadIdx.x; i < T * L::TABLE_DWORDS; i += kBlockThreads) lut[i] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
        const uint32_t key = (uint32_t)a.keys_dev[k];
        const bool in_range = C == 32 || (key >> (C & 31)) == 0;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                __hip_atomic_fetch_or(lut + (k >> 5) * L::TABLE_DWORDS + d * L::ENTRIES + e, 1u << (k & 31), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // full pieces: lane l has table l mod Tf and, in step s, row l / Tf + s * (64 / Tf)
    const uint32_t Tq = Tf ? Tf : 1u;
    const uint32_t quarter = (uint32_t)lane % Tq;
    const uint32_t rows_per_step = 64u / Tq;
    const bool has_table = Tf != 0 && (uint32_t)lane < rows_per_step * Tq;
    const uint32_t row_first = (uint32_t)lane / Tq;
    const uint32_t nsteps_full = Tf ? ((uint32_t)ROWS + rows_per_step - 1) / rows_per_step : 0u;
    const uint32_t *const table_full = lut + (has_table ? quarter : 0u) * L::TABLE_DWORDS;
    const uint32_t *const table_short = lut + Tf * L::TABLE_DWORDS;
    // histogram hit counts: the Tf lanes of a row share its values -- value i belongs to the lane of table i mod Tf
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((uint32_t)i % Tq == quarter) mine |= 1u << i;
    // packed per-byte hit counters of a piece's result bytes (at most 8 per step: flushed to the block's LDS counters every
    // 31 steps, counted ACROSS tiles), one set for the lane's full table, one for the short one
    uint32_t cbf[8], cbs[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cbf[i] = cbs[i] = 0;
    uint32_t since_f = 0, since_s = 0;
    auto flush = [&](uint32_t (&cb)[8], uint32_t first_key, uint32_t &since) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int bq = 0; bq < 4; bq++) {
                const uint32_t v = (cb[i] >> (8 * bq)) & 0xffu;
                if (v) atomicAdd(&s_hits[first_key + 4 * i + bq], v); // result dword i of a piece = keys 4 i .. 4 i + 3 of its table
            }
            cb[i] = 0;
        }
        since = 0;
    };
    const bool count_packed = a.hits && !use_hist;

    // the row's C bytes start at byte row * C of the tile: fetch the dwords around them, shift into place, decode 8 values
    auto fetch = [&](uint32_t row, uint32_t (&x)[8]) {
        const uint32_t byte0 = row * C;
        const uint32_t *src = (const uint32_t *)(lds_wave + (byte0 & ~3u));
        uint32_t d[LOAD_DW];
#pragma unroll
        for (int i = 0; i < LOAD_DW; i++) d[i] = src[i];
        const uint32_t sh = (byte0 & 3u) * 8u;
        uint32_t w[ROW_DW];
#pragma unroll
        for (int i = 0; i < ROW_DW; i++) w[i] = (i + 1 < LOAD_DW) ? __builtin_amdgcn_alignbit(d[i + 1], d[i], sh) : (d[i] >> sh);
        extract_all<C, 8, 0, ROW_DW>(w, x);
    };
    // one piece: 8 lookups in `table`, regroup to key-bytes, 8x8 bit transposes of the key-bytes in use, store, count.
    // nbytes (1 .. 32) is wave-uniform.
    bool ragged_tile = false;
    auto piece = [&](const uint32_t (&x)[8], const uint32_t *table, uint32_t nbytes, uint32_t nvalid, uint8_t *dst, uint32_t (&cb)[8]) {
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = L::lookup(table, x[i]);
        if (ragged_tile) { // wave-uniform: only the column's last tile can hold a short row
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((uint32_t)i >= nvalid) m[i] = 0u;
        }
        uint32_t y[8];
        if (nbytes <= 8u) { // one key-byte: gather the low bytes of the 8 dwords
            uint32_t lo = __builtin_amdgcn_perm(m[1], m[0], 0x0c0c0400u) | __builtin_amdgcn_perm(m[3], m[2], 0x04000c0cu);
            uint32_t hi = __builtin_amdgcn_perm(m[5], m[4], 0x0c0c0400u) | __builtin_amdgcn_perm(m[7], m[6], 0x04000c0cu);
            transpose8x8(lo, hi);
            y[0] = lo;
            y[1] = hi;
#pragma unroll
            for (int i = 2; i < 8; i++) y[i] = 0;
        } else {
            const uint32_t r0[4] = {m[0], m[1], m[2], m[3]}, r1[4] = {m[4], m[5], m[6], m[7]};
            uint32_t lo4[4], hi4[4];
            transpose4x4_bytes(r0, lo4);
            transpose4x4_bytes(r1, hi4);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t lo = 0, hi = 0;
                if (8u * b < nbytes) { // (key-bytes behind the end of a short piece are neither transposed nor counted)
                    lo = lo4[b], hi = hi4[b];
                    transpose8x8(lo, hi);
                }
                y[2 * b] = lo;
                y[2 * b + 1] = hi;
            }
        }
        if (nbytes == 32u) {
            if (aligned16) {
                ((u32x4 *)dst)[0] = u32x4{y[0], y[1], y[2], y[3]};
                ((u32x4 *)dst)[1] = u32x4{y[4], y[5], y[6], y[7]};
            } else {
                *(Unaligned16 *)dst = Unaligned16{y[0], y[1], y[2], y[3]};
                *(Unaligned16 *)(dst + 16) = Unaligned16{y[4], y[5], y[6], y[7]};
            }
        } else {
            store_row_piece(dst, y, nbytes);
        }
        if (count_packed) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (4u * i < nbytes) {
                    uint32_t v = y[i];
                    v = v - ((v >> 1) & 0x55555555u);
                    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
                    cb[i] += (v + (v >> 4)) & 0x0F0F0F0Fu;
                }
            }
        }
    };

    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // rows of the tile that exist (the ragged tile: fewer, the last one possibly with fewer than 8 values)
        const uint64_t left = tc.n - tile * G::TILE_VALUES;
        const uint32_t vals_here = left >= (uint64_t)G::TILE_VALUES ? (uint32_t)G::TILE_VALUES : (uint32_t)left;
        const uint32_t rows_here = (vals_here + 7) / 8;
        ragged_tile = vals_here != (uint32_t)G::TILE_VALUES;
        uint8_t *const out_tile = a.out + tile * (uint64_t)ROWS * P;
        // one step of short pieces: rows [64 j, 64 j + 64), a lane per row
        auto short_step = [&](uint32_t j) {
            const uint32_t row = 64u * j + (uint32_t)lane;
            if (row < rows_here) {
                uint32_t x[8];
                fetch(row, x);
                const uint32_t nvalid = vals_here - row * 8 >= 8u ? 8u : vals_here - row * 8;
                piece(x, table_short, R, nvalid, out_tile + (uint64_t)row * P + 32u * Tf, cbs);
            }
            if (count_packed && ++since_s == 31) flush(cbs, 32u * Tf, since_s);
        };
        uint32_t short_done = 0; // 64-row blocks whose short pieces are written
#pragma unroll 1
        for (uint32_t s = 0; s < nsteps_full; s++) {
            const uint32_t row = row_first + s * rows_per_step;
            if (row < rows_here && has_table) { // (rows beyond the column: nothing is written)
                uint32_t x[8];
                fetch(row, x);
                const uint32_t nvalid = vals_here - row * 8 >= 8u ? 8u : vals_here - row * 8; // short only at the column's end
                piece(x, table_full, 32u, nvalid, out_tile + (uint64_t)row * P + 32u * quarter, cbf);
                if (attached && quarter + 1 == Tq) piece(x, table_short, R, nvalid, out_tile + (uint64_t)row * P + 32u * Tf, cbs);
                if (HIST && use_hist && a.hits) {
                    // (Tf >= 8: one value per lane, picked by a select chain -- one atomic instruction per step)
                    if (Tf >= 8) {
                        uint32_t xi = x[0];
#pragma unroll
                        for (int i = 1; i < 8; i++) xi = quarter == (uint32_t)i ? x[i] : xi;
                        if (quarter < 8 && quarter < nvalid)
                            __hip_atomic_fetch_add(&hist[xi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (Tq >= 4) { // values `quarter` and `quarter + Tq`: two atomic instructions a step instead of eight
                        uint32_t xa = x[0], xb = x[0];
                        const uint32_t ib = quarter + Tq;
#pragma unroll
                        for (int i = 1; i < 8; i++) {
                            xa = quarter == (uint32_t)i ? x[i] : xa;
                            xb = ib == (uint32_t)i ? x[i] : xb;
                        }
                        if (quarter < nvalid) __hip_atomic_fetch_add(&hist[xa], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (ib < 8u && ib < nvalid) __hip_atomic_fetch_add(&hist[xb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            if (((mine >> i) & 1u) && (uint32_t)i < nvalid)
                                __hip_atomic_fetch_add(&hist[x[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (count_packed) {
                if (++since_f == 31) flush(cbf, 32u * quarter, since_f);
                if (attached && ++since_s == 31) flush(cbs, 32u * Tf, since_s);
            }
            if (dedicated) { // short pieces of the 64-row blocks the full steps have passed
                const uint32_t covered = (s + 1) * rows_per_step;
                while (64u * (short_done + 1) <= covered && short_done < (uint32_t)ROWS / 64u) short_step(short_done++);
            }
        }
        if (dedicated)
            while (short_done < (uint32_t)ROWS / 64u) short_step(short_done++);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the tile's LDS reads are done: the next DMA may overwrite it
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        tile = next;
    }
    if (count_packed) {
        if (has_table) flush(cbf, 32u * quarter, since_f);
        if (R != 0) flush(cbs, 32u * Tf, since_s);
    }
    if constexpr (HIST) {
        if (a.hits && use_hist) {
            __syncthreads(); // every wave's histogram adds are done
            for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
                const uint32_t key = (uint32_t)a.keys_dev[k];
                s_hits[k] = (key >> C) == 0 ? hist[key] : 0u; // keys outside [0, 2^C) match nothing
            }
        }
    }
    if (a.hits) block_hits_flush(a, s_hits, P);
    hits_finalize(a, P, lane);
}

// (Round 3 also tried tables kept ALREADY TRANSPOSED -- Entry[v] of 16 keys = 16 bytes, byte j = 1 iff v == key[j], so that a row's
// result bytes are OR_i Entry[v_i] << i: 16 ds_read_b128 + 64 v_lshl_or_b32 per row and 32 keys instead of 8 ds_read_b32 + ~120
// VALU operations for regrouping and bit transposes.  It LOSES everywhere: random 16-byte LDS reads run at ~40 cycles per wave
// instruction (bank conflicts), 8x the read traffic made the LDS the bottleneck -- 2.5e8 x 9 bit, linear, same box, with hit
// counts: P = 16 3.06 against 4.23 TB/s, P = 64 2.86 against 3.97; profiles/r03_spread_tables_linear_negative.txt.  Removed.)

// ---- shared scan, per-predicate bitmaps, digit-table widths (c > 10): one 32-value word at a time ---------------------------
// shared_wide2_kernel keeps a lane's 64 decoded values, the 64 result words of a 32-key round and the 32 registers of the
// transposes live at once: with three or four digits per lookup the compiler needs 256 VGPRs + 60 .. 110 AGPRs for it -- ONE
// wave per SIMD, every LDS lookup -> AND -> transpose chain exposed (profiles/r03_wide_widths_before.txt: 2.6 .. 3.2 TB/s at
// c = 17 .. 25, P = 64, waves waiting 40 .. 47 % of their cycles).  Here the word loop is OUTSIDE the rounds: the 32 values
// of a word are decoded from the lane's raw dwords, and each round's lookups, byte regrouping and register transposes end
// in the 32 keys' bitmap words of that ONE word, which are counted and stored (4 bytes per lane and key: a wave writes 256
// contiguous bytes per key) straight from the transposed registers -- half the live state, several waves per SIMD.
// Hit counts only in registers: RC = 1 (P <= 32, a register per key) or 2 (P <= 64, two keys per register in 16-bit halves,
// flushed every kPackedFlushTiles tiles); RC = 0 counts nothing (the launcher sends scans WITH hit counts over more than 64
// keys to shared_wide2_kernel).  BIG: WideLutGeom's wider digits.
template <int C, int BASE, int N, int K, int NW> __device__ __forceinline__ void extract_range(const uint32_t (&w)[NW], uint32_t (&x)[N])
{
    x[K] = extract<C, BASE + K, NW>(w);
    if constexpr (K + 1 < N) extract_range<C, BASE, N, K + 1, NW>(w, x);
}

// (waves per SIMD asked of the register allocator: 3 without counters and at the single-table widths; with counters at the
// digit-table widths 2 -- at 3 it spilled 90 .. 430 bytes per lane to scratch and the counted scans of c = 21 / 25 ran at
// 2.2 .. 3.3 TB/s where the uncounted ones reached 4.6 .. 5.4)
template <int C, int AUX_, int RC, bool BIG>
__global__ __launch_bounds__(kBlockThreads, ((RC == 0 || C <= 10) ? 3 : 2)) void shared_wide3_kernel(ScanArgs a)
{
    constexpr int VPL = 64;
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C, BIG>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 32) ? 2 : ((AUX_ & 16) ? 1 : 0); // result stores: 1 non-temporal, 2 write-through (sc1)
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    uint32_t *const lut = (uint32_t *)mi355_dyn_lds; // ceil(P/32) * TABLE_BYTES dynamic bytes

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t npass32 = (P + 31) / 32;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    for (uint32_t i = threadIdx.x; i < npass32 * L::TABLE_DWORDS; i += kBlockThreads) lut[i] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
        const uint32_t key = (uint32_t)a.keys_dev[k];
        const bool in_range = C == 32 || (key >> (C & 31)) == 0;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                __hip_atomic_fetch_or(lut + (k >> 5) * L::TABLE_DWORDS + d * L::ENTRIES + e, 1u << (k & 31), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // RC == 1: acc[0][b][q] = hits of key 8 b + q.  RC == 2: acc[p][b][q], q < 4: keys 32 p + 8 b + 2 q (low half), + 1 (high half)
    constexpr int NR = RC == 2 ? 2 : 1;
    uint32_t acc[NR][4][RC == 1 ? 8 : 4];
#pragma unroll
    for (int p = 0; p < NR; p++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int q = 0; q < (RC == 1 ? 8 : 4); q++) acc[p][b][q] = 0;
    uint32_t tiles_counted = 0;
    auto flush_counts = [&]() {
        if constexpr (RC != 0) {
#pragma unroll
            for (int p = 0; p < NR; p++)
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int q = 0; q < (RC == 1 ? 8 : 4); q++) {
                        // (a lane's count stays below 2^32 / 2^16; the wave's sum need not: two halves)
                        const uint32_t lo = wave_sum(acc[p][b][q] & 0xffffu), hi = wave_sum(acc[p][b][q] >> 16);
                        acc[p][b][q] = 0;
                        if (lane == 0) {
                            unsigned long long *slot = a.scratch + (blockIdx.x % kHitSlots) * kMaxKeys + 32 * p + 8 * b + (RC == 1 ? q : 2 * q);
                            if constexpr (RC == 1) {
                                const unsigned long long v = (unsigned long long)lo + ((unsigned long long)hi << 16);
                                if (v) __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                if (lo) __hip_atomic_fetch_add(slot, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (hi) __hip_atomic_fetch_add(slot + 1, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
        }
        tiles_counted = 0;
    };

    uint32_t behind = 0; // stores issued after the DMA the next wait is for (wait_dma_behind_stores)
    while (tile < tc.ntiles) {
        wait_dma_behind_stores(behind);
        const uint64_t next = tile + stride;
        const bool full = tile < tc.nfull;
        // (the tile's first word is stored before the next tile's DMA is issued, its second word behind it)
        behind = full ? P : 0;
        const uint64_t tile_left = tc.n - tile * G::TILE_VALUES; // values of the column from the tile's first on
        uint8_t *const out_tile = a.out + tile * G::BITMAP_BYTES;

        auto do_word = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            // Phase j: lane l takes the tile's 32-value word 64 j + l -- C whole dwords of the LDS image -- so that the 64
            // lanes' result words of a key are 256 CONTIGUOUS bytes of its bitmap (two whole lines per store instruction.
            // A lane owning 64 consecutive values, as everywhere else, would write its two words 4 bytes at a time, 8 bytes
            // apart: half-written lines, 1.4 TB/s).  The dwords are fetched when the phase's turn comes (all 2 C at once cost
            // c >= 27 a wave per SIMD); the next tile's DMA may overwrite the tile once the LAST phase's dwords are in registers.
            const uint32_t word = 64u * j + (uint32_t)lane;
            const int64_t left = (int64_t)tile_left - (int64_t)word * 32;
            const int valid = left >= 32 ? 32 : (left <= 0 ? 0 : (int)left); // values of the word inside the column
            uint8_t *const out_lane = out_tile + word * 4u;
            uint32_t wj[C];
            {
                const uint32_t *src = (const uint32_t *)lds_wave + word * (uint32_t)C;
#pragma unroll
                for (int q = 0; q < C; q++) wj[q] = src[q];
            }
            if constexpr (j == WORDS - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
            }
            uint32_t x[32];
            extract_range<C, 0, 32, 0, C>(wj, x);
            auto do_round = [&](const uint32_t pi) __attribute__((always_inline)) {
                const uint32_t *table = lut + pi * L::TABLE_DWORDS;
                const uint32_t nb = ((P - pi * 32) < 32 ? (P - pi * 32) + 7 : 39) / 8; // key-bytes in use, 1..4
                uint32_t R[4][8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint32_t r[4], c[4];
#pragma unroll
                    for (int Lg = 0; Lg < 4; Lg++) {
                        const int v = 8 * Lg + i;
                        uint32_t m = L::lookup(table, x[v]);
                        if (!full) m = v < valid ? m : 0u; // ragged tile: values >= n contribute nothing
                        r[Lg] = m;
                    }
                    transpose4x4_bytes(r, c); // c[b] byte Lg = key-byte b of value 8 Lg + i
#pragma unroll
                    for (int b = 0; b < 4; b++) R[b][i] = c[b];
                    // (4 x ND lookups in flight are enough: let the scheduler hoist all 32 x ND of a word and the four-digit
                    // widths need 256 VGPRs + AGPRs again)
                    if constexpr (L::ND >= 3) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if ((uint32_t)b < nb) { // wave-uniform
                        transpose_bits_8regs(R[b]); // R[b][q] = this word of key 32 pi + 8 b + q
                        if constexpr (RC == 1) {
#pragma unroll
                            for (int q = 0; q < 8; q++) acc[0][b][q] += __builtin_popcount(R[b][q]);
                        } else if constexpr (RC == 2) {
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                acc[pi][b][q] += (uint32_t)__builtin_popcount(R[b][2 * q]) | ((uint32_t)__builtin_popcount(R[b][2 * q + 1]) << 16);
                        }
                        uint8_t *dst = out_lane + (uint64_t)(pi * 32 + 8 * b) * a.out_stride;
                        if (full) {
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                if (pi * 32 + 8 * b + q < P) {
                                    const uint32_t one[1] = {R[b][q]};
                                    store_words<1, NTS>(dst, one);
                                }
                                dst += a.out_stride;
                            }
                        } else {
                            const int nbytes = (valid + 7) / 8; // the bitmap is written byte-exact
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                if (pi * 32 + 8 * b + q < P) {
#pragma unroll
                                    for (int bb = 0; bb < 4; bb++)
                                        if (bb < nbytes) dst[bb] = (uint8_t)(R[b][q] >> (8 * bb));
                                }
                                dst += a.out_stride;
                            }
                        }
                    }
                }
            };
            if constexpr (RC == 2) { // two rounds at most, unrolled: the round indexes acc
#pragma unroll
                for (uint32_t pi = 0; pi < 2; pi++)
                    if (pi < npass32) do_round(pi);
            } else {
#pragma unroll 1
                for (uint32_t pi = 0; pi < npass32; pi++) do_round(pi);
            }
        };
        do_word(std::integral_constant<int, 0>{});
        do_word(std::integral_constant<int, 1>{});
        static_assert(WORDS == 2, "two words per lane");
        if constexpr (RC != 0) {
            if (++tiles_counted == kPackedFlushTiles) flush_counts();
        }
        tile = next;
    }
    if constexpr (RC != 0) flush_counts();
    hits_finalize(a, P, lane);
}

// ---- shared scan, linear layout, P = 9 .. 64: the per-predicate machinery + an LDS stage ---------------------------------
// shared_linear_kernel gives a lane one row of 8 values and pays, per row and 32 keys, 16 v_perm_b32 and four 8x8 bit
// transposes in register pairs (~25 operations each) plus packed byte counters: 34 VALU operations per value and 32 keys,
// VALU-bound at 3-4 TB/s (profiles/r03_shared_linear_c9_profile.txt).  The per-predicate kernels get the same bits for ~20:
// a lane takes a whole 32-value word, the eight-register transposes handle 32 values x 8 keys in 72 operations, and the
// hit counts are one v_bcnt per key word.  This kernel does exactly that (the body of shared_wide3_kernel: phase j, lane l
// = the tile's word 64 j + l = rows 4 (64 j + l) .. + 3) and THEN turns the 32 key words of a round back into the rows'
// bytes: eight 4x4 byte transposes (64 v_perm_b32 per 32 values x 32 keys) give every one of the lane's four rows its 32
// bytes of the round, which go to a wave-private LDS stage of 256 rows x 32 bytes and leave it in row order -- consecutive
// lanes write consecutive 16-byte units of the output, 1 KiB per store instruction at P = 32 -- however P relates to 16.
// The stage's 16-byte units are XOR-swizzled (unit u lives at u ^ ((u >> 3) & 15)): both the writers (lane l: units
// 8 l + 2 r + h) and the readers (lane i: units i + 64 s) then spread over the 16 unit positions of the LDS's 64 banks four
// lanes apiece, which is the minimum for 16-byte accesses.
template <int C, int AUX_, int RC, bool BIG>
__global__ __launch_bounds__(kBlockThreads, 2) void shared_linear3_kernel(ScanArgs a)
{
    constexpr int VPL = 64;
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C, BIG>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr int STAGE_ROWS = 256; // rows of a phase: 64 lanes x 4
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) u32x4 stage_all[kWavesPerBlock][STAGE_ROWS * 2]; // 32 bytes per row
    uint32_t *const lut = (uint32_t *)mi355_dyn_lds; // ceil(P/32) * TABLE_BYTES dynamic bytes

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    u32x4 *const stage = stage_all[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;
    const uint32_t npass32 = (P + 31) / 32;
    const bool aligned16 = (P & 15u) == 0;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
    for (uint32_t i = threadIdx.x; i < npass32 * L::TABLE_DWORDS; i += kBlockThreads) lut[i] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
        const uint32_t key = (uint32_t)a.keys_dev[k];
        const bool in_range = C == 32 || (key >> (C & 31)) == 0;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < L::ND; d++) {
                const uint32_t e = L::SINGLE ? key : L::digit(key, d);
                __hip_atomic_fetch_or(lut + (k >> 5) * L::TABLE_DWORDS + d * L::ENTRIES + e, 1u << (k & 31), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // hit counts in registers, as in shared_wide3_kernel
    constexpr int NR = RC == 2 ? 2 : 1;
    uint32_t acc[NR][4][RC == 1 ? 8 : 4];
#pragma unroll
    for (int p = 0; p < NR; p++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int q = 0; q < (RC == 1 ? 8 : 4); q++) acc[p][b][q] = 0;
    uint32_t tiles_counted = 0;
    auto flush_counts = [&]() {
        if constexpr (RC != 0) {
#pragma unroll
            for (int p = 0; p < NR; p++)
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int q = 0; q < (RC == 1 ? 8 : 4); q++) {
                        const uint32_t lo = wave_sum(acc[p][b][q] & 0xffffu), hi = wave_sum(acc[p][b][q] >> 16);
                        acc[p][b][q] = 0;
                        if (lane == 0) {
                            unsigned long long *slot = a.scratch + (blockIdx.x % kHitSlots) * kMaxKeys + 32 * p + 8 * b + (RC == 1 ? q : 2 * q);
                            if constexpr (RC == 1) {
                                const unsigned long long v = (unsigned long long)lo + ((unsigned long long)hi << 16);
                                if (v) __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                if (lo) __hip_atomic_fetch_add(slot, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (hi) __hip_atomic_fetch_add(slot + 1, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
        }
        tiles_counted = 0;
    };
    auto swz = [](uint32_t u) -> uint32_t { return u ^ ((u >> 3) & 15u); };

    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        const bool full = tile < tc.nfull;
        const uint64_t tile_left = tc.n - tile * G::TILE_VALUES; // values of the column from the tile's first on
        uint8_t *const out_tile = a.out + tile * (uint64_t)(G::TILE_VALUES / 8) * P;

        auto do_word = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const uint32_t word = 64u * j + (uint32_t)lane;
            const int64_t left = (int64_t)tile_left - (int64_t)word * 32;
            const int valid = left >= 32 ? 32 : (left <= 0 ? 0 : (int)left); // values of the word inside the column
            // rows of this phase that exist (ragged tile: fewer, the last one possibly with fewer than 8 values)
            const int64_t phase_left = (int64_t)tile_left - (int64_t)j * (STAGE_ROWS * 8);
            const uint32_t rows_here = phase_left >= STAGE_ROWS * 8 ? (uint32_t)STAGE_ROWS : (phase_left <= 0 ? 0u : (uint32_t)((phase_left + 7) / 8));
            uint8_t *const out_phase = out_tile + (uint64_t)j * STAGE_ROWS * P;
            uint32_t wj[C];
            {
                const uint32_t *src = (const uint32_t *)lds_wave + word * (uint32_t)C;
#pragma unroll
                for (int q = 0; q < C; q++) wj[q] = src[q];
            }
            if constexpr (j == WORDS - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
            }
            uint32_t x[32];
            extract_range<C, 0, 32, 0, C>(wj, x);
            auto do_round = [&](const uint32_t pi) __attribute__((always_inline)) {
                const uint32_t *table = lut + pi * L::TABLE_DWORDS;
                const uint32_t nk = (P - pi * 32) < 32 ? (P - pi * 32) : 32; // keys of this round
                const uint32_t nb = (nk + 7) / 8;                            // key-bytes in use, 1..4
                uint32_t R[4][8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint32_t r[4], c[4];
#pragma unroll
                    for (int Lg = 0; Lg < 4; Lg++) {
                        const int v = 8 * Lg + i;
                        uint32_t m = L::lookup(table, x[v]);
                        if (!full) m = v < valid ? m : 0u; // ragged tile: values >= n contribute nothing
                        r[Lg] = m;
                    }
                    transpose4x4_bytes(r, c); // c[b] byte Lg = key-byte b of value 8 Lg + i
#pragma unroll
                    for (int b = 0; b < 4; b++) R[b][i] = c[b];
                    if constexpr (L::ND >= 3) __builtin_amdgcn_sched_barrier(0);
                }
                // the stage must be free: the previous round's copy-out reads are done (LDS is in order per wave, and those
                // reads were waited for before their stores were issued)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if ((uint32_t)(2 * h) < nb) { // wave-uniform
                        uint32_t lo[2][4], hi[2][4]; // [key-byte of the half][row]: bytes of keys 8 b .. + 3 / + 4 .. + 7
#pragma unroll
                        for (int bb = 0; bb < 2; bb++) {
                            const int b = 2 * h + bb;
                            if ((uint32_t)b < nb) {
                                transpose_bits_8regs(R[b]); // R[b][q] = this word of key 32 pi + 8 b + q: byte r = row r
                                if constexpr (RC == 1) {
#pragma unroll
                                    for (int q = 0; q < 8; q++) acc[0][b][q] += __builtin_popcount(R[b][q]);
                                } else if constexpr (RC == 2) {
#pragma unroll
                                    for (int q = 0; q < 4; q++)
                                        acc[pi][b][q] += (uint32_t)__builtin_popcount(R[b][2 * q]) | ((uint32_t)__builtin_popcount(R[b][2 * q + 1]) << 16);
                                }
                                const uint32_t r0[4] = {R[b][0], R[b][1], R[b][2], R[b][3]}, r1[4] = {R[b][4], R[b][5], R[b][6], R[b][7]};
                                transpose4x4_bytes(r0, lo[bb]); // lo[bb][r] byte q = row r's byte of key 8 b + q
                                transpose4x4_bytes(r1, hi[bb]);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; r++) lo[bb][r] = hi[bb][r] = 0;
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            stage[swz(8u * lane + 2u * r + h)] = u32x4{lo[0][r], hi[0][r], lo[1][r], hi[1][r]};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the wave's stage writes have landed
                if (nk == 32u) {
                    // row order: lane i takes the 16-byte units i + 64 s; unit u = half (u & 1) of row u >> 1
                    uint8_t *dst = out_phase + (uint64_t)((uint32_t)lane >> 1) * P + 32u * pi + 16u * ((uint32_t)lane & 1u);
                    const uint64_t step = 32ull * P;
#pragma unroll
                    for (int sb = 0; sb < 8; sb += 4) {
                        u32x4 v[4];
#pragma unroll
                        for (int s = 0; s < 4; s++) v[s] = stage[swz((uint32_t)lane + 64u * (sb + s))];
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            if (((uint32_t)lane >> 1) + 32u * (sb + s) < rows_here) {
                                if (aligned16)
                                    *(u32x4 *)dst = v[s];
                                else
                                    *(Unaligned16 *)dst = Unaligned16{v[s].x, v[s].y, v[s].z, v[s].w};
                            }
                            dst += step;
                        }
                    }
                } else {
                    // the short last round: a lane per row, nk bytes each
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const uint32_t row = (uint32_t)lane + 64u * s;
                        const u32x4 v0 = stage[swz(2u * row)];
                        u32x4 v1 = u32x4{0, 0, 0, 0};
                        if (nk > 16u) v1 = stage[swz(2u * row + 1u)];
                        const uint32_t y[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        if (row < rows_here) store_row_piece(out_phase + (uint64_t)row * P + 32u * pi, y, nk);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // stage reads done before the next round overwrites it
            };
            if constexpr (RC == 2) { // two rounds at most, unrolled: the round indexes acc
#pragma unroll
                for (uint32_t pi = 0; pi < 2; pi++)
                    if (pi < npass32) do_round(pi);
            } else {
#pragma unroll 1
                for (uint32_t pi = 0; pi < npass32; pi++) do_round(pi);
            }
        };
        do_word(std::integral_constant<int, 0>{});
        do_word(std::integral_constant<int, 1>{});
        static_assert(WORDS == 2, "two words per lane");
        if constexpr (RC != 0) {
            if (++tiles_counted == kPackedFlushTiles) flush_counts();
        }
        tile = next;
    }
    if constexpr (RC != 0) flush_counts();
    hits_finalize(a, P, lane);
}

} // namespace mi355

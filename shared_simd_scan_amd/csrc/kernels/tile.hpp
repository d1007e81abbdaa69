// kernels/tile.hpp -- tile geometry, LDS-DMA, value extraction, compare / table decode helpers, result stores, hit counts.  Part of kernels.hpp (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define MI355_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define MI355_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int kWavesPerBlock = 4;
constexpr int kBlockThreads = kWavesPerBlock * 64;
constexpr int kMaxKeysPerPass = 8;
constexpr int kMaxKeys = 1024;      // what the reference's linear_simple tops out at (src/simd_scan_shared_linear.cpp:78)

// ---- tile geometry of the scan kernels -----------------------------------------------------
// A wave owns a tile of 64*VPL consecutive values; lane l owns values [l*VPL, (l+1)*VPL) of it,
// i.e. VPL*C bits = LANE_DWORDS whole dwords starting on a dword boundary.  VPL in {32, 64, 128}
// selects the LDS read width (ds_read_b32 / b64 / b128) and the bitmap store width (4 / 8 / 16 B).
template <int C, int VPL> struct ScanGeom {
    static_assert(VPL == 32 || VPL == 64 || VPL == 128, "VPL");
    static constexpr int WORDS = VPL / 32;                 // bitmap dwords per lane per tile
    static constexpr int LANE_DWORDS = VPL * C / 32;       // packed dwords per lane
    static constexpr int LANE_BYTES = LANE_DWORDS * 4;
    static constexpr int TILE_VALUES = 64 * VPL;
    static constexpr int TILE_BYTES = 64 * LANE_BYTES;
    static constexpr int DMA_INSTRS = (TILE_BYTES + 1023) / 1024;
    static constexpr int LDS_BYTES = DMA_INSTRS * 1024;    // per wave
    static constexpr int BITMAP_BYTES = TILE_VALUES / 8;
    // waves per SIMD the LDS footprint admits (160 KiB per CU, 4 waves per block), capped at 8:
    // the register allocator is told to aim for that
    static constexpr int OCC_LDS = (160 * 1024) / (4 * LDS_BYTES + 64); // +64: the block's ticket word
    static constexpr int OCC = OCC_LDS >= 8 ? 8 : (OCC_LDS < 1 ? 1 : OCC_LDS);
};

enum ScanMode { kModeEq = 0, kModeRange = 1, kModeShared = 2 };

// occupancy target handed to the register allocator: what LDS admits, but the 8-key shared scan keeps
// 8 accumulators + 8x(VPL/32) result words + hit counters live and wants up to 128 VGPRs
// values per LDS table lookup of the narrow-width decode (see decode_words_narrow); 0 = compare chain
template <int C> constexpr int narrow_k() { return C == 1 ? 8 : (C <= 3 ? 4 : (C <= 5 ? 3 : (C <= 7 ? 2 : 0))); }

template <int C, int VPL, int MODE> constexpr int scan_occ()
{
    constexpr int lds = ScanGeom<C, VPL>::OCC;
    if (MODE != 2 && C <= 7) {
        // table-lookup decode (narrow_k): many lookups in flight, the predicate table (<= 16 KiB) sits next to the
        // tiles, and the launcher runs 1-2 blocks per CU anyway
        const int with_table = (160 * 1024) / (4 * ScanGeom<C, VPL>::LDS_BYTES + (1 << (narrow_k<C>() * C)) + 64);
        return with_table > 4 ? 4 : (with_table < 1 ? 1 : with_table);
    }
    return MODE == 2 ? (lds > 4 ? 4 : lds) : lds;
}

// values per lane per tile used by the shipped dispatch (tools/tune_scan.hip sweeps the alternatives)
constexpr int scan_vpl(int C, int MODE)
{
    if (MODE == kModeShared) return 64;
    return C <= 16 ? 128 : 64;
}

struct ScanArgs {
    const uint8_t *packed;     // 16 B aligned
    uint64_t n;                // values
    uint8_t *out;              // bitmap(s)
    uint64_t out_stride;       // bytes between per-predicate bitmaps (kModeShared, layout 0)
    unsigned long long *hits;  // device counters (one per key), OVERWRITTEN with the totals; may be null
    unsigned long long *scratch; // context scratch: kScratchWords words, all zero between launches
    const int32_t *keys_dev;   // kModeShared with P > kMaxKeysPerPass: device key array (padded to 8)
    uint32_t key[kMaxKeysPerPass]; // kModeEq: key[0]; kModeRange: key[0]=lo, key[1]=hi-lo; kModeShared: P<=8 keys
    uint32_t nkeys;            // P
    uint32_t layout;           // 0 per-predicate, 1 linear
    const uint8_t *and_mask;   // kModeEq / kModeRange: optional bitmap combined with the result (conjunctions ...), may be null
    uint32_t invert;           // kModeEq / kModeRange: 0, or 0xffffffff to negate the predicate (!=, NOT BETWEEN)
    uint32_t mask_op;          // how and_mask is combined: 0 result & mask, 1 result | mask, 2 result ^ mask, 3 mask & ~result
                               // (kModeEq / kModeRange; in_kernel: AND only)
    uint32_t flags;            // A/B switches (option "kernel_flags", DESIGN.md "A/B switches"; they never change a result):
                               // shared scans: 1 shared_wide_kernel drains its stores every tile; 2 per-group kernels instead of
                               // shared_wide2 / shared_linear; 4 rotate the 32-key rounds; 8 per-tile hit-count reductions; 16 P = 16
                               // linear one row per piece; 32 P = 2 on the LUT kernel; 64 compare chain; 128 shared_linear_kernel
                               // whatever the width; 256 round 2's shared_linear_kernel instead of shared_linear2_kernel; 0x200 .. 0x80000: round 3's
                               // switches (0x20000 is set by the launcher: short last table attached).  select_kernel (bits 9, 10 of the option arrive here as 2, 4): timing ablations
    const uint8_t *packed2;    // scan2_kernel: the second column (same width, same n)
    uint32_t key2[2];          // scan2_kernel: second predicate as (lo, hi - lo)
    uint32_t invert2;          // scan2_kernel: negation word of the second predicate
    unsigned long long *tile_state; // select_kernel: one {status, count} word per wave tile (decoupled look-back), zeroed per launch
    uint64_t *rowids;          // select_kernel: ascending row ids out
    uint64_t capacity;         // select_kernel: at most this many ids are written
    uint64_t first_row;        // select_kernel: global row index of row 0
};
// `out` may be null in scan_burst_kernel / scan2_kernel: count-only scan (hits without a bitmap).

// ---- DMA: HBM -> LDS ---------------------------------------------------------------------------
// One wave-instruction moves 64 x 16 B; the LDS destination is wave-uniform base + lane*16, the
// global source is per lane.  AUX carries the cache-policy bits (0 default, 2 = nt).
template <int TILE_BYTES, int AUX>
__device__ __forceinline__ void dma_tile_full(const uint8_t *src, uint8_t *lds_wave, int lane)
{
    constexpr int N = (TILE_BYTES + 1023) / 1024;
#pragma unroll
    for (int j = 0; j < N; j++) {
        if ((j + 1) * 1024 <= TILE_BYTES) {
            __builtin_amdgcn_global_load_lds(MI355_GPTR(src + j * 1024 + lane * 16), MI355_LPTR(lds_wave + j * 1024), 16,
                                             0, AUX);
        } else if (lane * 16 < TILE_BYTES - j * 1024) { // trailing partial instruction
            __builtin_amdgcn_global_load_lds(MI355_GPTR(src + j * 1024 + lane * 16), MI355_LPTR(lds_wave + j * 1024), 16,
                                             0, AUX);
        }
    }
}

// Last (partial) tile: only 16-byte chunks that start inside the payload are fetched.  A chunk may
// run up to 15 bytes past the payload: that is inside the 256-byte pad every packed buffer carries
// (src/simd_scan.hpp:20-26).  Whatever stays stale in LDS only feeds bits >= n, which are masked.
template <int TILE_BYTES, int AUX>
__device__ __forceinline__ void dma_tile_partial(const uint8_t *src, uint64_t bytes_left, uint8_t *lds_wave, int lane)
{
    constexpr int N = (TILE_BYTES + 1023) / 1024;
#pragma unroll
    for (int j = 0; j < N; j++) {
        uint32_t o = j * 1024 + lane * 16;
        if (o < TILE_BYTES && o < bytes_left) {
            __builtin_amdgcn_global_load_lds(MI355_GPTR(src + o), MI355_LPTR(lds_wave + j * 1024), 16, 0, AUX);
        }
    }
}

// ---- value extraction at a compile-time position -------------------------------------------
// LEN bits starting at compile-time bit position BIT of the lane's dwords
template <int BIT, int LEN, int NW> __device__ __forceinline__ uint32_t extract_at(const uint32_t (&w)[NW])
{
    constexpr int d = BIT >> 5;
    constexpr int s = BIT & 31;
    if constexpr (LEN == 32 && s == 0) {
        return w[d];
    } else if constexpr (s + LEN <= 32) {
        return __builtin_amdgcn_ubfe(w[d], s, LEN);
    } else {
        return __builtin_amdgcn_alignbit(w[d + 1], w[d], s) & (LEN == 32 ? 0xffffffffu : ((1u << (LEN & 31)) - 1u));
    }
}

template <int C, int K, int NW> __device__ __forceinline__ uint32_t extract(const uint32_t (&w)[NW])
{
    constexpr int bit = K * C;
    constexpr int d = bit >> 5;
    constexpr int s = bit & 31;
    if constexpr (C == 32) {
        return w[d];
    } else if constexpr (s + C <= 32) {
        return __builtin_amdgcn_ubfe(w[d], s, C); // v_bfe_u32 (folds to v_and / v_lshrrev at the edges)
    } else {
        // straddles a dword boundary: funnel shift, then mask
        return __builtin_amdgcn_alignbit(w[d + 1], w[d], s) & ((1u << C) - 1u);
    }
}

// ---- compare + append one result bit:  acc = 2*acc + predicate(x) ---------------------------------
// v_cmp writes a lane mask, v_addc_co_u32 adds the accumulator to itself (a left shift) with that mask as
// carry-in: two VALU ops per result bit.  A v_cmp -> v_addc pair through ONE mask register is a dependent
// chain (measured ~10 cycles per instruction with 2 waves per SIMD), so the helpers below always run several
// independent chains side by side -- N compares into N different SGPR pairs, then the N add-with-carry --
// which also keeps every VALU-written SGPR at least N instructions away from the VALU that reads it
// (gfx950 wants 2 wait states there; hipcc does not look inside an asm statement).

// one value against 8 keys (shared scan)
__device__ __forceinline__ void push_eq8(uint32_t (&acc)[8], uint32_t x, const uint32_t (&key)[kMaxKeysPerPass])
{
    unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
    asm("v_cmp_eq_u32_e64 %8, %17, %16\n\t"
        "v_cmp_eq_u32_e64 %9, %18, %16\n\t"
        "v_cmp_eq_u32_e64 %10, %19, %16\n\t"
        "v_cmp_eq_u32_e64 %11, %20, %16\n\t"
        "v_cmp_eq_u32_e64 %12, %21, %16\n\t"
        "v_cmp_eq_u32_e64 %13, %22, %16\n\t"
        "v_cmp_eq_u32_e64 %14, %23, %16\n\t"
        "v_cmp_eq_u32_e64 %15, %24, %16\n\t"
        "v_addc_co_u32_e64 %0, %8, %0, %0, %8\n\t"
        "v_addc_co_u32_e64 %1, %9, %1, %1, %9\n\t"
        "v_addc_co_u32_e64 %2, %10, %2, %2, %10\n\t"
        "v_addc_co_u32_e64 %3, %11, %3, %3, %11\n\t"
        "v_addc_co_u32_e64 %4, %12, %4, %4, %12\n\t"
        "v_addc_co_u32_e64 %5, %13, %5, %5, %13\n\t"
        "v_addc_co_u32_e64 %6, %14, %6, %6, %14\n\t"
        "v_addc_co_u32_e64 %7, %15, %7, %7, %15"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
          "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
        : "v"(x), "s"(key[0]), "s"(key[1]), "s"(key[2]), "s"(key[3]), "s"(key[4]), "s"(key[5]), "s"(key[6]), "s"(key[7]));
}

// N values (one per bitmap word of the lane) against one key / one range
template <int MODE> __device__ __forceinline__ void push1(uint32_t &a0, uint32_t x0, uint32_t k0, uint32_t k1)
{
    if constexpr (MODE == 1) {
        uint32_t t;
        asm("v_subrev_u32_e32 %1, %3, %2\n\t"
            "v_cmp_ge_u32_e32 vcc, %4, %1\n\t"
            "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
            : "+v"(a0), "=&v"(t)
            : "v"(x0), "s"(k0), "s"(k1)
            : "vcc");
    } else {
        asm("v_cmp_eq_u32_e32 vcc, %2, %1\n\t"
            "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
            : "+v"(a0)
            : "v"(x0), "s"(k0)
            : "vcc");
    }
}

template <int MODE>
__device__ __forceinline__ void push2(uint32_t &a0, uint32_t &a1, uint32_t x0, uint32_t x1, uint32_t k0, uint32_t k1)
{
    unsigned long long m0, m1;
    if constexpr (MODE == 1) {
        uint32_t t0, t1;
        asm("v_subrev_u32_e32 %4, %8, %6\n\t"
            "v_subrev_u32_e32 %5, %8, %7\n\t"
            "v_cmp_ge_u32_e64 %2, %9, %4\n\t"
            "v_cmp_ge_u32_e64 %3, %9, %5\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"
            "v_addc_co_u32_e64 %1, %3, %1, %1, %3"
            : "+v"(a0), "+v"(a1), "=&s"(m0), "=&s"(m1), "=&v"(t0), "=&v"(t1)
            : "v"(x0), "v"(x1), "s"(k0), "s"(k1));
    } else {
        asm("v_cmp_eq_u32_e64 %2, %6, %4\n\t"
            "v_cmp_eq_u32_e64 %3, %6, %5\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"
            "v_addc_co_u32_e64 %1, %3, %1, %1, %3"
            : "+v"(a0), "+v"(a1), "=&s"(m0), "=&s"(m1)
            : "v"(x0), "v"(x1), "s"(k0));
    }
}

template <int MODE>
__device__ __forceinline__ void push4(uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3, uint32_t x0, uint32_t x1,
                                      uint32_t x2, uint32_t x3, uint32_t k0, uint32_t k1)
{
    unsigned long long m0, m1, m2, m3;
    if constexpr (MODE == 1) {
        uint32_t t0, t1, t2, t3;
        asm("v_subrev_u32_e32 %8, %16, %12\n\t"
            "v_subrev_u32_e32 %9, %16, %13\n\t"
            "v_subrev_u32_e32 %10, %16, %14\n\t"
            "v_subrev_u32_e32 %11, %16, %15\n\t"
            "v_cmp_ge_u32_e64 %4, %17, %8\n\t"
            "v_cmp_ge_u32_e64 %5, %17, %9\n\t"
            "v_cmp_ge_u32_e64 %6, %17, %10\n\t"
            "v_cmp_ge_u32_e64 %7, %17, %11\n\t"
            "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"
            "v_addc_co_u32_e64 %1, %5, %1, %1, %5\n\t"
            "v_addc_co_u32_e64 %2, %6, %2, %2, %6\n\t"
            "v_addc_co_u32_e64 %3, %7, %3, %3, %7"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&v"(t0), "=&v"(t1),
              "=&v"(t2), "=&v"(t3)
            : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(k0), "s"(k1));
    } else {
        asm("v_cmp_eq_u32_e64 %4, %12, %8\n\t"
            "v_cmp_eq_u32_e64 %5, %12, %9\n\t"
            "v_cmp_eq_u32_e64 %6, %12, %10\n\t"
            "v_cmp_eq_u32_e64 %7, %12, %11\n\t"
            "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"
            "v_addc_co_u32_e64 %1, %5, %1, %1, %5\n\t"
            "v_addc_co_u32_e64 %2, %6, %2, %2, %6\n\t"
            "v_addc_co_u32_e64 %3, %7, %3, %3, %7"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
            : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(k0));
    }
}

// ---- decode a lane's run ------------------------------------------------------------------------------
// Values are pushed from the highest index of each 32-group down to the lowest, so that after 32 pushes value
// 32J+0 sits in bit 0 (src/util.cpp:51-58 bit order).

// single predicate: step K handles value 32J+K of every bitmap word J of the lane at once
template <int C, int VPL, int K, int MODE, int NW>
__device__ __forceinline__ void decode_step1(const uint32_t (&w)[NW], uint32_t (&res)[1][VPL / 32], uint32_t k0, uint32_t k1)
{
    if constexpr (VPL == 128) {
        push4<MODE>(res[0][0], res[0][1], res[0][2], res[0][3], extract<C, K, NW>(w), extract<C, 32 + K, NW>(w),
                    extract<C, 64 + K, NW>(w), extract<C, 96 + K, NW>(w), k0, k1);
    } else if constexpr (VPL == 64) {
        push2<MODE>(res[0][0], res[0][1], extract<C, K, NW>(w), extract<C, 32 + K, NW>(w), k0, k1);
    } else {
        push1<MODE>(res[0][0], extract<C, K, NW>(w), k0, k1);
    }
    if constexpr (K > 0) decode_step1<C, VPL, K - 1, MODE, NW>(w, res, k0, k1);
}

// 8 predicates: one value at a time against the 8 keys
template <int C, int J, int K, int NW>
__device__ __forceinline__ void decode_step8(const uint32_t (&w)[NW], uint32_t (&acc)[8], const uint32_t (&key)[kMaxKeysPerPass])
{
    push_eq8(acc, extract<C, 32 * J + K, NW>(w), key);
    if constexpr (K > 0) decode_step8<C, J, K - 1, NW>(w, acc, key);
}

template <int C, int VPL, int J, int NW>
__device__ __forceinline__ void decode_words8(const uint32_t (&w)[NW], uint32_t (&res)[8][VPL / 32],
                                              const uint32_t (&key)[kMaxKeysPerPass])
{
    uint32_t acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = 0;
    decode_step8<C, J, 31, NW>(w, acc, key);
#pragma unroll
    for (int q = 0; q < 8; q++) res[q][J] = acc[q];
    if constexpr (J + 1 < VPL / 32) decode_words8<C, VPL, J + 1, NW>(w, res, key);
}

template <int C, int VPL, int J, int NK, int MODE, int NW>
__device__ __forceinline__ void decode_words(const uint32_t (&w)[NW], uint32_t (&res)[NK][VPL / 32],
                                             const uint32_t (&key)[kMaxKeysPerPass])
{
    if constexpr (NK == 8) {
        decode_words8<C, VPL, 0, NW>(w, res, key);
    } else {
#pragma unroll
        for (int j = 0; j < VPL / 32; j++) res[0][j] = 0;
        decode_step1<C, VPL, 31, MODE, NW>(w, res, key[0], key[1]);
    }
}

// lane-local packed data: LDS -> VGPRs (ds_read_b128 / b64 / b32 by VPL)
template <int C, int VPL>
__device__ __forceinline__ void read_lane_data(const uint8_t *lds_wave, int lane, uint32_t (&w)[VPL * C / 32])
{
    using G = ScanGeom<C, VPL>;
    if constexpr (VPL == 128) {
        const u32x4 *p = (const u32x4 *)(lds_wave + lane * G::LANE_BYTES);
#pragma unroll
        for (int q = 0; q < C; q++) {
            u32x4 v = p[q];
            w[4 * q + 0] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
    } else if constexpr (VPL == 64) {
        const u32x2 *p = (const u32x2 *)(lds_wave + lane * G::LANE_BYTES);
#pragma unroll
        for (int q = 0; q < C; q++) {
            u32x2 v = p[q];
            w[2 * q + 0] = v.x;
            w[2 * q + 1] = v.y;
        }
    } else {
        const uint32_t *p = (const uint32_t *)(lds_wave + lane * G::LANE_BYTES);
#pragma unroll
        for (int q = 0; q < C; q++) w[q] = p[q];
    }
}

// inclusive prefix sum over the 64 lanes with DPP moves (seven VALU operations; a __shfl_up ladder is six dependent
// ds_bpermute round trips through the LDS crossbar): three row_shr of the input give every lane the sum of its group of
// four, row_shr:4 / row_shr:8 (bank-masked) finish the 16-lane rows, row_bcast:15 / row_bcast:31 carry the row totals on.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    uint32_t r = x;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); // row_shr:1
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); // row_shr:2
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false); // row_shr:3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, false); // row_shr:4, banks 1-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, false); // row_shr:8, banks 2-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
    return r;
}

// sum over the 64 lanes, the same value in every lane (wave-uniform): the DPP scan above, then lane 63's total
// (a __shfl_xor butterfly is six dependent ds_bpermute round trips; the multi-pass shared scans reduce four packed
// counters per eight keys and tile)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

// mask for bitmap word J of a lane that owns `valid` (0..VPL) in-range values
__device__ __forceinline__ uint32_t tail_mask(int valid, int J)
{
    int v = valid - 32 * J;
    return v >= 32 ? 0xffffffffu : (v <= 0 ? 0u : ((1u << v) - 1u));
}

// 8 bytes at an arbitrary byte address: one global_store_dwordx2 (gfx950 handles the misalignment in hardware)
struct __attribute__((packed, aligned(1))) Unaligned64 { uint32_t lo, hi; };
__device__ __forceinline__ void store8_unaligned(uint8_t *dst, uint32_t lo, uint32_t hi)
{
    Unaligned64 v;
    v.lo = lo;
    v.hi = hi;
    *(Unaligned64 *)dst = v;
}

// NT: 1 = non-temporal store (the bitmap is written once and not re-read by this kernel); 2 = sc1 write-through
// store (experiment, 16-byte form only)
template <int WORDS, int NT = 0> __device__ __forceinline__ void store_words(uint8_t *dst, const uint32_t (&v)[WORDS])
{
    if constexpr (WORDS == 4) {
        u32x4 t = {v[0], v[1], v[2], v[3]};
        if constexpr (NT == 2) {
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(t) : "memory");
        } else if constexpr (NT == 1) __builtin_nontemporal_store(t, (u32x4 *)dst); else *(u32x4 *)dst = t;
    } else if constexpr (WORDS == 2) {
        u32x2 t = {v[0], v[1]};
        if constexpr (NT == 2) {
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(t) : "memory");
        } else if constexpr (NT == 1) __builtin_nontemporal_store(t, (u32x2 *)dst); else *(u32x2 *)dst = t;
    } else {
        if constexpr (NT) __builtin_nontemporal_store(v[0], (uint32_t *)dst); else *(uint32_t *)dst = v[0];
    }
}

// ---- hit counts without a memset launch ---------------------------------------------------------
// Same-address device atomics serialise at ~12 ns each on MI355X, so 4096 waves x 8 keys adding into 8 words
// cost ~0.4 ms at the tail of a shared scan.  Counts therefore go to kHitSlots replicas of the totals (slot =
// block index mod kHitSlots, rows 8 KiB apart so replicas never share a line): each address sees only
// (#waves / kHitSlots) adds.  Completion is detected per BLOCK: every wave drains its adds (vmcnt counts
// atomics), the block barriers, one lane takes a ticket on the "done" counter.  All of these are device-scope
// atomic RMWs, which gfx950 executes at the memory side (coherent across the 8 XCDs, never held in a CU's L1
// or an XCD's L2), so no cache write-back / invalidate is needed -- a per-wave agent-scope release fence
// (buffer_wbl2) here cost 25 % of the kernel when launches ran back to back.  The block that draws the last
// ticket sums the replicas into the caller's `hits` array with atomic exchanges that also zero the scratch
// for the next launch.  One kernel launch per scan.
constexpr int kHitSlots = 64;
constexpr int kDoneGroups = 16; // first-level "done" counters, one 128-byte line each (see hits_finalize)
constexpr int kScratchWords = kHitSlots * kMaxKeys + 8 + kDoneGroups * 16; // replicas + "done" counter (+ diagnostics) + group counters
constexpr int kScratchDone = kHitSlots * kMaxKeys;
constexpr int kScratchGroupDone = kScratchDone + 8;

__device__ __forceinline__ void hits_add(const ScanArgs &a, uint32_t k, uint32_t wave_total, int lane)
{
    if (lane == 0 && wave_total)
        __hip_atomic_fetch_add(a.scratch + (blockIdx.x % kHitSlots) * kMaxKeys + k, (unsigned long long)wave_total,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every thread of the block must call this (it contains a block barrier)
__device__ __forceinline__ void hits_finalize(const ScanArgs &a, uint32_t P, int lane)
{
    if (!a.hits) return;
    __shared__ uint32_t s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's adds have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        // Two levels of tickets: same-address device atomics serialise at ~12 ns each, so ONE counter taken by every block cost
        // a 512-block launch 6 us (a 2e7-row scan takes 6 us without its count).  Block b takes a ticket of group b mod 16
        // (its own 128-byte line); the block that completes its group takes a ticket of the top counter; the block that
        // completes that one has seen -- transitively -- every block's ticket, i.e. every block's adds performed.
        const uint32_t g = blockIdx.x % kDoneGroups;
        const uint32_t in_group = (gridDim.x - g + kDoneGroups - 1) / kDoneGroups;
        const uint32_t ngroups = gridDim.x < (unsigned)kDoneGroups ? gridDim.x : (unsigned)kDoneGroups;
        unsigned long long *const gdone = a.scratch + kScratchGroupDone + g * 16;
        uint32_t last = 0;
        if (__hip_atomic_fetch_add(gdone, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)in_group - 1) {
            __hip_atomic_store(gdone, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // zero again for the next launch
            last = __hip_atomic_fetch_add(a.scratch + kScratchDone, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)ngroups - 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last) {
        // the last block: thread t sums the replicas of keys t, t + 256, ... (consecutive threads = consecutive words of a
        // replica row; the exchanges of one key are independent, so they overlap -- a single wave walking the keys one
        // after the other waited a memory-side round trip per key: 0.36 ms at P = 512)
        const uint32_t nslots = gridDim.x < (unsigned)kHitSlots ? gridDim.x : (unsigned)kHitSlots;
        if (P <= 16u) {
            // few keys (the single scans: one): a wave per key, a lane per replica -- ONE round of exchanges per key instead
            // of eight dependent ones by a single thread (a 2e7-row scan with its count, launches back to back: 12.5 -> 9.9 us;
            // the same scan without a count: 7.2 us)
            for (uint32_t k = threadIdx.x >> 6; k < P; k += kBlockThreads / 64) {
                const unsigned long long v = (uint32_t)lane < nslots
                                                 ? __hip_atomic_exchange(a.scratch + (uint32_t)lane * kMaxKeys + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : 0ull;
                // 64 lanes x 2^24 stays inside 32 bits: three 24-bit limbs
                const unsigned long long lo = wave_sum((uint32_t)(v & 0xffffffull)), mid = wave_sum((uint32_t)((v >> 24) & 0xffffffull)),
                                         hi = wave_sum((uint32_t)(v >> 48));
                if (lane == 0) a.hits[k] = lo + (mid << 24) + (hi << 48);
            }
            if (threadIdx.x == 0) __hip_atomic_store(a.scratch + kScratchDone, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        for (uint32_t k = threadIdx.x; k < P; k += kBlockThreads) {
            unsigned long long v = 0;
#pragma unroll 8
            for (uint32_t sl = 0; sl < nslots; sl++)
                v += __hip_atomic_exchange(a.scratch + sl * kMaxKeys + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.hits[k] = v;
        }
        if (threadIdx.x == 0) __hip_atomic_store(a.scratch + kScratchDone, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Multi-pass shared scans (P > 8) count hits per tile and pass.  Doing that with a wave reduction and a global
// atomic per key stalled the pipeline (the atomics sit in front of the next tile's vmcnt wait).  Instead: the
// lane's 8 counts (<= 64 each) are packed four to a dword in 16-bit fields, two wave reductions sum them
// (<= 4096 per field), lane 0 adds the 8 sums to per-block counters in LDS, and the block flushes those to the
// replicated global totals once, at the end.
__device__ __forceinline__ void block_hits_add8(uint32_t *s_hits, uint32_t kbase, uint32_t P, const uint32_t (&cnt)[8], int lane)
{
    uint32_t p0 = cnt[0] | (cnt[1] << 16), p1 = cnt[2] | (cnt[3] << 16), p2 = cnt[4] | (cnt[5] << 16), p3 = cnt[6] | (cnt[7] << 16);
    p0 = wave_sum(p0);
    p1 = wave_sum(p1);
    p2 = wave_sum(p2);
    p3 = wave_sum(p3);
    if (lane == 0) {
        const uint32_t v[8] = {p0 & 0xffff, p0 >> 16, p1 & 0xffff, p1 >> 16, p2 & 0xffff, p2 >> 16, p3 & 0xffff, p3 >> 16};
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (kbase + q < P && v[q]) atomicAdd(&s_hits[kbase + q], v[q]);
    }
}

// every thread of the block calls this once, after its last block_hits_add8 and before hits_finalize
__device__ __forceinline__ void block_hits_flush(const ScanArgs &a, uint32_t *s_hits, uint32_t P)
{
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) {
        const uint32_t v = s_hits[k];
        if (v)
            __hip_atomic_fetch_add(a.scratch + (blockIdx.x % kHitSlots) * kMaxKeys + k, (unsigned long long)v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- narrow widths: several values per LDS lookup -------------------------------------------------
// Extract + v_cmp + v_addc costs 12-17 SIMD-cycles per value (all three are half-rate ops on gfx950), more than the
// HBM stream leaves per value below ~8 bits (1e9 x 5 bit: 17 cycles per wave-value at 6 TB/s).  For C <= 7 the
// eq / range scans therefore evaluate LK values per step through a table in LDS: index = the LK*C packed bits of LK
// consecutive values (one v_bfe_u32 / v_alignbit_b32), entry = their LK predicate bits, appended to the bitmap word
// with one v_lshl_or_b32 -- ~9 cycles per LK values.  LK = 8 / 4 / 4 / 3 / 3 / 2 / 2 for C = 1..7 (tables of 256 B to
// 32 KiB; at C = 5 the lookups themselves are the next limit: 3 values per lookup instead of 2 took the scan from 90 %
// to 95 % of a trivial kernel with the same byte mix, tools/ceilings.hip).  The
// block builds the 2^(LK*C)-entry table from the predicate while its first tile's DMA is in flight.
// one lookup: values [32J + LK*GI, +LEN) of the lane, LEN = LK except for the last group of a word when LK does not
// divide 32 (its entry's upper bits describe fields that are not there: masked)
template <int C, int LK, int J, int GI, int NW>
__device__ __forceinline__ void narrow_step(const uint32_t (&w)[NW], uint32_t &acc, const uint8_t *table)
{
    constexpr int FIRST = LK * GI;
    constexpr int LEN = (32 - FIRST) < LK ? (32 - FIRST) : LK;
    uint32_t m = table[extract_at<(32 * J + FIRST) * C, LEN * C, NW>(w)];
    if constexpr (LEN < LK) m &= (1u << LEN) - 1u;
    acc = (acc << LEN) | m;
    if constexpr (GI > 0) narrow_step<C, LK, J, GI - 1, NW>(w, acc, table);
}

template <int C, int VPL, int LK, int J, int NW>
__device__ __forceinline__ void decode_words_narrow(const uint32_t (&w)[NW], uint32_t (&res)[1][VPL / 32], const uint8_t *table)
{
    uint32_t acc = 0;
    narrow_step<C, LK, J, (32 + LK - 1) / LK - 1, NW>(w, acc, table); // from the word's last group down to its first
    res[0][J] = acc;
    if constexpr (J + 1 < VPL / 32) decode_words_narrow<C, VPL, LK, J + 1, NW>(w, res, table);
}

} // namespace mi355

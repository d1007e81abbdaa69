// kernels.hpp -- CDNA4 (gfx950) kernels of the bit-packed column engine.  gfx950 only.
//
// Data flow of every kernel that reads a packed column (DESIGN.md "kernels"):
//
//   HBM --global_load_lds_dwordx4 (1 KiB / wave-instruction, coalesced)--> per-wave LDS tile
//       --ds_read_b128/b64 (each lane fetches the 16C / 8C bytes that hold ITS OWN run of
//         128 / 64 consecutive values: the LDS does the gather a CPU does with pshufb)-->
//   VGPRs --v_bfe_u32 / v_alignbit_b32 at COMPILE-TIME bit offsets (the run starts on a dword
//         boundary, so every shift is a constant)--> value
//       --v_cmp + v_addc_co_u32 (acc = 2*acc + match: one VALU op appends a result bit)-->
//   one 32-bit bitmap word per 32 values in the lane, written back as 16 B / lane (1 KiB / wave).
//
// Kernels in this file:
//   scan_kernel            equality / range scan (+ negation, + AND with an earlier bitmap), one bitmap
//   shared_lut_kernel      shared multi-predicate scan through an LDS lookup table + 8x8 bit transposes
//   shared_general_kernel  shared scan by compare chain, for key counts whose tables do not fit in LDS
//   in_kernel              IN-list scan (one bitmap for a key set)
//   decompress_kernel      packed -> int32, lane per value
//   pack_kernel            packer and synthetic column generators
//   bitmap_kernel, rowid_* bitmap combine / count, selection vector
// ABL / DEPTH template knobs of scan_kernel exist for tools/tune_scan.hip (ablations, diagnostics, experiments);
// the shipped dispatch (width_group.hip) always uses ABL = 0, DEPTH = 1.
//
// What this replaces in the reference (RRr89/Shared_SIMD_Scan): the pshufb byte-gather + pmulld /
// psrld shift + pcmpeqd + movmskps chains of src/simd_scan.cpp:103-306, src/simd_scan_shared.cpp:34-151,
// src/simd_scan_shared_linear.cpp:9-62 and src/simd_scan_decompression.cpp:237-470.  No MFMA: this is
// integer/bit work bound by HBM bandwidth.
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
    const uint8_t *and_mask;   // kModeEq / kModeRange: optional bitmap ANDed into the result (conjunctions), may be null
    uint32_t invert;           // kModeEq / kModeRange: 0, or 0xffffffff to negate the predicate (!=, NOT BETWEEN)
};

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

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
        if constexpr (NT) __builtin_nontemporal_store(t, (u32x2 *)dst); else *(u32x2 *)dst = t;
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
constexpr int kScratchWords = kHitSlots * kMaxKeys + 8; // replicas + "done" counter (+ diagnostics)
constexpr int kScratchDone = kHitSlots * kMaxKeys;

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
    __shared__ unsigned long long s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's adds have been performed
    __syncthreads();
    if (threadIdx.x == 0)
        s_ticket = __hip_atomic_fetch_add(a.scratch + kScratchDone, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket == (unsigned long long)gridDim.x - 1 && threadIdx.x < 64) {
        const uint32_t nslots = gridDim.x < (unsigned)kHitSlots ? gridDim.x : (unsigned)kHitSlots;
        for (uint32_t k = 0; k < P; k++) {
            unsigned long long v = 0;
            if ((uint32_t)lane < nslots)
                v = __hip_atomic_exchange(a.scratch + lane * kMaxKeys + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) a.hits[k] = v;
        }
        if (lane == 0) __hip_atomic_store(a.scratch + kScratchDone, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ---- the scan kernel ------------------------------------------------------------------------
// Per wave, per tile:  wait for the tile's DMA -> ds_read the lane's run into VGPRs -> (LDS is free)
// store the PREVIOUS tile's bitmap words, then issue the NEXT tile's DMA -> decode/compare in registers.
// Stores are issued before the DMA that the next iteration waits for, so a plain vmcnt(0) never waits
// on a store younger than the data it needs, whatever the number of stores per tile is; the DMA of
// tile t+1 is in flight during the whole compute phase of tile t.
//
// MODE kModeEq / kModeRange: one bitmap.  MODE kModeShared: up to 8 keys, one bitmap per key at
// out + k*out_stride (one decode, 8 compares per value); the column is read from HBM once.  Larger P and
// the linear layout go through shared_general_kernel below.
// AUX_: bits 0-3 = cache policy of the DMA loads (0 default, 2 nt); bit 4 = non-temporal bitmap stores.
// ABL (ablation / diagnostics, tools/tune_scan.hip only): 1 = DMA only, 2 = DMA + LDS reads, 3 = no bitmap
// stores, 4 = normal + clock / placement stamps.
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
    __device__ __forceinline__ uint32_t finish_tail(uint64_t t, uint32_t (&v)[VPL / 32], uint8_t *dst, uint64_t byte_stride, int lane) const
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
                if (4 * j + b < nbytes) dst[(uint64_t)(4 * j + b) * byte_stride] = (uint8_t)(v[j] >> (8 * b));
        }
        return cnt;
    }
};

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
        if (prev != ~0ull) { // every tile but a wave's last is a full tile
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
                if (tile < tc.nfull) {
#pragma unroll
                    for (int j = 0; j < WORDS; j++) res[0][j] &= ((const uint32_t *)mp)[j];
                } else { // tail tile: read only the bytes the mask is guaranteed to hold (ceil(n/8))
                    const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
                    const int nbytes = left <= 0 ? 0 : (int)((left >= VPL ? VPL : left) + 7) / 8;
#pragma unroll
                    for (int j = 0; j < WORDS; j++) {
                        uint32_t m = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            if (4 * j + b < nbytes) m |= (uint32_t)mp[4 * j + b] << (8 * b);
                        res[0][j] &= m;
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
                if ((uint32_t)q < P) hits[q] += tc.finish_tail(tile, res[q], dst, 1, lane);
                dst += kstride;
            }
            prev = ~0ull;
        }
        tile = next;
    }
    if (prev != ~0ull) {
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
// Keys outside [0, 2^C) get no bit anywhere (they match nothing, as in the reference).  P <= 64 (8 passes).
constexpr int kLutMaxPasses = kMaxKeys / 8; // as many as fit in LDS beside the tiles (checked by the launcher)

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

// LAYOUT 0: per-predicate bitmaps at out + k*out_stride; 1: linear (byte of 8-value group g and key k at
// g*P + k, src/simd_scan_shared_linear.cpp:57).  MULTI false: P <= 8, one pass, stores deferred by one tile
// (as in scan_kernel); true: ceil(P/8) passes per tile, stored pass by pass.
template <int C, int AUX_, int VPL, int LAYOUT, bool MULTI>
__global__ __launch_bounds__(kBlockThreads) void shared_lut_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    using L = LutGeom<C, MULTI>;
    constexpr int WORDS = G::WORDS;
    constexpr int GROUPS = VPL / 8;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 16) ? 1 : 0; // non-temporal result stores (outputs larger than the Infinity Cache)
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
            if (!MULTI && P == 8) {
                // the lane's GROUPS x 8 keys are 8*GROUPS contiguous bytes, the wave's tile 64 x that: transpose
                // through a per-wave LDS stage so every store instruction writes 1 KiB contiguous instead of
                // 64 x 16 B at a 64-B stride
                u32x4 *st = (u32x4 *)stage[wave];
#pragma unroll
                for (int g = 0; g < GROUPS; g += 2) {
                    u32x4 v = {res[2 * g], res[2 * g + 1], res[2 * g + 2], res[2 * g + 3]};
                    st[lane * (GROUPS / 2) + g / 2] = v;
                }
                u32x4 *dst = (u32x4 *)(a.out + (t * G::BITMAP_BYTES) * 8);
#pragma unroll
                for (int j = 0; j < GROUPS / 2; j++) {
                    if constexpr (NTS)
                        __builtin_nontemporal_store(st[j * 64 + lane], dst + j * 64 + lane);
                    else
                        dst[j * 64 + lane] = st[j * 64 + lane];
                }
            } else if (!MULTI && P == 4) {
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

    uint32_t resp[NRES]; // !MULTI: results of the previous tile, not yet stored
    uint64_t prev = ~0ull;

    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!MULTI) {
            if (prev != ~0ull) store_full(prev, 0, resp);
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
                lut_groups_x<C, VPL, false, MULTI>(xs, table, VPL, Y);
                if (LAYOUT == 0 || a.hits) lut_gather_keys<VPL>(Y, out);
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
#pragma unroll
                    for (int i = 0; i < NRES; i++) resp[i] = res[i];
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
                lut_groups_x<C, VPL, true, MULTI>(xs, table, valid, Y);
                lut_gather_keys<VPL>(Y, out);
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
        if (prev != ~0ull) store_full(prev, 0, resp);
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
template <int C> struct WideLutGeom {
    static constexpr int ND = C <= 10 ? 1 : (C + 7) / 8;
    static constexpr bool SINGLE = ND == 1;
    static constexpr int DIGIT_BITS = SINGLE ? C : 8;
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

template <int C, int AUX_, int VPL, int LAYOUT>
__global__ __launch_bounds__(kBlockThreads) void shared_wide_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    using L = WideLutGeom<C>;
    constexpr int WORDS = G::WORDS;
    constexpr int GROUPS = VPL / 8;
    constexpr int AUX = AUX_ & 15;
    constexpr int NTS = (AUX_ & 16) ? 1 : 0; // non-temporal result stores
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

    while (tile < tc.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[G::LANE_DWORDS];
        read_lane_data<C, VPL>(lds_wave, lane, w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);
        const bool full = tile < tc.nfull;
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

// ---- IN-list scan: bitmap[i] = (value_i in {keys}) ----------------------------------------------------
// One result bitmap for a set of keys (the OR-reduction of a shared scan).
//   C <= 16: the set is a 2^C-bit bitset in LDS (<= 8 KiB) built by the block; one byte lookup + bit extract per value.
//   C  > 16: compare chain over the key list (device array, padded to 8): O(P) half-rate compares per value.
// Same tile / DMA / deferred-store skeleton as scan_kernel (VPL from scan_vpl(C, kModeEq)); and_mask / invert apply.
template <int C, int AUX_, int VPL>
__global__ __launch_bounds__(kBlockThreads) void in_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr bool BITSET = C <= 16;
    constexpr int SET_BYTES = BITSET ? ((1 << (C < 16 ? C : 16)) + 7) / 8 : 16;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t set_words[(SET_BYTES + 3) / 4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t P = a.nkeys;

    if (tile < tc.ntiles) tc.template issue<AUX>(a.packed, tile, lds_wave, lane);
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (prev != ~0ull) store_words<WORDS>(out_lane + prev * G::BITMAP_BYTES, res);
        const uint64_t next = tile + stride;
        if (next < tc.ntiles) tc.template issue<AUX>(a.packed, next, lds_wave, lane);

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
                const uint32_t *mp = (const uint32_t *)(a.and_mask + tile * G::BITMAP_BYTES + lane * (WORDS * 4));
#pragma unroll
                for (int j = 0; j < WORDS; j++) res[j] &= mp[j];
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
    if (prev != ~0ull) store_words<WORDS>(out_lane + prev * G::BITMAP_BYTES, res);
    if (a.hits) hits_add(a, 0, wave_sum(hits), lane);
    hits_finalize(a, 1, lane);
}

// ---- bitmap consumers (the step after the path) ----------------------------------------
// combine: out = a OP b over ceil(n/8) bytes (16 B per lane), popcount of the result in the same pass.
// The bitmaps are canonical (bits >= n are zero), so AND / OR / XOR / ANDNOT keep them canonical.
enum BitmapOp { kBitAnd = 0, kBitOr = 1, kBitXor = 2, kBitAndNot = 3, kBitCount = 4 };

struct BitmapArgs {
    const uint8_t *a, *b;
    uint8_t *out;
    uint64_t nbytes;             // ceil(n/8)
    unsigned long long *count;   // device counter, pre-zeroed, may be null
};

template <int OP> __global__ __launch_bounds__(256) void bitmap_kernel(BitmapArgs g)
{
    const uint64_t nvec = g.nbytes / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t cnt = 0;
    auto combine = [](uint32_t x, uint32_t y) -> uint32_t {
        return OP == kBitAnd ? (x & y) : OP == kBitOr ? (x | y) : OP == kBitXor ? (x ^ y) : OP == kBitAndNot ? (x & ~y) : x;
    };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        u32x4 x = ((const u32x4 *)g.a)[i];
        u32x4 r;
        if constexpr (OP != kBitCount) {
            u32x4 y = ((const u32x4 *)g.b)[i];
            r = u32x4{combine(x.x, y.x), combine(x.y, y.y), combine(x.z, y.z), combine(x.w, y.w)};
            ((u32x4 *)g.out)[i] = r;
        } else {
            r = x;
        }
        cnt += __builtin_popcount(r.x) + __builtin_popcount(r.y) + __builtin_popcount(r.z) + __builtin_popcount(r.w);
    }
    // the < 16 trailing bytes
    if (blockIdx.x == 0 && threadIdx.x < (g.nbytes & 15)) {
        const uint64_t i = nvec * 16 + threadIdx.x;
        uint32_t r = g.a[i];
        if constexpr (OP != kBitCount) {
            r = combine(r, g.b[i]) & 0xffu;
            g.out[i] = (uint8_t)r;
        }
        cnt += __builtin_popcount(r);
    }
    if (g.count) {
        cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(g.count + (blockIdx.x % kHitSlots), (unsigned long long)cnt);
    }
}

// moves the kHitSlots partial counts of bitmap_kernel into *out and re-zeroes them
static __global__ __launch_bounds__(64) void sum_slots_kernel(unsigned long long *slots, unsigned long long *out)
{
    unsigned long long v = slots[threadIdx.x];
    slots[threadIdx.x] = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (threadIdx.x == 0) *out = v;
}

// selection vector: row ids (first_row + i for every set bit i) in ascending order.
// Pass 1 (rowid_count_kernel): popcount per chunk of kRowidChunk bytes.  Pass 2: exclusive scan of the chunk counts
// (one block; the chunk array is small).  Pass 3 (rowid_write_kernel): each wave expands its chunk.
constexpr int kRowidChunk = 2048; // bytes of bitmap per wave = 16384 rows

struct RowidArgs {
    const uint8_t *bitmap;
    uint64_t nbytes;
    uint64_t first_row;
    unsigned long long *chunk_counts; // nchunks + 1 entries (exclusive scan in place; [nchunks] = total)
    uint64_t nchunks;
    uint64_t *rowids;
    uint64_t capacity;
};

static __global__ __launch_bounds__(256) void rowid_count_kernel(RowidArgs g)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t ch = wave; ch < g.nchunks; ch += nwaves) {
        const uint64_t base = ch * kRowidChunk;
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < kRowidChunk / (64 * 4); k++) {
            const uint64_t o = base + (uint64_t)(k * 64 + lane) * 4;
            uint32_t w = 0;
            if (o + 4 <= g.nbytes)
                w = *(const uint32_t *)(g.bitmap + o);
            else
                for (int b = 0; b < 4; b++)
                    if (o + b < g.nbytes) w |= (uint32_t)g.bitmap[o + b] << (8 * b);
            cnt += __builtin_popcount(w);
        }
        cnt = wave_sum(cnt);
        if (lane == 0) g.chunk_counts[ch] = cnt;
    }
}

static __global__ __launch_bounds__(1024) void rowid_scan_kernel(RowidArgs g)
{
    // single block exclusive scan over nchunks counts (nchunks = n / 16384: 61k for 1e9 rows)
    __shared__ unsigned long long part[1024];
    const uint64_t per = (g.nchunks + 1023) / 1024;
    const uint64_t lo = threadIdx.x * per, hi = lo + per < g.nchunks ? lo + per : g.nchunks;
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; i++) s += g.chunk_counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) {
            unsigned long long t = part[i];
            part[i] = run;
            run += t;
        }
        g.chunk_counts[g.nchunks] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint64_t i = lo; i < hi; i++) {
        unsigned long long t = g.chunk_counts[i];
        g.chunk_counts[i] = run;
        run += t;
    }
}

// Pass 3.  A step covers 64 lanes x 32 bits = 2048 rows.  Each lane expands its word into a wave-private LDS buffer
// (16-bit offsets inside the step, at the position given by the wave prefix of the popcounts), then the wave copies
// the buffer out with consecutive lanes writing consecutive ids (512 B per store instruction).  Expanding straight
// into global memory made every store instruction touch up to 64 lines: 2.7 ms for 5e8 ids against 4 GB / 6 TB/s.
static __global__ __launch_bounds__(256) void rowid_write_kernel(RowidArgs g)
{
    __shared__ uint16_t stage[4][2048];
    const int lane = threadIdx.x & 63;
    uint16_t *const st = stage[threadIdx.x >> 6];
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t ch = wave; ch < g.nchunks; ch += nwaves) {
        const uint64_t base = ch * kRowidChunk;
        uint64_t out = g.chunk_counts[ch];
#pragma unroll 1
        for (int k = 0; k < kRowidChunk / (64 * 4); k++) {
            const uint64_t o = base + (uint64_t)(k * 64 + lane) * 4;
            uint32_t w = 0;
            if (o + 4 <= g.nbytes)
                w = *(const uint32_t *)(g.bitmap + o);
            else
                for (int b = 0; b < 4; b++)
                    if (o + b < g.nbytes) w |= (uint32_t)g.bitmap[o + b] << (8 * b);
            // exclusive prefix of the lanes' popcounts inside the wave
            const uint32_t c = __builtin_popcount(w);
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            const uint32_t total = __shfl(incl, 63, 64);
            uint32_t pos = incl - c;
            const uint32_t off0 = lane * 32; // row offset of the lane's bit 0 inside the step
            while (w) {
                const int bit = __builtin_ctz(w);
                w &= w - 1;
                st[pos++] = (uint16_t)(off0 + bit);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the wave's LDS writes are done (LDS is in order per wave)
            const uint64_t row0 = g.first_row + (base + (uint64_t)k * 256) * 8;
            for (uint32_t i = lane; i < total; i += 64) {
                const uint64_t p = out + i;
                if (p < g.capacity) g.rowids[p] = row0 + st[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the next step overwrites the buffer
            out += total;
        }
    }
}

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
    constexpr bool NTS = (AUX_ & 16) != 0;  // non-temporal stores of the int32 output
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
                if constexpr (NTS) __builtin_nontemporal_store(v, (u32x4 *)(dst + s * 256)); else *(u32x4 *)(dst + s * 256) = v;
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

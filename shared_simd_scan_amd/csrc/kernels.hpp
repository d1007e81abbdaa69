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
constexpr int kScratchDone = kMaxKeys; // index of the "waves finished" counter in the scratch array

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
    static constexpr int OCC_LDS = (160 * 1024) / (4 * LDS_BYTES);
    static constexpr int OCC = OCC_LDS >= 8 ? 8 : (OCC_LDS < 1 ? 1 : OCC_LDS);
};

enum ScanMode { kModeEq = 0, kModeRange = 1, kModeShared = 2 };

// occupancy target handed to the register allocator: what LDS admits, but the 8-key shared scan keeps
// 8 accumulators + 8x(VPL/32) result words + hit counters live and wants up to 128 VGPRs
template <int C, int VPL, int MODE> constexpr int scan_occ()
{
    constexpr int lds = ScanGeom<C, VPL>::OCC;
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
    unsigned long long *scratch; // context scratch: kScratchDone+1 words, all zero between launches
    const int32_t *keys_dev;   // kModeShared with P > kMaxKeysPerPass: device key array (padded to 8)
    uint32_t key[kMaxKeysPerPass]; // kModeEq: key[0]; kModeRange: key[0]=lo, key[1]=hi-lo; kModeShared: P<=8 keys
    uint32_t nkeys;            // P
    uint32_t layout;           // 0 per-predicate, 1 linear
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

// acc = 2*acc + (x == key).  v_cmp writes VCC, v_addc_co_u32 shifts the accumulator left by adding
// it to itself and takes the compare bit as carry-in: one VALU op per result bit.
__device__ __forceinline__ void push_eq(uint32_t &acc, uint32_t x, uint32_t key)
{
    asm("v_cmp_eq_u32_e32 vcc, %2, %1\n\t"
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
        : "+v"(acc)
        : "v"(x), "s"(key)
        : "vcc");
}

// the same for 8 keys against one value, as ONE asm statement (between separate asm statements that
// clobber VCC hipcc inserts an s_nop per pair)
__device__ __forceinline__ void push_eq8(uint32_t (&acc)[8], uint32_t x, const uint32_t (&key)[kMaxKeysPerPass])
{
    asm("v_cmp_eq_u32_e32 vcc, %9, %8\n\t"
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %10, %8\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %11, %8\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %12, %8\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %13, %8\n\t"
        "v_addc_co_u32_e32 %4, vcc, %4, %4, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %14, %8\n\t"
        "v_addc_co_u32_e32 %5, vcc, %5, %5, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %15, %8\n\t"
        "v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, %16, %8\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %7, vcc"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(x), "s"(key[0]), "s"(key[1]), "s"(key[2]), "s"(key[3]), "s"(key[4]), "s"(key[5]), "s"(key[6]), "s"(key[7])
        : "vcc");
}

// acc = 2*acc + (x - lo <= span)   (unsigned: lo <= x <= lo+span)
__device__ __forceinline__ void push_range(uint32_t &acc, uint32_t x, uint32_t lo, uint32_t span)
{
    uint32_t t;
    asm("v_subrev_u32_e32 %1, %3, %2\n\t"
        "v_cmp_ge_u32_e32 vcc, %4, %1\n\t"
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
        : "+v"(acc), "=&v"(t)
        : "v"(x), "s"(lo), "s"(span)
        : "vcc");
}

template <int C, int J, int K, int NK, int MODE, int NW>
__device__ __forceinline__ void decode_step(const uint32_t (&w)[NW], uint32_t (&acc)[NK], const uint32_t (&key)[kMaxKeysPerPass])
{
    // values are pushed from the highest index of the 32-group down to the lowest, so that after
    // 32 pushes value 32J+0 sits in bit 0 (src/util.cpp:51-58 bit order)
    uint32_t x = extract<C, 32 * J + K, NW>(w);
    if constexpr (MODE == kModeRange) {
        push_range(acc[0], x, key[0], key[1]);
    } else if constexpr (NK == 8) {
        push_eq8(acc, x, key);
    } else {
#pragma unroll
        for (int q = 0; q < NK; q++) push_eq(acc[q], x, key[q]);
    }
    if constexpr (K > 0) decode_step<C, J, K - 1, NK, MODE, NW>(w, acc, key);
}

// bitmap word J (values 32J..32J+31 of the lane) for each of NK predicates
template <int C, int J, int NK, int MODE, int NW>
__device__ __forceinline__ void decode_word(const uint32_t (&w)[NW], uint32_t (&acc)[NK], const uint32_t (&key)[kMaxKeysPerPass])
{
#pragma unroll
    for (int q = 0; q < NK; q++) acc[q] = 0;
    decode_step<C, J, 31, NK, MODE, NW>(w, acc, key);
}

template <int C, int VPL, int J, int NK, int MODE, int NW>
__device__ __forceinline__ void decode_words(const uint32_t (&w)[NW], uint32_t (&res)[NK][VPL / 32],
                                             const uint32_t (&key)[kMaxKeysPerPass])
{
    uint32_t acc[NK];
    decode_word<C, J, NK, MODE, NW>(w, acc, key);
#pragma unroll
    for (int q = 0; q < NK; q++) res[q][J] = acc[q];
    if constexpr (J + 1 < VPL / 32) decode_words<C, VPL, J + 1, NK, MODE, NW>(w, res, key);
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

// NT: non-temporal store (the bitmap is written once and not re-read by this kernel)
template <int WORDS, bool NT = false> __device__ __forceinline__ void store_words(uint8_t *dst, const uint32_t (&v)[WORDS])
{
    if constexpr (WORDS == 4) {
        u32x4 t = {v[0], v[1], v[2], v[3]};
        if constexpr (NT) __builtin_nontemporal_store(t, (u32x4 *)dst); else *(u32x4 *)dst = t;
    } else if constexpr (WORDS == 2) {
        u32x2 t = {v[0], v[1]};
        if constexpr (NT) __builtin_nontemporal_store(t, (u32x2 *)dst); else *(u32x2 *)dst = t;
    } else {
        if constexpr (NT) __builtin_nontemporal_store(v[0], (uint32_t *)dst); else *(uint32_t *)dst = v[0];
    }
}

// ---- hit counts without a memset launch ---------------------------------------------------------
// Every wave adds its per-key counts to the context's scratch totals, waits until those adds have been
// performed, then takes a ticket on the "done" counter.  All of these are device-scope atomic RMWs, which
// gfx950 executes at the memory side (coherent across the 8 XCDs, never held in a CU's L1 or an XCD's L2),
// so no cache write-back / invalidate is needed -- a per-wave agent-scope release fence (buffer_wbl2) here
// cost 25 % of the kernel when launches run back to back.  The wave that draws the last ticket therefore
// sees every other wave's adds: it moves the totals to the caller's `hits` array with atomic exchanges that
// also zero the scratch for the next launch.  One kernel launch per scan.
__device__ __forceinline__ void hits_add(const ScanArgs &a, uint32_t k, uint32_t wave_total, int lane)
{
    if (lane == 0 && wave_total)
        __hip_atomic_fetch_add(a.scratch + k, (unsigned long long)wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void hits_finalize(const ScanArgs &a, uint32_t P, int lane)
{
    if (!a.hits) return;
    unsigned long long ticket = 0;
    // this wave's adds are complete (vmcnt counts atomics) before its ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
        ticket = __hip_atomic_fetch_add(a.scratch + kScratchDone, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0, 64);
    const unsigned long long nwaves = (unsigned long long)gridDim.x * kWavesPerBlock;
    if (ticket == nwaves - 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t k = lane; k < P; k += 64) {
            a.hits[k] = __hip_atomic_exchange(a.scratch + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) __hip_atomic_store(a.scratch + kScratchDone, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the scan kernel ------------------------------------------------------------------------
// Per wave, per tile:  wait for the tile's DMA -> ds_read the lane's run into VGPRs -> (LDS is free)
// store the PREVIOUS tile's bitmap words, then issue the NEXT tile's DMA -> decode/compare in registers.
// Stores are issued before the DMA that the next iteration waits for, so a plain vmcnt(0) never waits
// on a store younger than the data it needs, whatever the number of stores per tile is; the DMA of
// tile t+1 is in flight during the whole compute phase of tile t.
//
// MODE kModeEq / kModeRange: one bitmap.  MODE kModeShared: 8 keys per pass over the lane's registers
// (one decode, 8 compares per value), ceil(P/8) passes per tile; the column is read from HBM once.
// ABL (ablation, tools/tune_scan.hip only): 1 = DMA only, 2 = DMA + LDS reads, 3 = no bitmap stores.
// AUX: bits 0-3 = cache policy of the DMA loads (0 default, 2 nt); bit 4 = non-temporal bitmap stores.
template <int C, int MODE, int AUX_, int VPL, int ABL = 0>
__global__ __launch_bounds__(kBlockThreads, (scan_occ<C, VPL, MODE>())) void scan_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    constexpr int NK = (MODE == kModeShared) ? kMaxKeysPerPass : 1;
    constexpr int WORDS = G::WORDS;
    constexpr int AUX = AUX_ & 15;
    constexpr bool NTS = (AUX_ & 16) != 0;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];

    const uint64_t n = a.n;
    const uint64_t ntiles = (n + G::TILE_VALUES - 1) / G::TILE_VALUES;
    const uint64_t nfull = n / G::TILE_VALUES;
    const uint64_t data_bytes = (n * C + 7) / 8;
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t tile = (uint64_t)blockIdx.x * kWavesPerBlock + wave;

    auto issue = [&](uint64_t t) {
        const uint8_t *src = a.packed + t * G::TILE_BYTES;
        if (t < nfull)
            dma_tile_full<G::TILE_BYTES, AUX>(src, lds_wave, lane);
        else
            dma_tile_partial<G::TILE_BYTES, AUX>(src, data_bytes - t * G::TILE_BYTES, lds_wave, lane);
    };
    // tail tile: zero bits >= n, write exactly ceil(n/8) bytes of the tile's bitmap
    auto finish_tail = [&](uint64_t t, uint32_t (&v)[WORDS], uint8_t *dst, uint64_t byte_stride) -> uint32_t {
        const int64_t left = (int64_t)(n - t * G::TILE_VALUES) - (int64_t)lane * VPL;
        const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
        const int nbytes = (valid + 7) / 8;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < WORDS; j++) {
            v[j] &= tail_mask(valid, j);
            cnt += __builtin_popcount(v[j]);
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4 * j + b < nbytes) dst[(uint64_t)(4 * j + b) * byte_stride] = (uint8_t)(v[j] >> (8 * b));
        }
        return cnt;
    };

    const uint32_t P = (MODE == kModeShared) ? a.nkeys : 1;
    const bool one_pass = P <= (uint32_t)kMaxKeysPerPass;

    if (MODE != kModeShared || (one_pass && a.layout == 0)) {
        // ---------------- pipelined loop: one pass of NK keys, per-predicate bitmaps ----------------
        uint32_t key[kMaxKeysPerPass];
#pragma unroll
        for (int q = 0; q < kMaxKeysPerPass; q++) key[q] = a.key[q];
        uint32_t hits[NK];
#pragma unroll
        for (int q = 0; q < NK; q++) hits[q] = 0;

        uint32_t res[NK][WORDS];
        uint64_t prev = ~0ull; // tile whose results sit in `res`, not yet stored
        if (tile < ntiles) issue(tile);
        while (tile < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t w[G::LANE_DWORDS];
            if constexpr (ABL != 1) read_lane_data<C, VPL>(lds_wave, lane, w);
            // the LDS tile must be fully read before the next DMA may overwrite it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (prev != ~0ull) { // every tile but a wave's last is a full tile
                if constexpr (ABL == 0) {
#pragma unroll
                    for (int q = 0; q < NK; q++)
                        if ((uint32_t)q < P)
                            store_words<WORDS, NTS>(a.out + (uint64_t)q * a.out_stride + prev * G::BITMAP_BYTES + lane * (WORDS * 4), res[q]);
                }
            }
            const uint64_t next = tile + stride;
            if (next < ntiles) issue(next);

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
            } else {
                decode_words<C, VPL, 0, NK, MODE, G::LANE_DWORDS>(w, res, key);
            }
            if (tile < nfull) {
#pragma unroll
                for (int q = 0; q < NK; q++)
#pragma unroll
                    for (int j = 0; j < WORDS; j++) hits[q] += __builtin_popcount(res[q][j]);
                prev = tile;
            } else {
#pragma unroll
                for (int q = 0; q < NK; q++)
                    if ((uint32_t)q < P)
                        hits[q] += finish_tail(tile, res[q], a.out + (uint64_t)q * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4), 1);
                prev = ~0ull;
            }
            tile = next;
        }
        if (prev != ~0ull) {
            if constexpr (ABL == 0) {
#pragma unroll
                for (int q = 0; q < NK; q++)
                    if ((uint32_t)q < P)
                        store_words<WORDS, NTS>(a.out + (uint64_t)q * a.out_stride + prev * G::BITMAP_BYTES + lane * (WORDS * 4), res[q]);
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
    } else if constexpr (MODE == kModeShared) {
        // ---------------- general shared scan: any P (multi-pass), either layout ----------------
        const uint32_t npass = (P + kMaxKeysPerPass - 1) / kMaxKeysPerPass;
        if (tile < ntiles) issue(tile);
        while (tile < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t w[G::LANE_DWORDS];
            read_lane_data<C, VPL>(lds_wave, lane, w);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint64_t next = tile + stride;
            if (next < ntiles) issue(next);
            const bool full = tile < nfull;

            for (uint32_t pass = 0; pass < npass; pass++) {
                uint32_t key[kMaxKeysPerPass];
                if (one_pass) {
#pragma unroll
                    for (int q = 0; q < kMaxKeysPerPass; q++) key[q] = a.key[q];
                } else {
#pragma unroll
                    for (int q = 0; q < kMaxKeysPerPass; q++)
                        key[q] = __builtin_amdgcn_readfirstlane((uint32_t)a.keys_dev[pass * kMaxKeysPerPass + q]);
                }
                uint32_t res[NK][WORDS];
                decode_words<C, VPL, 0, NK, MODE, G::LANE_DWORDS>(w, res, key);
#pragma unroll
                for (int q = 0; q < NK; q++) {
                    const uint32_t k = pass * kMaxKeysPerPass + q;
                    if (k < P) {
                    uint32_t cnt = 0;
                    if (a.layout == 0) {
                        uint8_t *dst = a.out + (uint64_t)k * a.out_stride + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                        if (full) {
#pragma unroll
                            for (int j = 0; j < WORDS; j++) cnt += __builtin_popcount(res[q][j]);
                            store_words<WORDS, NTS>(dst, res[q]);
                        } else {
                            cnt = finish_tail(tile, res[q], dst, 1);
                        }
                    } else {
                        // linear: byte of 8-value group g and key k at g*P + k (src/simd_scan_shared_linear.cpp:57)
                        uint8_t *dst = a.out + (tile * G::BITMAP_BYTES + (uint64_t)lane * (WORDS * 4)) * P + k;
                        if (full) {
#pragma unroll
                            for (int j = 0; j < WORDS; j++) {
                                cnt += __builtin_popcount(res[q][j]);
#pragma unroll
                                for (int b = 0; b < 4; b++) dst[(uint64_t)(4 * j + b) * P] = (uint8_t)(res[q][j] >> (8 * b));
                            }
                        } else {
                            cnt = finish_tail(tile, res[q], dst, P);
                        }
                    }
                    if (a.hits) hits_add(a, k, wave_sum(cnt), lane);
                    }
                }
            }
            tile = next;
        }
        hits_finalize(a, P, lane);
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

template <int C, int AUX>
__global__ __launch_bounds__(kBlockThreads) void decompress_kernel(DecompArgs a)
{
    using G = DecompGeom<C>;
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
    for (uint64_t D = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; D < a.out_dwords; D += gstride) {
        const uint64_t grp = D / c;
        const uint32_t r = (uint32_t)(D - grp * c);
        const uint32_t lo_bit = 32 * r;
        const uint32_t k0 = lo_bit / c;
        const uint32_t k1 = (lo_bit + 31) / c;
        uint32_t word = 0;
        for (uint32_t k = k0; k <= k1 && k < 32; k++) {
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
    }
}

} // namespace mi355

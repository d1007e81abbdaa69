// kernels/bitmap.hpp -- bitmap consumers: combine / count / row ids.  Part of kernels.hpp (gfx950 only).
#pragma once

#include "tile.hpp"

namespace mi355 {

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
// Pass 1 (rowid_count_kernel): popcount per chunk of kRowidChunk bytes.  Pass 2 (rowid_scan_kernel): exclusive scan of
// the chunk counts, in groups.  Pass 3 (rowid_write_kernel): each wave expands its chunk.
constexpr int kRowidChunk = 2048; // bytes of bitmap per wave = 16384 rows

struct RowidArgs {
    const uint8_t *bitmap;
    uint64_t nbytes;
    uint64_t first_row;
    unsigned long long *chunk_counts; // nchunks + 1 entries (exclusive scan inside each group of kRowidScanGroup, in place;
                                      // [nchunks] = total), followed by one total per group
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

// Pass 2: exclusive scan of the chunk counts in groups of kRowidScanGroup (one block per group: coalesced loads of
// four counts per thread, wave shuffles, four wave totals through LDS); each group's total goes to group_totals[],
// and pass 3 adds the (few) totals of the groups before its own.  The first version scanned all counts in ONE block
// with every thread walking a private strip: 115 us for the 61 k chunks of 1e9 rows, more than passes 1 and 3 together
// at low selectivity.
constexpr int kRowidScanGroup = 1024;

__device__ __forceinline__ unsigned long long *rowid_group_totals(const RowidArgs &g) { return g.chunk_counts + g.nchunks + 1; }

static __global__ __launch_bounds__(256) void rowid_scan_kernel(RowidArgs g)
{
    __shared__ unsigned long long wave_tot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * kRowidScanGroup + (uint64_t)threadIdx.x * 4;
    unsigned long long v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = i0 + j < g.nchunks ? g.chunk_counts[i0 + j] : 0ull;
    const unsigned long long mine = v[0] + v[1] + v[2] + v[3];
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long run = incl - mine;
    for (int w = 0; w < wave; w++) run += wave_tot[w];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (i0 + j < g.nchunks) g.chunk_counts[i0 + j] = run;
        run += v[j];
    }
    if (threadIdx.x == 255) rowid_group_totals(g)[blockIdx.x] = run;
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
    const unsigned long long *totals = rowid_group_totals(g);
    if (wave == 0) { // the grand total, for the caller's count
        const uint64_t ngroups = (g.nchunks + kRowidScanGroup - 1) / kRowidScanGroup;
        unsigned long long s = 0;
        for (uint64_t i = lane; i < ngroups; i += 64) s += totals[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) g.chunk_counts[g.nchunks] = s;
    }
    for (uint64_t ch = wave; ch < g.nchunks; ch += nwaves) {
        const uint64_t base = ch * kRowidChunk;
        // ids before this chunk = totals of the earlier scan groups + the chunk's prefix inside its group
        unsigned long long before = 0;
        for (uint64_t i = lane; i < ch / kRowidScanGroup; i += 64) before += totals[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
        uint64_t out = before + g.chunk_counts[ch];
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

} // namespace mi355

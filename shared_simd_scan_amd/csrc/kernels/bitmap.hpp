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

} // namespace mi355

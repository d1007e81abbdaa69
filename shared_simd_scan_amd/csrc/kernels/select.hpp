// kernels/select.hpp -- select_kernel: predicate scan -> ascending row ids in ONE launch, no bitmap in HBM.
// Part of kernels.hpp (gfx950 only).
#pragma once

#include "scan.hpp"

namespace mi355 {

// ---- fused scan + selection vector ("fused so the bitmap need not round-trip HBM") --------------------
// The chain scan -> bitmap -> mi355_bitmap_to_rowids_dev writes the bitmap, reads it twice (count, expand) and runs
// four launches.  Here a wave decodes tiles exactly as scan_burst_kernel does, parks the bitmap words of a CHUNK of
// select_tiles(c) consecutive tiles in LDS (16 KiB per wave = 128 Ki rows at 128 values per lane; registers would force
// the tile loop to be unrolled 16 times -- a runtime index into a register array goes to scratch memory) and turns them
// into row ids on the spot.  The only cross-chunk information an id needs is the number of hits in all
// earlier chunks: a decoupled look-back over per-chunk state words supplies it inside the launch.
//
//   state[q] = { status : 2, value : 62 }   status 0 = nothing yet, 1 = value is chunk q's own count (aggregate),
//                                           2 = value is the count of chunks 0..q (inclusive prefix), 3 = poisoned
//   wave of chunk q:  publish {1, count_q}; read the state of the 1024 chunks in front of it in one poll (16 per
//   lane, nearest first), add aggregates until a chunk that already knows its inclusive prefix; publish
//   {2, prefix + count_q}.  The persistent grid has ~1024 waves in flight, each on one chunk, so one poll normally
//   reaches a finished chunk; the look-back costs a few microseconds per chunk of ~30 us.  (A look-back per 8 Ki-row
//   tile would walk the same 1024 in-flight tiles 64 at a time, ~16 dependent polls per 2 us tile.)
//
// Visibility across CUs / XCDs (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
// every state word is ONE naturally aligned 8-byte granule that carries its own tag, written by one agent-scope
// relaxed atomic store (write-through, `sc1`) and polled with agent-scope relaxed atomic loads (`sc1`: served past
// the CU's L1) -- there is no flag / payload ordering to get wrong.  The words are zeroed by a memset node in front
// of every launch.  Forward progress WITHOUT assuming a resident grid: chunks are not dealt out by block index but CLAIMED,
// a BLOCK at a time -- wave 0 takes the block's next four chunks from a ticket counter (one agent-scope atomic add per
// generation, two generations ahead of use; zeroed by the same memset node as the state words) and hands them to the
// block's waves through LDS behind a barrier -- so a chunk a look-back waits for always belongs to a block that is
// running, whose waves publish their aggregates without waiting for anybody.  Blocks that the dispatcher holds back
// (another context's kernel occupying CUs / LDS: contexts and streams are independent, include/mi355_scan.h) simply
// claim later chunks when they start.  Round 2 assigned chunk = blockIdx.x * 4 + wave (+ stride): with part of the grid
// not resident, block 0's second chunk waited for chunks of undispatched blocks until the spin limit and the call
// returned count = ~0.  (One ticket per WAVE and chunk was tried first: 7630 same-address device-scope atomics of ~40 ns
// each in a 0.27 ms kernel -- the counter became the bottleneck, 0.36 ms.  Per block it is a quarter of that, off the
// critical path: the atomic's return is not waited for until a tile later, see the tile loop.)
// The spin limit stays as a guard against a device that makes no progress at all; it is sound now: a wave that gives up
// publishes status 3 (poison) for its chunk, every look-back that meets a poisoned word gives up too, and the count is
// written with an atomic max (the entry point zeroes it), so ~0 always wins over a partial sum.
//
// Expansion, one 32-row word of every lane (2048 rows of the tile... in lane-major row order: lane l owns rows
// [VPL l, VPL l + VPL) of a tile) at a time: writing ids lane by lane would make every store instruction touch 64
// different lines at high selectivity, so the lanes drop their 16-bit in-tile row offsets into a per-wave LDS stage at
// the positions a wave prefix sum of the lane counts gives, then the wave copies the stage out with consecutive lanes
// writing consecutive ids (512 B per store instruction).  Ids must ascend, and lane l's rows all precede lane l+1's:
// the words of a tile are therefore expanded lane-major, i.e. all WORDS words of a lane form ONE run of the prefix sum.
// tiles per chunk (64 Ki rows where it fits): the wave's LDS -- tile + mask image + half-tile id stage + the parked words
// of TWO chunks -- must let four waves share the CU's 160 KiB
constexpr int select_tiles(int c) { return c > 16 ? 16 : (c >= 14 ? 4 : 8); }
constexpr int kSelectWindow = 16;         // state words per lane and poll (1024 per wave)
constexpr uint32_t kSelectSpinLimit = 1u << 20;
constexpr unsigned long long kSelectValueMask = (1ull << 62) - 1ull;
// the chunk-ticket counter sits behind the state words on its own 128-byte line (same allocation, same memset node)
__host__ __device__ constexpr uint64_t select_ticket_index(uint64_t nchunks) { return (nchunks + 15) / 16 * 16 + 16; }
__host__ __device__ constexpr uint64_t select_state_words(uint64_t nchunks) { return select_ticket_index(nchunks) + 16; }
// a tile with at least 1 id per kSelectDenseRatio rows is expanded 64 consecutive rows at a time straight from the
// registers (see expand): the cost of that form is per row, the cost of the LDS stage per id
constexpr uint32_t kSelectDenseRatio = 8;

template <int C, int MODE, int VPL>
__global__ __launch_bounds__(kBlockThreads, 1) void select_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    static_assert(MODE == kModeEq || MODE == kModeRange, "MODE");
    constexpr int WORDS = G::WORDS;
    constexpr int K = select_tiles(C);
    constexpr int HALF = G::TILE_VALUES / 2; // ids one expansion pass can stage (the rows of 32 lanes)
    constexpr int AUX = 2; // the column is streamed once: non-temporal DMA
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][G::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mlds[kWavesPerBlock][1024];
    __shared__ __attribute__((aligned(16))) uint16_t stage[kWavesPerBlock][HALF];
    __shared__ __attribute__((aligned(16))) uint32_t parked[kWavesPerBlock][2][K][64 * WORDS]; // bitmap words of two chunks
    constexpr int LK = narrow_k<C>();
    __shared__ __attribute__((aligned(16))) uint8_t nlut[LK ? (1 << (LK * C)) : 16];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *lds_wave = lds[wave];
    uint8_t *mlds_wave = mlds[wave];
    uint16_t *st = stage[wave];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t nchunks = (tc.ntiles + K - 1) / K;
    unsigned long long *const ticket = a.tile_state + select_ticket_index(nchunks);
    __shared__ unsigned long long mail[2]; // first chunk of the block's generation g + 2, posted by wave 0 at the bottom of generation g

    const uint32_t key[2] = {a.key[0], a.key[1]};
    const uint8_t *const mask = a.and_mask;
    const uint32_t mop = a.mask_op, inv = a.invert;
    auto combine = [mop](uint32_t r, uint32_t m) -> uint32_t {
        return mop == 0 ? (r & m) : mop == 1 ? (r | m) : mop == 2 ? (r ^ m) : (m & ~r);
    };
    unsigned long long *const state = a.tile_state;

    auto issue_tile = [&](uint64_t t) {
        // the tile's mask bytes first (a full tile: one 1 KiB / 512 B image), then the packed tile: the wait at the top
        // of the tile's iteration drains both
        if (mask && t < tc.nfull) {
            if (lane * 16 < G::BITMAP_BYTES)
                __builtin_amdgcn_global_load_lds(MI355_GPTR(mask + t * G::BITMAP_BYTES + lane * 16), MI355_LPTR(mlds_wave), 16, 0, 0);
        }
        tc.template issue<AUX>(a.packed, t, lds_wave, lane);
    };

    // (A/B switches, timing only -- the results are the same: flags bit 3 = chunks dealt out by block index as in round 2,
    // which is only safe while the whole grid is resident; bit 4 = no barrier per generation, with bit 3 only)
    const bool by_index = (a.flags & 8u) != 0, no_barrier = (a.flags & 24u) == 24u;
    const unsigned long long gen_stride = (unsigned long long)gridDim.x * kWavesPerBlock;
    // the block's first two generations: one claim of 4 chunks each (two dependent atomics: the second lands behind the
    // first claims of the blocks that started at about the same time, so generation 1 lies mostly above generation 0)
    if (threadIdx.x == 0) {
        if (by_index) {
            mail[0] = (unsigned long long)blockIdx.x * kWavesPerBlock;
            mail[1] = mail[0] + gen_stride;
        } else {
            const unsigned long long t0 = __hip_atomic_fetch_add(ticket, (unsigned long long)kWavesPerBlock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mail[0] = t0; // (the LDS write needs the value: the first atomic has returned before the second is issued)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            mail[1] = __hip_atomic_fetch_add(ticket, (unsigned long long)kWavesPerBlock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    auto uniform64 = [](unsigned long long v) -> uint64_t { // (the value is the same in every lane: keep it in SGPRs)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    uint64_t base = uniform64(mail[0]), base_next = uniform64(mail[1]); // first chunk of this / the next generation of the block
    __syncthreads(); // (both read before wave 0 posts generation 2 into mail[0])
    if (base + wave < nchunks) issue_tile((base + wave) * K);
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

    bool gave_up = false;

    // one poll = kSelectWindow loads of 64 CONSECUTIVE state words each (512 B, four lines per instruction; a lane
    // reading its own run of 16 words would touch 1024 lines per poll), all issued before the first is used: group k,
    // lane l looks at chunk pos - 64 k - l
    auto poll_issue = [&](int64_t pos, unsigned long long (&s)[kSelectWindow]) {
#pragma unroll
        for (int k = 0; k < kSelectWindow; k++) {
            const int64_t i = pos - 64 * k - lane;
            // chunks before the column: "prefix of nothing" = inclusive 0
            s[k] = i >= 0 ? __hip_atomic_load(state + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
        }
    };

    // hits in all chunks before q (decoupled look-back), then publishes q's inclusive prefix.  The FIRST poll's loads are
    // issued by the caller (poll_issue(q - 1, s)), who puts work that does not need the answer between issue and call.
    auto resolve = [&](uint64_t q, unsigned long long q_hits, unsigned long long (&s)[kSelectWindow]) -> unsigned long long {
        if (q == 0) return 0ull; // chunk 0 published its inclusive prefix right away
        unsigned long long before = 0;
        int64_t pos = (int64_t)q - 1; // nearest chunk of the poll
        uint32_t spins = 0;
        bool done = false;
        while (!done) {
            // aggregates (status 1: a chunk's own count, < 2^17) are summed per lane in 32 bits and over the wave by DPP; the
            // one inclusive prefix that ends the walk (status 2, up to n) is read from its lane -- a 64-bit __shfl_xor
            // butterfly here was twelve dependent ds_bpermute round trips per chunk
            uint32_t acc = 0;
            unsigned long long prefix = 0;
            bool retry = false;
#pragma unroll
            for (int k = 0; k < kSelectWindow; k++) {
                if (!done && !retry) { // wave-uniform
                    const uint32_t status = (uint32_t)(s[k] >> 62);
                    const unsigned long long fmask = __ballot(status == 2);
                    const unsigned long long bmask = __ballot(status == 0);
                    const unsigned long long pmask = __ballot(status == 3);
                    // the nearest chunk that knows its inclusive prefix ends the walk; every chunk nearer than it must
                    // at least have published its own count
                    const int stop = fmask ? __builtin_ctzll(fmask) : 64;
                    const unsigned long long need = stop >= 63 ? ~0ull : ((2ull << stop) - 1ull);
                    if (pmask & need) { // a chunk in front gave up: its prefix will never come
                        gave_up = true;
                        done = true;
                    } else if (bmask & need) {
                        retry = true; // what was summed so far stays valid: resume at this group
                        pos -= 64 * k;
                    } else {
                        if (lane < stop) acc += (uint32_t)s[k];
                        if (stop < 64) {
                            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s[k], stop);
                            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(s[k] >> 32), stop);
                            prefix = (((unsigned long long)hi << 32) | lo) & kSelectValueMask;
                            done = true;
                        }
                    }
                }
            }
            before += (unsigned long long)wave_sum(acc) + prefix;
            if (retry) {
                if (++spins > kSelectSpinLimit || gave_up) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            } else if (!done) {
                pos -= 64 * kSelectWindow;
            }
            if (!done) poll_issue(pos, s);
        }
        if (lane == 0) // (poison, not a partial prefix, when the walk was given up)
            __hip_atomic_store(state + q, gave_up ? (3ull << 62) : ((2ull << 62) | ((before + q_hits) & kSelectValueMask)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return before;
    };

    // ---- expansion of chunk q (its words parked in `pk`) -------------------------------------------------------------
    // Part A needs no prefix: the lane's words of ALL tiles of the chunk in one batch of LDS reads, their counts, and
    // the per-tile prefix sums over the wave two tiles at a time (16-bit fields: a tile has at most 8192 ids) -- K / 2
    // independent DPP chains instead of one dependent chain per tile with an LDS round trip and a readlane in between
    // (one wave per SIMD: nothing else hides those latencies; the per-tile form cost 1400 cycles per tile).
    struct ChunkWords {
        uint32_t b[K][WORDS];
        uint32_t cnt[K];  // the lane's ids per tile
        uint32_t incl[K]; // inclusive prefix over the lanes
        uint32_t tot[K];  // wave-uniform
    };
    auto count_chunk = [&](int ntiles_q, uint32_t(*pk)[64 * WORDS], ChunkWords &cw) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            cw.cnt[k] = 0;
#pragma unroll
            for (int j = 0; j < WORDS; j++) {
                const uint32_t w = pk[k][lane * WORDS + j];
                cw.b[k][j] = k < ntiles_q ? w : 0u; // (tiles behind the column: stale words of an earlier chunk)
                cw.cnt[k] += __builtin_popcount(cw.b[k][j]);
            }
        }
#pragma unroll
        for (int m = 0; m < K / 2; m++) {
            const uint32_t sc = wave_inclusive_scan(cw.cnt[2 * m] | (cw.cnt[2 * m + 1] << 16));
            cw.incl[2 * m] = sc & 0xffffu;
            cw.incl[2 * m + 1] = sc >> 16;
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
            cw.tot[2 * m] = t & 0xffffu;
            cw.tot[2 * m + 1] = t >> 16;
        }
    };

    // Part B: sparse tiles (<= 64 ids: the usual case of a selective predicate) write their few ids straight from the
    // lanes; denser ones go through the LDS stage, tile by tile.
    auto expand = [&](uint64_t q, int ntiles_q, uint32_t(*pk)[64 * WORDS], unsigned long long out, const ChunkWords &cw) {
        bool any_dense = false;
        {
            uint64_t *dst = a.rowids + out;
            const unsigned long long room64 = out >= a.capacity ? 0ull : a.capacity - out;
            const uint32_t room = room64 > 0xffffffffull ? 0xffffffffu : (uint32_t)room64;
            uint32_t off = 0; // ids of the chunk in front of tile k
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t total = cw.tot[k];
                if (total > 64u) any_dense = true;
                if (total != 0u && total <= 64u) { // wave-uniform
                    // one loop over the lane's whole run (lowest set bit of the first non-empty word): the wave iterates
                    // max-hits-per-lane times (1-2 for a selective predicate), not once per word and hit
                    uint32_t b[WORDS];
#pragma unroll
                    for (int j = 0; j < WORDS; j++) b[j] = cw.b[k][j];
                    uint32_t o = off + cw.incl[k] - cw.cnt[k];
                    const uint64_t r0 = a.first_row + (q * K + k) * G::TILE_VALUES + lane * VPL;
                    uint32_t left = cw.cnt[k];
                    while (left) {
                        uint32_t w, jb;
                        if constexpr (WORDS == 4) {
                            w = b[0] ? b[0] : (b[1] ? b[1] : (b[2] ? b[2] : b[3]));
                            jb = b[0] ? 0u : (b[1] ? 32u : (b[2] ? 64u : 96u));
                        } else {
                            w = b[0] ? b[0] : b[1];
                            jb = b[0] ? 0u : 32u;
                        }
                        const uint32_t i = (uint32_t)__builtin_ctz(w);
                        if (o < room) dst[o] = r0 + jb + i;
                        o++;
                        left--;
                        const uint32_t cleared = w & (w - 1);
                        if constexpr (WORDS == 4) {
                            if (jb == 0) b[0] = cleared; else if (jb == 32) b[1] = cleared; else if (jb == 64) b[2] = cleared; else b[3] = cleared;
                        } else {
                            if (jb == 0) b[0] = cleared; else b[1] = cleared;
                        }
                    }
                }
                off += total;
            }
        }
        if (!any_dense) return;
#pragma unroll 1
        for (int k = 0; k < ntiles_q; k++) {
            uint32_t b[WORDS];
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < WORDS; j++) {
                b[j] = pk[k][lane * WORDS + j];
                cnt += __builtin_popcount(b[j]);
            }
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            const uint64_t row0 = a.first_row + (q * K + k) * G::TILE_VALUES;
            if (total <= 64) { // written above
                out += total;
                continue;
            }
            if (total >= (uint32_t)G::TILE_VALUES / kSelectDenseRatio) {
                // Dense tile: no LDS stage.  The 64 rows [128 L + 64 h, +64) are two words of lane L: read them into an
                // SGPR pair (v_readlane), make that pair the EXEC mask, and the hit lanes -- lane l = row l of the step --
                // store their row id at the running offset + v_mbcnt.  One store instruction writes popcount(m)
                // CONSECUTIVE ids (256 B at selectivity 1/2), ~12 mostly scalar instructions per 64 rows, no dependent
                // LDS round trip anywhere: this is what one wave per SIMD can keep up, where the stage's ds_write -> wait
                // -> ds_read -> store chains could not (5e8 ids of 1e9 rows: 1.49 ms, the unfused chain 1.11-1.23).
                const bool roomy = out + total <= a.capacity; // wave-uniform
                unsigned long long o = out;
#pragma unroll 1
                for (int L = 0; L < 64; L++) {
#pragma unroll
                    for (int h = 0; h < WORDS / 2; h++) {
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)b[2 * h], L);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)b[2 * h + 1], L);
                        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                        if (m) { // wave-uniform
                            const uint32_t pos = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                            if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                                if (roomy || o + pos < a.capacity) a.rowids[o + pos] = row0 + (uint32_t)(L * VPL + 64 * h) + lane;
                            }
                            o += (unsigned long long)__builtin_popcountll(m);
                        }
                    }
                }
                out += total;
                continue;
            }
            const uint32_t first_half = __builtin_amdgcn_readlane(incl, 31); // ids of lanes 0..31
            // the stage holds the ids of half a tile: one pass when they fit, else lanes 0..31, then lanes 32..63
            const int npass = total <= (uint32_t)HALF ? 1 : 2;
#pragma unroll 1
            for (int h = 0; h < npass; h++) {
                const uint32_t base = h ? first_half : 0u;
                const uint32_t count = npass == 1 ? total : (h ? total - first_half : first_half);
                if (count == 0) continue;
                if (npass == 1 || (lane >> 5) == h) {
                    // the lane's WORDS words side by side: one loop of max-popcount iterations with WORDS independent
                    // ctz / clear / ds_write chains in each, instead of sum-of-popcounts iterations of one dependent chain
                    // (one wave per SIMD: only instruction-level parallelism hides the latencies here)
                    uint32_t pw[WORDS], ww[WORDS];
                    uint32_t p = incl - cnt - base, any = 0;
#pragma unroll
                    for (int j = 0; j < WORDS; j++) {
                        ww[j] = b[j];
                        pw[j] = p;
                        p += __builtin_popcount(b[j]);
                        any |= b[j];
                    }
                    while (any) {
                        any = 0;
#pragma unroll
                        for (int j = 0; j < WORDS; j++) {
                            if (ww[j]) {
                                const int i = __builtin_ctz(ww[j]);
                                ww[j] &= ww[j] - 1;
                                st[pw[j]++] = (uint16_t)(lane * VPL + 32 * j + i);
                            }
                            any |= ww[j];
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the wave's stage writes are done (LDS is in order per wave)
                // copy-out, four ids per lane and round: the LDS reads of a round are issued together
                uint32_t i = lane;
                for (; i + 192 < count; i += 256) {
                    const uint32_t s0 = st[i], s1 = st[i + 64], s2 = st[i + 128], s3 = st[i + 192];
                    const uint64_t o = out + base + i;
                    if (o + 192 < a.capacity) {
                        a.rowids[o] = row0 + s0;
                        a.rowids[o + 64] = row0 + s1;
                        a.rowids[o + 128] = row0 + s2;
                        a.rowids[o + 192] = row0 + s3;
                    } else {
                        if (o < a.capacity) a.rowids[o] = row0 + s0;
                        if (o + 64 < a.capacity) a.rowids[o + 64] = row0 + s1;
                        if (o + 128 < a.capacity) a.rowids[o + 128] = row0 + s2;
                        if (o + 192 < a.capacity) a.rowids[o + 192] = row0 + s3;
                    }
                }
                for (; i < count; i += 64) {
                    const uint64_t o = out + base + i;
                    if (o < a.capacity) a.rowids[o] = row0 + st[i];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // stage reads done before it is overwritten
            }
            out += total;
        }
    };

    // (Tried and dropped: sending the first poll out half a chunk early, from inside the next chunk's tile loop, so that its
    // round trip passes behind the remaining tiles -- 0.267 -> 0.294 ms at 1e9 x 9 bit, 1/512: the branch and the 32 live
    // registers in the tile loop cost the decode more than the poll's latency.)
    auto finish = [&](uint64_t q, unsigned long long q_hits, int ntiles_q, uint32_t(*pk)[64 * WORDS]) {
        const bool look_back = !(a.flags & 4u), want_ids = q_hits && !(a.flags & 2u); // (flags: tuning aids, ablations)
        unsigned long long s[kSelectWindow];
        if (look_back && q > 0) poll_issue((int64_t)q - 1, s);
        ChunkWords cw;
        if (want_ids) count_chunk(ntiles_q, pk, cw); // needs no prefix: in front of the poll's evaluation
        const unsigned long long before = look_back ? resolve(q, q_hits, s) : 0ull;
        if (want_ids && !gave_up) expand(q, ntiles_q, pk, before, cw);
        // the column's hit count: atomic max over a word the entry point zeroed, so a give-up's ~0 (below) always wins
        if (q == nchunks - 1 && lane == 0 && !gave_up)
            __hip_atomic_fetch_max(a.hits, before + q_hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // Software pipeline over the wave's chunks: chunk g is resolved and expanded AFTER chunk g+1 has been decoded.  The
    // persistent waves run in lock step, so right after its own decode a wave would find none of the ~1000 chunks in
    // front of it finished and wait for the slowest of them -- a grid-wide barrier per generation (measured: 0.51 ms
    // against 0.27 for the unfused chain).  One generation later their aggregates have long been published and the
    // generation before that knows its inclusive prefixes: the look-back is one or two polls and never waits.
    bool pend = false;
    uint64_t pend_chunk = 0;
    unsigned long long pend_hits = 0;
    int pend_ntiles = 0;
    int buf = 0;
    uint32_t gen = 0;
    while (base < nchunks) { // block-uniform: the four waves leave together (a wave whose chunk lies behind the column idles a round)
        const uint64_t chunk = base + wave, chunk_next = base_next + wave;
        // wave 0: the block's ticket for generation gen + 2.  Issued behind the DMA of this chunk's first tile, so the
        // first tile waits with vmcnt(1) -- everything but this youngest operation -- and the atomic's round trip passes
        // behind that tile's decode; its value is read at the bottom of the iteration
        unsigned long long pending_ticket = 0;
        const bool claiming = wave == 0 && !by_index;
        if (claiming && lane == 0)
            pending_ticket = __hip_atomic_fetch_add(ticket, (unsigned long long)kWavesPerBlock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool have = chunk < nchunks;
        const uint64_t tfirst = chunk * K;
        uint32_t(*const park)[64 * WORDS] = parked[wave][buf];
        uint32_t lane_hits = 0;
        int ntiles_here = 0; // tiles of this chunk inside the column
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const uint64_t tile = tfirst + k;
            if (have && tile < tc.ntiles) { // wave-uniform
                ntiles_here = k + 1;
                if (claiming && k == 0)
                    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t w[G::LANE_DWORDS];
                read_lane_data<C, VPL>(lds_wave, lane, w);
                uint32_t mcur[WORDS];
                const bool full = tile < tc.nfull;
                if (mask && full) {
#pragma unroll
                    for (int j = 0; j < WORDS; j++) mcur[j] = ((const uint32_t *)(mlds_wave + lane * (WORDS * 4)))[j];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint64_t next = (k + 1 < K && tile + 1 < tc.ntiles) ? tile + 1 : chunk_next * K;
                if (next < tc.ntiles && (k + 1 < K || chunk_next < nchunks)) issue_tile(next);

                uint32_t r1[1][WORDS];
                if constexpr (LK > 0) {
                    decode_words_narrow<C, VPL, LK, 0, G::LANE_DWORDS>(w, r1, nlut);
                } else {
                    const uint32_t key8[kMaxKeysPerPass] = {key[0], key[1], 0, 0, 0, 0, 0, 0};
                    decode_words<C, VPL, 0, 1, MODE, G::LANE_DWORDS>(w, r1, key8);
                }
#pragma unroll
                for (int j = 0; j < WORDS; j++) r1[0][j] ^= inv;
                if (full) {
                    if (mask) {
#pragma unroll
                        for (int j = 0; j < WORDS; j++) r1[0][j] = combine(r1[0][j], mcur[j]);
                    }
                } else {
                    const int64_t left = (int64_t)(tc.n - tile * G::TILE_VALUES) - (int64_t)lane * VPL;
                    const int valid = left >= VPL ? VPL : (left <= 0 ? 0 : (int)left);
                    if (mask) { // ragged tile: read only the bytes the mask is guaranteed to hold (ceil(n/8))
                        const uint8_t *mp = mask + tile * G::BITMAP_BYTES + lane * (WORDS * 4);
                        const int nbytes = (valid + 7) / 8;
#pragma unroll
                        for (int j = 0; j < WORDS; j++) {
                            uint32_t m = 0;
#pragma unroll
                            for (int b = 0; b < 4; b++)
                                if (4 * j + b < nbytes) m |= (uint32_t)mp[4 * j + b] << (8 * b);
                            r1[0][j] = combine(r1[0][j], m);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < WORDS; j++) r1[0][j] &= tail_mask(valid, j);
                }
#pragma unroll
                for (int j = 0; j < WORDS; j++) {
                    park[k][lane * WORDS + j] = r1[0][j];
                    lane_hits += __builtin_popcount(r1[0][j]);
                }
            }
        }
        unsigned long long chunk_hits = 0;
        if (have) {
            // the chunk's hits, published as its aggregate (chunk 0: as the first inclusive prefix)
            chunk_hits = wave_sum(lane_hits);
            if (lane == 0)
                __hip_atomic_store(state + chunk, ((chunk == 0 ? 2ull : 1ull) << 62) | chunk_hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pend) finish(pend_chunk, pend_hits, pend_ntiles, parked[wave][buf ^ 1]);
        pend = have;
        pend_chunk = chunk;
        pend_hits = chunk_hits;
        pend_ntiles = ntiles_here;
        buf ^= 1;
        // hand-over of generation gen + 2: posted before the barrier, read behind it; the slot is written again two
        // barriers later, when every wave has long read it
        if (by_index) {
            if (!no_barrier) __syncthreads();
            base = base_next;
            base_next += gen_stride;
        } else {
            if (wave == 0 && lane == 0) mail[gen & 1] = pending_ticket;
            __syncthreads();
            base = base_next;
            base_next = uniform64(mail[gen & 1]);
        }
        gen++;
    }
    if (pend) finish(pend_chunk, pend_hits, pend_ntiles, parked[wave][buf ^ 1]);
    if (gave_up && lane == 0) __hip_atomic_fetch_max(a.hits, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace mi355

// kernels/select2.hpp -- select2_kernel: predicate scan -> ascending row ids in ONE launch, decoder and expander waves.
// Part of kernels.hpp (gfx950 only).
#pragma once

#include "select.hpp"

namespace mi355 {

// ---- fused scan + selection vector, round 3: roles --------------------------------------------------------------------
// select_kernel (select.hpp) runs ONE wave per SIMD -- its LDS (tile + parked words of two chunks per wave) admits one
// block of four waves per CU -- and that wave does everything in turn: decode a chunk, look back, expand the chunk before.
// rocprofv3 (profiles/r03_select_pmc.txt): the waves issue ~1 instruction per 10 cycles whatever the selectivity (nothing
// else on the SIMD hides a dependent instruction's latency), so the expansion's instruction stream simply adds to the
// decode's: 0.045 ms of look-back + 0.045 ms of expansion on a 0.19 ms decode at selectivity 1/512, 1.2 ms of expansion
// at 1/2 -- where the unfused chain's expansion kernel, eight waves per SIMD, needs 1.0 ms for everything.
//
// Here a block is TWELVE waves on the same LDS: waves 0-3 DECODE (one per SIMD: chunk -> bitmap words parked in LDS,
// chunk count added to the block's), waves 4-11 EXPAND (two per SIMD) the chunks the decoders parked one generation earlier:
// wave 4 does the block's decoupled look-back and hands every chunk's base to its expanders through LDS, and
// a chunk's tiles are dealt out to its two (sixteen waves -- 128 VGPRs each -- made the decode spill: 0.33 ms against 0.19).  A block barrier per generation is the whole producer / consumer
// protocol (generation g: decoders fill buffer g & 1, expanders drain buffer (g - 1) & 1).  The look-back and the
// expansion no longer cost the decoders anything, and the expansion's instruction streams run four to a SIMD.
//
// The expansion needs no LDS stage any more: a tile with at most 64 ids writes them straight from the lanes (the lane's
// position = DPP prefix of the lanes' counts); every other tile goes 64 consecutive rows at a time -- the two words of
// lane L that hold rows [128 L + 64 h, + 64) are read into an SGPR pair (v_readlane), made the EXEC mask, and the hit
// lanes (lane l = row l of the step) store their row id at the running offset + v_mbcnt: one store instruction writes
// popcount(m) CONSECUTIVE ids, ~12 mostly scalar instructions per 64 rows, no dependent LDS round trip.
//
// Chunks are CLAIMED (a ticket counter, one atomic add per block and generation, see select.hpp), look-back state words,
// poison-on-give-up and the atomic-max count are select_kernel's.
constexpr int kSel2Waves = 12;
constexpr int kSel2Decoders = 4;
constexpr int kSel2ExpPerDec = (kSel2Waves - kSel2Decoders) / kSel2Decoders; // 2
constexpr int kSel2Window = 4; // state words per lane and poll of the block's look-back (256 super-chunks per poll)
// ids per tile up to which a tile goes through the expander's LDS stage: 2048, or 1024 where the CU's 160 KiB leave no more
// next to the tiles, the parked chunks and the narrow widths' predicate table (12 expanders x 2 bytes x this)
constexpr int sel2_stage_ids(int fixed_lds_bytes)
{
    return fixed_lds_bytes + (kSel2Waves - kSel2Decoders) * 2 * 2048 + 2048 <= 160 * 1024 ? 2048 : 1024;
}

template <int C, int MODE, int VPL>
__global__ __launch_bounds__(kSel2Waves * 64, 1) void select2_kernel(ScanArgs a)
{
    using G = ScanGeom<C, VPL>;
    static_assert(MODE == kModeEq || MODE == kModeRange, "MODE");
    constexpr int WORDS = G::WORDS;
    constexpr int K = select_tiles(C);
    constexpr int AUX = 2; // the column is streamed once: non-temporal DMA
    constexpr int D = kSel2Decoders;
    __shared__ __attribute__((aligned(16))) uint8_t lds[D][G::LDS_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mlds[D][1024];
    __shared__ __attribute__((aligned(16))) uint32_t parked[D][2][K][64 * WORDS]; // bitmap words of two chunks per decoder
    constexpr int LK = narrow_k<C>();
    __shared__ __attribute__((aligned(16))) uint8_t nlut[LK ? (1 << (LK * C)) : 16];
    __shared__ unsigned long long mail[2];              // first chunk of the block's generation g + 2 (posted by wave 0)
    __shared__ unsigned long long s_chunk[D][2];        // chunk parked in the buffer (~0: none)
    __shared__ unsigned long long s_hits[D][2];         // its hit count
    __shared__ int s_ntiles[D][2];                      // its tiles inside the column
    __shared__ unsigned long long s_sum[2];             // the generation's ids over the block's D chunks (LDS atomic adds)
    __shared__ unsigned int s_arrive[2];                // decoders that have added theirs
    __shared__ unsigned long long s_base[D];            // look-back result: ids in front of the chunk being expanded
    __shared__ unsigned int s_base_tag[D];              // generation + 1 the base belongs to; bit 31: the look-back gave up
    __shared__ uint32_t s_tilepre[D][K];                // ids of the chunk being expanded in front of each of its tiles
    __shared__ unsigned int s_pre_tag[D];               // generation + 1 they belong to
    // expanders: 16-bit in-tile row offsets of a tile's ids in id order (tiles of 65 .. STAGE_IDS ids), so that the copy-out
    // writes 64 consecutive ids per store instruction
    constexpr int STAGE_IDS = sel2_stage_ids(D * (G::LDS_BYTES + 1024) + D * 2 * K * 64 * WORDS * 4 + (LK ? (1 << (LK * C)) : 16));
    __shared__ __attribute__((aligned(16))) uint16_t stage_all[kSel2Waves - D][STAGE_IDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool decoder = wave < D;
    const int d = wave & (D - 1);            // decoders: own slot; expanders 4 + d, 8 + d, 12 + d: the slot they drain (same SIMD)
    const int e = decoder ? 0 : (wave - D) / D; // expander's index among the chunk's three
    uint8_t *lds_wave = lds[d];
    uint8_t *mlds_wave = mlds[d];
    const TileCtx<C, VPL> tc(a.n);
    const uint64_t nchunks = (tc.ntiles + K - 1) / K;
    unsigned long long *const ticket = a.tile_state + select_ticket_index(nchunks);
    unsigned long long *const state = a.tile_state;

    const uint32_t key[2] = {a.key[0], a.key[1]};
    const uint8_t *const mask = a.and_mask;
    const uint32_t mop = a.mask_op, inv = a.invert;
    auto combine = [mop](uint32_t r, uint32_t m) -> uint32_t {
        return mop == 0 ? (r & m) : mop == 1 ? (r | m) : mop == 2 ? (r ^ m) : (m & ~r);
    };
    auto uniform64 = [](unsigned long long v) -> uint64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    auto issue_tile = [&](uint64_t t) {
        if (mask && t < tc.nfull) {
            if (lane * 16 < G::BITMAP_BYTES)
                __builtin_amdgcn_global_load_lds(MI355_GPTR(mask + t * G::BITMAP_BYTES + lane * 16), MI355_LPTR(mlds_wave), 16, 0, 0);
        }
        tc.template issue<AUX>(a.packed, t, lds_wave, lane);
    };

    // the block's first two generations (two dependent claims, see select.hpp)
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __hip_atomic_fetch_add(ticket, (unsigned long long)D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mail[0] = t0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        mail[1] = __hip_atomic_fetch_add(ticket, (unsigned long long)D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < D) s_base_tag[threadIdx.x] = s_pre_tag[threadIdx.x] = 0;
    if (threadIdx.x < 2) {
        s_sum[threadIdx.x] = 0;
        s_arrive[threadIdx.x] = 0;
    }
    if constexpr (LK > 0) {
        constexpr uint32_t fmask = (1u << C) - 1u;
        for (uint32_t en = threadIdx.x; en < (1u << (LK * C)); en += kSel2Waves * 64) {
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < LK; j++) {
                const uint32_t f = (en >> (j * C)) & fmask;
                const bool hit = (MODE == kModeRange) ? (f - key[0]) <= key[1] : f == key[0];
                m |= (hit ? 1u : 0u) << j;
            }
            nlut[en] = (uint8_t)m;
        }
    }
    __syncthreads();
    uint64_t base = uniform64(mail[0]), base_next = uniform64(mail[1]);
    __syncthreads(); // (both read before wave 0 posts generation 2 into mail[0])
    if (decoder && base + d < nchunks) issue_tile((base + d) * K);

    // the decoder is the wave its SIMD must never keep waiting: the expanders (look-back polls, spin-waits, expansion) take
    // the issue slots it leaves
    if (decoder)
        __builtin_amdgcn_s_setprio(3);
    else
        __builtin_amdgcn_s_setprio(0);
    bool gave_up = false;
    uint32_t dbg_polls = 0, dbg_retries = 0, dbg_tagspins = 0; // (flags bit 6: diagnostics, written behind the count)
    // ---- the look-back, one per BLOCK and generation -------------------------------------------------------------------
    // A block's D chunks of a generation are consecutive (one ticket), so the look-back's unit is the block-generation
    // ("super-chunk" base / D): its aggregate leaves with the last of the block's decoders to finish (LDS atomics below), ONE
    // expander wave (wave D) looks back over the super-chunks in front and hands every decoder's chunk its base through
    // LDS.  One poll = W loads of 64 consecutive state words = 256 super-chunks, the whole grid's generation.  (Round 3's
    // first form looked back per chunk: four waves per block polling 1024 state words = 8 KiB each, two or three times per
    // generation -- a quarter of the column's own traffic through the same L2 and address pipelines, and 0.08 ms on a
    // 0.21 ms decode.)
    constexpr int W = kSel2Window;
    auto poll_issue = [&](int64_t pos, unsigned long long (&s)[W]) {
#pragma unroll
        for (int k = 0; k < W; k++) {
            const int64_t i = pos - 64 * k - lane;
            s[k] = i >= 0 ? __hip_atomic_load(state + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
        }
    };
    auto resolve = [&](uint64_t q, unsigned long long q_hits, unsigned long long (&s)[W]) -> unsigned long long {
        if (q == 0) return 0ull; // chunk 0 published its inclusive prefix right away
        unsigned long long before = 0;
        int64_t pos = (int64_t)q - 1;
        uint32_t spins = 0;
        bool done = false;
        while (!done) {
            uint32_t acc = 0;
            unsigned long long prefix = 0;
            bool retry = false;
#pragma unroll
            for (int k = 0; k < W; k++) {
                if (!done && !retry) { // wave-uniform
                    const uint32_t status = (uint32_t)(s[k] >> 62);
                    const unsigned long long fmask = __ballot(status == 2);
                    const unsigned long long bmask = __ballot(status == 0);
                    const unsigned long long pmask = __ballot(status == 3);
                    const int stop = fmask ? __builtin_ctzll(fmask) : 64;
                    const unsigned long long need = stop >= 63 ? ~0ull : ((2ull << stop) - 1ull);
                    if (pmask & need) { // a chunk in front gave up: its prefix will never come
                        gave_up = true;
                        done = true;
                    } else if (bmask & need) {
                        retry = true;
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
            dbg_polls++;
            if (retry) {
                dbg_retries++;
                if (++spins > kSelectSpinLimit || gave_up) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(32); // (a decoder shares this SIMD: poll rarely)
            } else if (!done) {
                pos -= 64 * W;
            }
            if (!done) poll_issue(pos, s);
        }
        if (lane == 0)
            __hip_atomic_store(state + q, gave_up ? (3ull << 62) : ((2ull << 62) | ((before + q_hits) & kSelectValueMask)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return before;
    };

    // ---- one tile's ids, `out` = index of its first id ------------------------------------------------------------------
    auto expand_tile = [&](uint64_t tile, const uint32_t (&b)[WORDS], uint32_t cnt, uint32_t incl, uint32_t total, unsigned long long out) {
        const uint64_t row0 = a.first_row + tile * G::TILE_VALUES;
        if (total <= 64u) {
            // one loop over the lane's whole run (lowest set bit of the first non-empty word): max-hits-per-lane iterations
            uint32_t w[WORDS];
#pragma unroll
            for (int j = 0; j < WORDS; j++) w[j] = b[j];
            unsigned long long o = out + incl - cnt;
            const uint64_t r0 = row0 + (uint64_t)lane * VPL;
            uint32_t left = cnt;
            while (left) {
                uint32_t ww, jb;
                if constexpr (WORDS == 4) {
                    ww = w[0] ? w[0] : (w[1] ? w[1] : (w[2] ? w[2] : w[3]));
                    jb = w[0] ? 0u : (w[1] ? 32u : (w[2] ? 64u : 96u));
                } else {
                    ww = w[0] ? w[0] : w[1];
                    jb = w[0] ? 0u : 32u;
                }
                const uint32_t i = (uint32_t)__builtin_ctz(ww);
                if (o < a.capacity) a.rowids[o] = r0 + jb + i;
                o++;
                left--;
                const uint32_t cleared = ww & (ww - 1);
                if constexpr (WORDS == 4) {
                    if (jb == 0) w[0] = cleared; else if (jb == 32) w[1] = cleared; else if (jb == 64) w[2] = cleared; else w[3] = cleared;
                } else {
                    if (jb == 0) w[0] = cleared; else w[1] = cleared;
                }
            }
            return;
        }
        if (total <= (uint32_t)STAGE_IDS) {
            // Medium density: every store instruction occupies the CU's one vector-memory address pipeline for a wave's worth of
            // cycles however few lanes it has, so the 64-rows-at-a-time form below (one store per 64 rows: 128 per tile) made a
            // 1/64 selection cost 0.6 ms.  Compact first: the lanes drop the 16-bit offsets of their ids into the stage at the
            // positions their prefix gives (WORDS independent ctz / clear / ds_write chains), then consecutive lanes copy
            // consecutive ids out: total / 64 store instructions.
            uint16_t *const st = stage_all[wave - D];
            {
                uint32_t pw[WORDS], ww[WORDS];
                uint32_t p = incl - cnt, any = 0;
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
            uint32_t i = lane;
            for (; i + 192 < total; i += 256) {
                const uint32_t s0 = st[i], s1 = st[i + 64], s2 = st[i + 128], s3 = st[i + 192];
                const uint64_t o = out + i;
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
            for (; i < total; i += 64) {
                const uint64_t o = out + i;
                if (o < a.capacity) a.rowids[o] = row0 + st[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // stage reads done before the next tile overwrites it
            return;
        }
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
    };

    uint64_t prev_base = ~0ull;
    uint32_t gen = 0;
    const bool stamps = (a.flags & 64u) && blockIdx.x == 0 && lane == 0; // diagnostics: 8 words per generation behind the ids
    auto stamp = [&](int slot) {
        if (stamps && gen < 64) a.rowids[a.capacity + gen * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    };
    while (true) {
        const bool dec_active = base < nchunks;                      // block-uniform
        const bool exp_active = gen > 0 && prev_base < nchunks;      // chunks were parked in the generation before
        if (!dec_active && !exp_active) break;
        const int buf = gen & 1;
        if (wave == 0) stamp(0);
        if (decoder) {
            // ---- decode chunk base + d into parked[d][buf] -------------------------------------------------------------------
            const uint64_t chunk = base + d, chunk_next = base_next + d;
            unsigned long long pending_ticket = 0;
            const bool claiming = wave == 0 && dec_active;
            if (claiming && lane == 0) // the block's ticket for generation gen + 2 (select.hpp: behind the first tile's DMA, vmcnt(1))
                pending_ticket = __hip_atomic_fetch_add(ticket, (unsigned long long)D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool have = dec_active && chunk < nchunks;
            const uint64_t tfirst = chunk * K;
            uint32_t(*const park)[64 * WORDS] = parked[d][buf];
            uint32_t lane_hits = 0;
            int ntiles_here = 0;
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
                                for (int bb = 0; bb < 4; bb++)
                                    if (4 * j + bb < nbytes) m |= (uint32_t)mp[4 * j + bb] << (8 * bb);
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
            if (have) chunk_hits = wave_sum(lane_hits);
            if (wave == 0) stamp(1);
            if (lane == 0) {
                s_chunk[d][buf] = have ? chunk : ~0ull;
                s_hits[d][buf] = chunk_hits;
                s_ntiles[d][buf] = ntiles_here;
                if (claiming) mail[buf] = pending_ticket;
                if (dec_active) {
                    // the super-chunk's aggregate (super-chunk 0: the first inclusive prefix) leaves with the last decoder
                    __hip_atomic_fetch_add(&s_sum[buf], chunk_hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__hip_atomic_fetch_add(&s_arrive[buf], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(D - 1)) {
                        const unsigned long long total = __hip_atomic_load(&s_sum[buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const uint64_t sc = base / D;
                        __hip_atomic_store(state + sc, ((sc == 0 ? 2ull : 1ull) << 62) | (total & kSelectValueMask), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        } else if (exp_active) {
            // ---- expand the chunk decoder d parked in the generation before ------------------------------------------------------
            const int pb = buf ^ 1;
            const bool look_back = !(a.flags & 4u); // (flags: timing ablations)
            const bool looker = wave == D;          // first expander of decoder 0: the block's look-back
            unsigned long long before = 0;
            if (looker) {
                const uint64_t sc = prev_base / D;
                unsigned long long s[W];
                // (tried: 7 us of sleep in front of the first poll, so that the slower blocks of the generation that has just
                // ended have published -- fewer repeats, the same time)
                if (look_back && sc > 0) poll_issue((int64_t)sc - 1, s);
                const unsigned long long total = uniform64(s_sum[pb]);
                unsigned long long hd[D];
#pragma unroll
                for (int j = 0; j < D; j++) hd[j] = uniform64(s_hits[j][pb]);
                if (lane == 0) { // (the decoders add to these again in the generation after this one, behind the barrier)
                    s_sum[pb] = 0;
                    s_arrive[pb] = 0;
                }
                stamp(2);
                if (look_back) before = resolve(sc, total, s);
                stamp(3);
                if (lane == 0) {
                    unsigned long long run = before;
#pragma unroll
                    for (int j = 0; j < D; j++) {
                        s_base[j] = run; // (release: the base is in LDS before its tag)
                        run += hd[j];
                    }
#pragma unroll
                    for (int j = 0; j < D; j++)
                        __hip_atomic_store(&s_base_tag[j], (gen + 1u) | (gave_up ? 0x80000000u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // the column's hit count: atomic max over a word the entry point zeroed, so a give-up's ~0 always wins
                    if (prev_base + D >= nchunks && !gave_up)
                        __hip_atomic_fetch_max(a.hits, before + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const uint64_t q = uniform64(s_chunk[d][pb]);
            if (q != ~0ull) {
                const unsigned long long q_hits = uniform64(s_hits[d][pb]);
                const int ntiles_q = __builtin_amdgcn_readfirstlane(s_ntiles[d][pb]);
                uint32_t(*const pk)[64 * WORDS] = parked[d][pb];
                const bool want_ids = q_hits && !(a.flags & 2u);
                // a sparse chunk (the usual case of a selective predicate) is its first expander's alone: no tile prefixes, no
                // hand-over, the other expander leaves the SIMD to the decoder
                const bool sparse_chunk = q_hits <= 64ull * K;
                if (want_ids && !sparse_chunk && e == kSel2ExpPerDec - 1) {
                    // pass 1, by the chunk's LAST expander while the block's look-back runs: every tile's id count -> exclusive
                    // prefix inside the chunk, for all of the chunk's expanders (two tiles per DPP scan, 16-bit fields: a tile
                    // has at most 8192 ids; the scans are independent chains)
                    uint32_t tot[K];
#pragma unroll
                    for (int m = 0; m < K / 2; m++) {
                        uint32_t c0 = 0, c1 = 0;
#pragma unroll
                        for (int j = 0; j < WORDS; j++) {
                            c0 += 2 * m < ntiles_q ? __builtin_popcount(pk[2 * m][lane * WORDS + j]) : 0;
                            c1 += 2 * m + 1 < ntiles_q ? __builtin_popcount(pk[2 * m + 1][lane * WORDS + j]) : 0;
                        }
                        const uint32_t t = wave_sum(c0 | (c1 << 16));
                        tot[2 * m] = t & 0xffffu;
                        tot[2 * m + 1] = t >> 16;
                    }
                    static_assert(K % 2 == 0, "tiles per chunk");
                    if (lane == 0) {
                        uint32_t run = 0;
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            s_tilepre[d][k] = run;
                            run += tot[k];
                        }
                        __hip_atomic_store(&s_pre_tag[d], gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (want_ids && !looker && (sparse_chunk ? e == 0 : true)) { // the chunk's base from the block's look-back
                    unsigned int tag;
                    uint32_t spins = 0;
                    do {
                        tag = __hip_atomic_load(&s_base_tag[d], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if ((tag & 0x7fffffffu) == gen + 1u) break;
                        dbg_tagspins++;
                        __builtin_amdgcn_s_sleep(16); // (a decoder shares this SIMD: poll rarely)
                    } while (++spins < (1u << 28));
                    gave_up = gave_up || (tag & 0x80000000u) || (tag & 0x7fffffffu) != gen + 1u;
                    before = uniform64(__hip_atomic_load(&s_base[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                }
                if (want_ids && sparse_chunk) {
                    if (e == 0 && !gave_up) {
                        unsigned long long off = before;
#pragma unroll 1
                        for (int k = 0; k < ntiles_q; k++) {
                            uint32_t b[WORDS], c = 0;
#pragma unroll
                            for (int j = 0; j < WORDS; j++) {
                                b[j] = pk[k][lane * WORDS + j];
                                c += __builtin_popcount(b[j]);
                            }
                            const uint32_t incl = wave_inclusive_scan(c);
                            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            if (total) expand_tile(q * K + k, b, c, incl, total, off);
                            off += total;
                        }
                    }
                } else if (want_ids) {
                    if (e != kSel2ExpPerDec - 1) { // wait for the tile prefixes from the last expander
                        uint32_t spins = 0;
                        while (__hip_atomic_load(&s_pre_tag[d], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gen + 1u && ++spins < (1u << 28))
                            __builtin_amdgcn_s_sleep(16);
                    }
                    if (!gave_up) {
                        // pass 2: this expander's tiles (k = e, e + kSel2ExpPerDec, ...), ONE instance of the expansion code
#pragma unroll 1
                        for (int k = e; k < ntiles_q; k += kSel2ExpPerDec) {
                            uint32_t b[WORDS], c = 0;
#pragma unroll
                            for (int j = 0; j < WORDS; j++) {
                                b[j] = pk[k][lane * WORDS + j];
                                c += __builtin_popcount(b[j]);
                            }
                            const uint32_t incl = wave_inclusive_scan(c);
                            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            const uint32_t pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tilepre[d][k]);
                            if (total) expand_tile(q * K + k, b, c, incl, total, before + pre);
                        }
                    }
                }
            }
        }
        if (wave == D) stamp(4);
        __syncthreads();
        if (wave == 0) stamp(5);
        prev_base = base;
        base = base_next;
        if (dec_active) base_next = uniform64(mail[buf]);
        gen++;
    }
    if (gave_up && lane == 0) __hip_atomic_fetch_max(a.hits, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((a.flags & 64u) && lane == 0 && !decoder) { // diagnostics: the caller's count buffer has 4 words
        __hip_atomic_fetch_add(a.hits + 1, (unsigned long long)dbg_polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.hits + 2, (unsigned long long)dbg_retries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.hits + 3, (unsigned long long)dbg_tagspins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace mi355

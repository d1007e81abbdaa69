/*
 * oracle_avx2.c -- TEST INFRASTRUCTURE / CPU BASELINE, never linked into or called by the product.
 *
 * An own AVX2 restatement of the reference's fastest CPU path, the 256-bit "in-place compare" equality scan
 * (RRr89/Shared_SIMD_Scan src/simd_scan.cpp:257-306 `scan_256_unrolled` with the masks of
 * src/simd_scan_commons.hpp:93-170): per group of 8 values, gather the 4 bytes that hold each value into its
 * 32-bit lane with one byte shuffle, AND with the lane's shifted field mask, compare against the key shifted by the
 * same amount (no shift of the data), take the 8 sign bits; four groups make one 32-bit bitmap word.  It exists so
 * that bench.py's `cpu_baseline` has a vectorised, multi-core CPU leg on hosts that got a clean checkout (the compiled
 * reference under oracle/_ref is git-ignored), and it is pinned to the reference-produced golden vectors by
 * tests/test_oracle_golden.py.
 *
 * Differences from the reference's function, all deliberate:
 *   - width is a run-time argument; the two 128-bit halves are loaded from their own byte offsets (the reference loads
 *     the same 16 bytes into both halves, which limits it to c <= 15: src/simd_scan.cpp:224), so this is
 *     exact for c <= 25; wider columns take the scalar oracle;
 *   - canonical tail: bits >= n are 0 and hits counts [0, n) only (the reference's variants disagree past n);
 *   - OpenMP over row ranges (the reference is single-threaded): `threads` = 1 reproduces its shape.
 * Reads at most 16 bytes past the last byte that holds a value of the chunk being decoded: callers pass the
 * reference's padded buffers (compressed_buffer_size = payload + 256, src/simd_scan.hpp:20-26).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

uint64_t oracle_scan_eq(const uint8_t *packed, size_t n, unsigned c, int32_t key, uint8_t *bitmap); /* oracle.c */

ORACLE_API int oracle_avx2_available(void)
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
    return 0;
#endif
}

#if defined(__x86_64__)
typedef struct {
    __m256i shuffle, field, key;
    size_t hi_offset; /* byte offset of the upper half's 16-byte load relative to the group's first byte */
} eq_masks;

/* value k of an 8-group sits at bit c*k of the group's c bytes: byte (c*k)/8, bit (c*k)%8 (src/simd_scan_commons.hpp:93-111) */
__attribute__((target("avx2"))) static eq_masks make_masks(unsigned c, uint32_t key)
{
    eq_masks m;
    uint8_t sh[32];
    uint32_t field[8], kv[8];
    const uint32_t ones = c == 32 ? 0xffffffffu : ((1u << c) - 1u);
    m.hi_offset = (4u * c) / 8u;
    for (unsigned k = 0; k < 8; k++) {
        const unsigned byte = (c * k) / 8u - (k >= 4 ? (unsigned)m.hi_offset : 0u);
        const unsigned pad = (c * k) % 8u;
        for (unsigned b = 0; b < 4; b++) sh[4 * k + b] = (uint8_t)(byte + b);
        field[k] = ones << pad;
        kv[k] = key << pad;
    }
    m.shuffle = _mm256_loadu_si256((const __m256i *)sh);
    m.field = _mm256_loadu_si256((const __m256i *)field);
    m.key = _mm256_loadu_si256((const __m256i *)kv);
    return m;
}

/* 8 result bits for the 8 values whose first byte is p */
__attribute__((target("avx2"))) static inline uint32_t group8(const uint8_t *p, const eq_masks *m)
{
    const __m128i lo = _mm_loadu_si128((const __m128i *)p);
    const __m128i hi = _mm_loadu_si128((const __m128i *)(p + m->hi_offset));
    const __m256i src = _mm256_inserti128_si256(_mm256_castsi128_si256(lo), hi, 1);
    const __m256i lanes = _mm256_and_si256(_mm256_shuffle_epi8(src, m->shuffle), m->field);
    return (uint32_t)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(lanes, m->key)));
}

/* rows [0, n) of a chunk that starts on a byte boundary (a multiple of 8 rows); returns its hit count */
__attribute__((target("avx2"))) static uint64_t scan_chunk(const uint8_t *p, size_t n, unsigned c, const eq_masks *m, uint8_t *out)
{
    uint64_t hits = 0;
    const size_t words = n / 32;
    for (size_t w = 0; w < words; w++) {
        const uint8_t *q = p + w * 4 * (size_t)c;
        const uint32_t bits = group8(q, m) | group8(q + c, m) << 8 | group8(q + 2 * (size_t)c, m) << 16 |
                              group8(q + 3 * (size_t)c, m) << 24;
        memcpy(out + 4 * w, &bits, 4);
        hits += (uint64_t)__builtin_popcount(bits);
    }
    /* ragged end: whole groups of 8, then mask the bits >= n of the last byte */
    for (size_t g = words * 4; g * 8 < n; g++) {
        uint32_t bits = group8(p + g * (size_t)c, m);
        const size_t left = n - g * 8;
        if (left < 8) bits &= (1u << left) - 1u;
        out[g] = (uint8_t)bits;
        hits += (uint64_t)__builtin_popcount(bits);
    }
    return hits;
}
#endif

/* bitmap: ceil(n/8) bytes, written completely.  threads <= 0: all the cores OpenMP sees. */
ORACLE_API uint64_t oracle_avx2_scan_eq(const uint8_t *packed, size_t n, unsigned c, int32_t key, uint8_t *bitmap, int threads)
{
#if defined(__x86_64__)
    if (c >= 1 && c <= 25 && oracle_avx2_available()) {
        const size_t nb = (n + 7) / 8;
        /* keys are compared unmasked (SURVEY 8c hazard 5): a key outside [0, 2^c) matches nothing.  The shifted key
         * then has bits outside the lane's field mask (or was negative): give it the same treatment explicitly, so
         * that a shift out of the 32-bit lane can never fake a match */
        if (key < 0 || ((uint32_t)key >> c) != 0) {
            memset(bitmap, 0, nb);
            return 0;
        }
        const eq_masks m = make_masks(c, (uint32_t)key);
        uint64_t hits = 0;
        int nt = 1;
#ifdef _OPENMP
        nt = threads > 0 ? threads : omp_get_max_threads();
#else
        (void)threads;
#endif
        /* chunks of whole 32-row words, so every chunk starts on a byte boundary of both streams */
        const size_t words = (n + 31) / 32;
        size_t per = (words + (size_t)nt - 1) / (size_t)nt;
        if (per < 2048) per = 2048; /* do not spread small columns over many cores */
        const long nchunks = (long)((words + per - 1) / per);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) reduction(+ : hits) schedule(static)
#endif
        for (long ch = 0; ch < nchunks; ch++) {
            const size_t w0 = (size_t)ch * per;
            const size_t w1 = w0 + per < words ? w0 + per : words;
            const size_t r0 = w0 * 32, r1 = w1 * 32 < n ? w1 * 32 : n;
            hits += scan_chunk(packed + r0 / 8 * c, r1 - r0, c, &m, bitmap + r0 / 8);
        }
        return hits;
    }
#endif
    (void)threads;
    return oracle_scan_eq(packed, n, c, key, bitmap);
}

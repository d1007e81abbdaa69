/*
 * oracle.c -- CPU restatement of the reference's bit-packed column path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The shipped
 * engine (shared_simd_scan_amd/csrc, libmi355scan.so) never links or calls it.
 *
 * Every function restates, in plain scalar C, what the reference computes for the
 * hot path; citations are to /root/reference (RRr89/Shared_SIMD_Scan).  The width `c`
 * is a runtime parameter here (the reference hard-wires BITS_NEEDED = 9,
 * src/simd_scan.hpp:12).
 *
 * Pinning: tests/test_oracle_golden.py checks this file against tests/golden/ref_*.npz,
 * which were produced by the reference itself (oracle/_ref, built from the reference's own
 * sources by oracle/Makefile) with tests/golden/make_golden.py, and against the
 * known-answer tests the reference's own test-suite holds (test/simd_scan_tests.cpp,
 * test/util_tests.cpp).
 *
 * Canonical tail rule (SURVEY 8c hazard 1): the reference's variants agree on bits [0,n)
 * of a bitmap and differ on bits >= n (pad values decode as 0, so key 0 "matches" the pad
 * in variant-specific amounts).  The oracle -- and the engine -- define bits >= n as 0 and
 * hits = popcount over [0,n).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ---- buffer sizing: src/simd_scan.hpp:20-40 ------------------------------------- */

ORACLE_API size_t oracle_compressed_buffer_size(unsigned c, size_t n)
{
    /* src/simd_scan.hpp:20-26: ceil(c*n/8) + 256 bytes of zero padding */
    size_t bits = (size_t)c * n;
    return bits / 8 + (bits % 8 != 0) + 256;
}

ORACLE_API size_t oracle_decompression_output_buffer_size(size_t n)
{
    return n * 4 + 32; /* src/simd_scan.hpp:28-33 */
}

ORACLE_API size_t oracle_scan_output_buffer_size(size_t n)
{
    return n / 8 + (n % 8 != 0) + 32; /* src/simd_scan.hpp:35-40 */
}

/* ---- get_bit: src/util.cpp:51-58 (pinned by test/util_tests.cpp:15-36) ----------- */

ORACLE_API int oracle_get_bit(const uint8_t *v, size_t i)
{
    return (v[i / 8] & (1u << (i % 8))) != 0;
}

/* ---- compression: src/simd_scan_compression.cpp:53-104 ---------------------------
 * Value i occupies bits [c*i, c*i+c) of a little-endian stream of 64-bit words, LSB
 * first.  The reference ORs the *unmasked* value in (":72 tmp << (i*compression)", which
 * relies on x86 masking the shift count mod 64) and, when a value straddles a word,
 * ORs `v << (64-remaining)` into the current and `v >> remaining` into the next word
 * (:85-97).  Restated with explicit mod-64 arithmetic; `out` must be zeroed and hold
 * oracle_compressed_buffer_size(c,n) bytes.  Values are NOT masked (caller's duty,
 * exactly as in the reference). */
static void pack_generic(const void *values, int elem_bytes, size_t n, unsigned c, uint8_t *out)
{
    uint64_t *buf = (uint64_t *)out;
    for (size_t i = 0; i < n; i++) {
        uint64_t v = elem_bytes == 2 ? ((const uint16_t *)values)[i] : ((const uint32_t *)values)[i];
        size_t bit = (size_t)c * i;
        size_t w = bit / 64;
        unsigned sh = (unsigned)(bit % 64);
        buf[w] |= v << sh;
        if (sh + c > 64)
            buf[w + 1] |= v >> (64 - sh);
    }
}

ORACLE_API void oracle_pack_u16(const uint16_t *values, size_t n, unsigned c, uint8_t *out)
{
    pack_generic(values, 2, n, c, out);
}

ORACLE_API void oracle_pack_u32(const uint32_t *values, size_t n, unsigned c, uint8_t *out)
{
    pack_generic(values, 4, n, c, out);
}

/* ---- value fetch -----------------------------------------------------------------
 * What every reference decompressor computes for index i: the c bits at stream position
 * c*i.  The SSE path does it as: 4-byte gather at byte floor(c*i/8) (pshufb,
 * src/simd_scan_commons.hpp:5-28), shift by pad = (c*i)%8 (pmulld by 1<<(32-c-pad) then
 * psrld 32-c, src/simd_scan_decompression.cpp:256-261), i.e. (load32 >> pad) & (2^c-1),
 * valid while c+7 <= 32.  The scalar path walks 64-bit words
 * (src/simd_scan_decompression.cpp:20-53).  Restated with a 64-bit window assembled from bytes so
 * that any c in [1,32] works and nothing past ceil(c*n/8)+8 bytes is read. */
static inline uint32_t fetch(const uint8_t *p, size_t i, unsigned c)
{
    size_t bit = (size_t)c * i;
    const uint8_t *q = p + bit / 8;
    unsigned sh = (unsigned)(bit % 8);
    uint64_t w = 0;
    unsigned need = (sh + c + 7) / 8; /* <= 5 bytes */
    for (unsigned b = 0; b < need; b++)
        w |= (uint64_t)q[b] << (8 * b);
    uint64_t mask = c == 32 ? 0xffffffffull : ((1ull << c) - 1);
    return (uint32_t)((w >> sh) & mask);
}

/* ---- decompression: src/simd_scan_decompression.cpp (all 9 variants agree on [0,n)) */

ORACLE_API void oracle_decompress(const uint8_t *packed, size_t n, unsigned c, int32_t *out)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; i++)
        out[i] = (int32_t)fetch(packed, (size_t)i, c);
}

/* ---- equality scan: src/simd_scan.cpp (scan_unvectorized :20-100, scan_128 :103-158,
 * scan_256_unrolled :273-306).  Bit i of the bitmap = (value_i == key), byte i/8 bit i%8
 * (src/util.cpp:51-58).  Keys are int and are compared unmasked: a key outside [0,2^c)
 * never matches (SURVEY 8c hazard 5).  Writes exactly ceil(n/8) bytes; returns hits. */

ORACLE_API uint64_t oracle_scan_eq(const uint8_t *packed, size_t n, unsigned c, int32_t key, uint8_t *bitmap)
{
    size_t nbytes = n / 8 + (n % 8 != 0);
    uint64_t hits = 0;
#pragma omp parallel for schedule(static) reduction(+ : hits)
    for (long long g = 0; g < (long long)nbytes; g++) {
        uint8_t out = 0;
        for (unsigned k = 0; k < 8; k++) {
            size_t i = (size_t)g * 8 + k;
            if (i < n && fetch(packed, i, c) == (uint32_t)key)
                out |= (uint8_t)(1u << k);
        }
        bitmap[g] = out;
        hits += (uint64_t)__builtin_popcount(out);
    }
    return hits;
}

/* ---- inclusive range scan: the reference only documents it (src/simd_scan.hpp:76-84,
 * "predicate_low<=key<=predicate_high", prototype commented out).  Semantics from that
 * comment; unsigned compare on the decoded value. */

ORACLE_API uint64_t oracle_scan_range(const uint8_t *packed, size_t n, unsigned c, uint32_t lo, uint32_t hi,
                                      uint8_t *bitmap)
{
    size_t nbytes = n / 8 + (n % 8 != 0);
    uint64_t hits = 0;
#pragma omp parallel for schedule(static) reduction(+ : hits)
    for (long long g = 0; g < (long long)nbytes; g++) {
        uint8_t out = 0;
        for (unsigned k = 0; k < 8; k++) {
            size_t i = (size_t)g * 8 + k;
            if (i < n) {
                uint32_t v = fetch(packed, i, c);
                if (v >= lo && v <= hi)
                    out |= (uint8_t)(1u << k);
            }
        }
        bitmap[g] = out;
        hits += (uint64_t)__builtin_popcount(out);
    }
    return hits;
}

/* ---- shared (multi-predicate) scan -----------------------------------------------
 * layout 0 = per-predicate bitmaps (src/simd_scan_shared.cpp:34-87: outputs[key][g]),
 *            written at out + k*stride, ceil(n/8) bytes each;
 * layout 1 = linear interleaved (src/simd_scan_shared_linear.cpp:48-58: byte of 8-value
 *            group g and key k at g*P + k), ceil(n/8)*P bytes; stride ignored.
 * hits[k] = popcount over [0,n) of predicate k (the reference does not count shared-scan hits). */

ORACLE_API void oracle_shared_scan_eq(const uint8_t *packed, size_t n, unsigned c, const int32_t *keys, size_t P,
                                      int layout, uint8_t *out, size_t stride, uint64_t *hits)
{
    size_t nbytes = n / 8 + (n % 8 != 0);
    if (P > 1024)
        return;
    for (size_t k = 0; k < P; k++)
        hits[k] = 0;
#pragma omp parallel
    {
        uint64_t local[1024];
        uint64_t *lh = local;
        /* P > 1024 is outside what the reference supports (linear_simple tops out at 1024,
         * src/simd_scan_shared_linear.cpp:78); callers keep P <= 1024. */
        for (size_t k = 0; k < P && k < 1024; k++)
            lh[k] = 0;
#pragma omp for schedule(static)
        for (long long g = 0; g < (long long)nbytes; g++) {
            uint32_t v[8];
            unsigned valid = 0;
            for (unsigned j = 0; j < 8; j++) {
                size_t i = (size_t)g * 8 + j;
                if (i < n) {
                    v[j] = fetch(packed, i, c);
                    valid |= 1u << j;
                } else
                    v[j] = 0;
            }
            for (size_t k = 0; k < P; k++) {
                uint8_t b = 0;
                for (unsigned j = 0; j < 8; j++)
                    if ((valid >> j & 1) && v[j] == (uint32_t)keys[k])
                        b |= (uint8_t)(1u << j);
                if (layout == 0)
                    out[k * stride + (size_t)g] = b;
                else
                    out[(size_t)g * P + k] = b;
                lh[k] += (uint64_t)__builtin_popcount(b);
            }
        }
#pragma omp critical
        for (size_t k = 0; k < P; k++)
            hits[k] += lh[k];
    }
}

/* ---- synthetic columns (reference generators src/benchmark.cpp:81,:173,:277) */

static inline uint64_t splitmix64(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* kind 0: v[i] = (first+i) % m            (bench_scan i%5, bench_shared_scan i%P)
 * kind 1: v[i] = splitmix64(seed, first+i) & (2^c-1)
 * kind 2: v[i] = (first+i) & (2^c-1)     (bench_decompression, src/benchmark.cpp:81)
 * `first` is the global row index of value 0 (row-range shards generate their own slice). */
ORACLE_API void oracle_gen_values(int kind, uint64_t first, size_t n, unsigned c, uint64_t param, uint32_t *out)
{
    uint32_t mask = c == 32 ? 0xffffffffu : ((1u << c) - 1);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; i++) {
        uint64_t g = first + (uint64_t)i;
        uint32_t v;
        if (kind == 0)
            v = (uint32_t)(g % param) & mask;
        else if (kind == 1)
            v = (uint32_t)splitmix64(param, g) & mask;
        else
            v = (uint32_t)g & mask;
        out[i] = v;
    }
}

ORACLE_API int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int t)
{
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}

// ref_shim.cpp -- extern "C" door onto the REAL reference implementation.
//
// TEST INFRASTRUCTURE.  This file contains no reference code: it #includes the reference's own
// translation units from where they lie under /root/reference (path given by -DREF_SRC=...) and
// re-exports the functions of src/simd_scan.hpp with C linkage so that Python (ctypes) can drive
// them.  It is compiled only by oracle/Makefile, only when /root/reference exists, into
// oracle/_ref/libref_w<W>.so (git-ignored).  Used to (1) validate oracle.c, (2) generate
// tests/golden/*.npz, (3) optionally serve as bench.py's cpu_baseline ("kind": "reference").
//
// Width: the reference hard-wires `#define BITS_NEEDED 9` (src/simd_scan.hpp:12).  For other
// widths this single translation unit re-defines the macro AFTER the header has been seen
// (#pragma once keeps it from being re-included) and BEFORE the .cpp bodies that read it, so the
// reference's algorithms run unmodified at width REF_WIDTH.  The one exception is the header-only
// template shared_scan_128_linear_static<NUM> (src/simd_scan.hpp:122-236), whose body is expanded
// while the macro is still 9: shared_scan_128_linear_simple is therefore only exported for W == 9.
//
// for c <= 15 every variant is right; for c in {17,21,25} only the SSE-128
// family is (decompress_256*/scan_256* gather from byte >= 16, *_unvectorized truncate to 16 bits).

#ifndef REF_WIDTH
#define REF_WIDTH 9
#endif

#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(name) REF_STR(REF_SRC/name)

#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

// clang-format off
#include REF_FILE(simd_scan.hpp)
#if REF_WIDTH != 9
#undef BITS_NEEDED
#define BITS_NEEDED REF_WIDTH
#endif
#include REF_FILE(util.cpp)
#include REF_FILE(simd_scan_compression.cpp)
#include REF_FILE(simd_scan_decompression.cpp)
#include REF_FILE(simd_scan.cpp)
#include REF_FILE(simd_scan_shared.cpp)
#include REF_FILE(simd_scan_shared_linear.cpp)
// clang-format on

#define REF_API extern "C" __attribute__((visibility("default")))

// Slack kept (as capacity, not size) behind every output vector handed to the reference: several
// variants store whole 32-bit words / whole SIMD registers a little past the sizes that
// src/simd_scan.hpp:28-40 computes.  Callers of ref_decompress* must likewise allocate
// ref_decompression_output_buffer_size(n) + 256 bytes (the *_unrolled variants store 32 ints per
// iteration).
static const size_t SLACK = 256;

REF_API int ref_width(void) { return BITS_NEEDED; }

REF_API size_t ref_compressed_buffer_size(size_t n) { return compressed_buffer_size(BITS_NEEDED, n); }
REF_API size_t ref_decompression_output_buffer_size(size_t n) { return decompression_output_buffer_size(n); }
REF_API size_t ref_scan_output_buffer_size(size_t n) { return scan_output_buffer_size(n); }

REF_API int ref_next_multiple(int number, int multiple) { return next_multiple(number, multiple); }

REF_API int ref_get_bit(const uint8_t *bytes, size_t nbytes, size_t i)
{
    std::vector<uint8_t> v(bytes, bytes + nbytes);
    return get_bit(v, i) ? 1 : 0;
}

// compress_9bit_input (src/simd_scan_compression.cpp:53); out must hold ref_compressed_buffer_size(n) bytes
REF_API void ref_compress(const uint16_t *in, size_t n, uint8_t *out)
{
    std::vector<uint16_t> v(in, in + n);
    auto buf = compress_9bit_input(v);
    size_t bytes = compressed_buffer_size(BITS_NEEDED, n) / 8 * 8; // the reference allocates size/8 words
    std::memcpy(out, buf.get(), bytes);
}

typedef void (*decomp_fn)(__m128i *, size_t, int *);
static decomp_fn decomp_table(int variant)
{
    switch (variant) {
    case 0: return decompress_unvectorized;
    case 1: return decompress_128_sweep;
    case 2: return decompress_128_nosweep;
    case 3: return decompress_128_9bit;
    case 4: return decompress_128;
    case 5: return decompress_128_unrolled;
    case 6: return decompress_128_aligned;
    case 7: return decompress_256;
    case 8: return decompress_256_avx2;
    }
    return nullptr;
}

// out must hold ref_decompression_output_buffer_size(n) + 256 bytes
REF_API int ref_decompress(int variant, const uint8_t *packed, size_t n, int32_t *out)
{
    decomp_fn f = decomp_table(variant);
    if (!f) return -1;
    f((__m128i *)packed, n, out);
    return 0;
}

typedef int (*scan_fn)(int, __m128i *, size_t, std::vector<uint8_t> &);
static scan_fn scan_table(int variant)
{
    switch (variant) {
    case 0: return scan_unvectorized;
    case 1: return scan_128;
    case 2: return scan_128_unrolled;
    case 3: return scan_256;
    case 4: return scan_256_unrolled;
    }
    return nullptr;
}

// out must hold ref_scan_output_buffer_size(n) bytes; returns the reference's `int hits`
REF_API int ref_scan(int variant, int key, const uint8_t *packed, size_t n, uint8_t *out)
{
    scan_fn f = scan_table(variant);
    if (!f) return -1;
    std::vector<uint8_t> o;
    o.reserve(scan_output_buffer_size(n) + SLACK); // some variants store whole words past size(): keep that benign
    o.resize(scan_output_buffer_size(n));
    int hits = f(key, (__m128i *)packed, n, o);
    std::memcpy(out, o.data(), o.size());
    return hits;
}

// Times `reps` calls exactly as do_scan_benchmark does (src/benchmark.cpp:142-163: one output
// vector allocated up front, clock around the call only).  seconds[r] = wall time of rep r.
REF_API int ref_scan_timed(int variant, int key, const uint8_t *packed, size_t n, int reps, double *seconds,
                           uint8_t *out_or_null)
{
    scan_fn f = scan_table(variant);
    if (!f) return -1;
    std::vector<uint8_t> o;
    o.reserve(scan_output_buffer_size(n) + SLACK);
    o.resize(scan_output_buffer_size(n));
    int hits = 0;
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        hits = f(key, (__m128i *)packed, n, o);
        auto t1 = std::chrono::steady_clock::now();
        seconds[r] = std::chrono::duration<double>(t1 - t0).count();
    }
    if (out_or_null) std::memcpy(out_or_null, o.data(), o.size());
    return hits;
}

typedef void (*shared_fn)(std::vector<int> const &, __m128i *, size_t, std::vector<std::vector<uint8_t>> &);
static shared_fn shared_table(int variant)
{
    switch (variant) {
    case 0: return shared_scan_128_sequential;
    case 1: return shared_scan_128_sequential_unrolled;
    case 2: return shared_scan_128_threaded;
    case 3: return shared_scan_128_standard;
    case 4: return shared_scan_128_standard_unrolled;
    case 5: return shared_scan_128_parallel;
    case 6: return shared_scan_256_sequential;
    case 7: return shared_scan_256_standard;
    case 8: return shared_scan_256_parallel;
    }
    return nullptr;
}

// out: P consecutive blocks of ref_scan_output_buffer_size(n) bytes (outputs[k] -> block k)
REF_API int ref_shared_scan(int variant, const int *keys, int P, const uint8_t *packed, size_t n, uint8_t *out)
{
    shared_fn f = shared_table(variant);
    if (!f) return -1;
    std::vector<int> k(keys, keys + P);
    size_t sz = scan_output_buffer_size(n);
    std::vector<std::vector<uint8_t>> o(P);
    for (auto &v : o) { v.reserve(sz + SLACK); v.resize(sz); }
    f(k, (__m128i *)packed, n, o);
    for (int i = 0; i < P; i++) std::memcpy(out + (size_t)i * sz, o[i].data(), sz);
    return 0;
}

REF_API int ref_shared_scan_timed(int variant, const int *keys, int P, const uint8_t *packed, size_t n, int reps,
                                  double *seconds)
{
    shared_fn f = shared_table(variant);
    if (!f) return -1;
    std::vector<int> k(keys, keys + P);
    size_t sz = scan_output_buffer_size(n);
    std::vector<std::vector<uint8_t>> o(P);
    for (auto &v : o) { v.reserve(sz + SLACK); v.resize(sz); }
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        f(k, (__m128i *)packed, n, o);
        auto t1 = std::chrono::steady_clock::now();
        seconds[r] = std::chrono::duration<double>(t1 - t0).count();
    }
    return 0;
}

// linear (interleaved) output: variant 0 = linear_standard, 1 = linear_simple (W == 9 only).
// out must hold P * ref_scan_output_buffer_size(n) bytes (src/benchmark.cpp:248-249).
REF_API int ref_shared_scan_linear(int variant, const int *keys, int P, const uint8_t *packed, size_t n, uint8_t *out)
{
    std::vector<int> k(keys, keys + P);
    std::vector<uint8_t> o;
    o.reserve((size_t)P * scan_output_buffer_size(n) + SLACK);
    o.resize((size_t)P * scan_output_buffer_size(n));
    if (variant == 0)
        shared_scan_128_linear_standard(k, (__m128i *)packed, n, o);
    else if (variant == 1 && REF_WIDTH == 9)
        shared_scan_128_linear_simple(k, (__m128i *)packed, n, o);
    else
        return -1;
    std::memcpy(out, o.data(), o.size());
    return 0;
}

REF_API int ref_decompress_timed(int variant, const uint8_t *packed, size_t n, int reps, double *seconds, int32_t *out)
{
    decomp_fn f = decomp_table(variant);
    if (!f) return -1;
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        f((__m128i *)packed, n, out);
        auto t1 = std::chrono::steady_clock::now();
        seconds[r] = std::chrono::duration<double>(t1 - t0).count();
    }
    return 0;
}

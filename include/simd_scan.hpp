// simd_scan.hpp -- drop-in for the header of RRr89/Shared_SIMD_Scan (its src/simd_scan.hpp:20-123):
// the same free functions, names, argument meaning and return types, implemented on an MI355X through
// the C ABI of libmi355scan.so (include/mi355_scan.h).  Header-only: include it instead of the reference's
// header and link -lmi355scan.  Every width-specific CPU variant of one operation (scalar / SSE-128 /
// AVX-256, unrolled or not) maps onto the same HIP kernel; results agree with the reference on [0,n)
// (and on the whole padded output buffers whenever the reference's variants agree among themselves; see
// DESIGN.md "tail rule").
//
// Differences a caller can observe:
//   * the host-pointer path copies the packed column to the GPU and the result back on every call
//     (drop-in convenience; use the *_dev entry points of mi355_scan.h to keep columns resident in HBM);
//   * like the reference's functions (src/simd_scan_shared.cpp:25-32 calls scan_128 from an OpenMP loop) these may
//     be called from any number of host threads at once: every call below passes ctx == NULL, which is the CALLING
//     THREAD's default context (own scratch and device buffers); nothing is shared between threads;
//   * failures of the device path (no GPU, HIP error) throw std::runtime_error -- the reference has no
//     failure modes; there is no CPU fallback;
//   * BITS_NEEDED may be defined before including this header (the reference hard-wires 9).
#pragma once

#include <bitset>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#if defined(__x86_64__) || defined(_M_X64)
#include <immintrin.h> // __m128i, only used as the opaque pointer type of the reference's signatures
#else
struct alignas(16) __m128i { long long v[2]; };
#endif

#include "mi355_scan.h"

#ifndef BITS_NEEDED
#define BITS_NEEDED 9 // src/simd_scan.hpp:12
#endif

// ---- buffer sizing (src/simd_scan.hpp:20-40), in bytes, padding included ---------------------------
constexpr size_t compressed_buffer_size(uint8_t compression, size_t input_array_size)
{
    size_t mem_size = (size_t)compression * input_array_size;
    return mem_size / 8 + (mem_size % 8 != 0) + 256;
}
constexpr size_t decompression_output_buffer_size(size_t input_array_size) { return input_array_size * 4 + 32; }
constexpr size_t scan_output_buffer_size(size_t input_array_size)
{
    return input_array_size / 8 + (input_array_size % 8 != 0) + 32;
}

// ---- the reference's L0 utilities (src/util.hpp:5-25: next_multiple, get_bit, dump_*, POPCNT) ---------------------
// They are NOT part of the scan path and stay the reference's: its header pulls them in with `#include "util.hpp"`
// (src/simd_scan.hpp:10) and its callers include that file themselves, before (test/simd_scan_tests.cpp:3-4) or after
// (src/main.cpp:7-8, src/benchmark.cpp:2-3) this header.  So inside the reference's tree this header does exactly what
// the reference's does -- the one util.hpp on the include path, defined once, src/util.cpp linked as before.  Only a
// build WITHOUT the reference's tree (no util.hpp reachable) gets the inline equivalents below.
#if __has_include("util.hpp")
#include "util.hpp"
#else
constexpr int next_multiple(int number, int multiple) { return ((number + multiple - 1) / multiple) * multiple; }
// src/util.cpp:51-58
inline bool get_bit(std::vector<uint8_t> const &vector, size_t absolute_index)
{
    return (vector[absolute_index / 8] & (1 << (absolute_index % 8))) > 0;
}
// src/util.cpp:60-67: the uint32_t overload reads element absolute_index/8 and keeps only its low byte
inline bool get_bit(std::vector<uint32_t> const &vector, size_t absolute_index)
{
    const uint8_t element = (uint8_t)vector[absolute_index / 8];
    return (element & (1 << (absolute_index % 8))) > 0;
}
#ifndef POPCNT
#define POPCNT(i) __builtin_popcount(i) // src/util.hpp:17-25
#endif
#endif

namespace mi355_dropin {
inline void check(int rc, const char *what)
{
    if (rc != MI355_OK) throw std::runtime_error(std::string(what) + ": " + mi355_last_error());
}
inline int scan_eq(int key, const void *input, size_t n, std::vector<uint8_t> &output)
{
    uint64_t hits = 0;
    check(mi355_scan_eq(nullptr, input, n, BITS_NEEDED, key, output.data(), &hits), "mi355_scan_eq");
    return (int)hits;
}
inline void shared(std::vector<int> const &keys, const void *input, size_t n, std::vector<std::vector<uint8_t>> &outputs)
{
    std::vector<uint8_t *> ptrs(keys.size());
    for (size_t k = 0; k < keys.size(); k++) ptrs[k] = outputs[k].data();
    check(mi355_shared_scan_eq(nullptr, input, n, BITS_NEEDED, keys.data(), (unsigned)keys.size(), ptrs.data(), nullptr),
          "mi355_shared_scan_eq");
}
inline void shared_linear(std::vector<int> const &keys, const void *input, size_t n, std::vector<uint8_t> &output)
{
    check(mi355_shared_scan_eq_linear(nullptr, input, n, BITS_NEEDED, keys.data(), (unsigned)keys.size(), output.data(),
                                      nullptr),
          "mi355_shared_scan_eq_linear");
}
inline void decompress(const void *input, size_t n, int *output)
{
    check(mi355_decompress(nullptr, input, n, BITS_NEEDED, output), "mi355_decompress");
}
} // namespace mi355_dropin

// ---- compression (src/simd_scan.hpp:46, src/simd_scan_compression.cpp:53-104) ------------------------
inline std::unique_ptr<uint64_t[]> compress_9bit_input(std::vector<uint16_t> &input)
{
    const size_t bytes = compressed_buffer_size(BITS_NEEDED, input.size());
    auto buffer = std::make_unique<uint64_t[]>((bytes + 7) / 8);
    mi355_dropin::check(mi355_pack_u16(nullptr, input.data(), input.size(), BITS_NEEDED, buffer.get()), "mi355_pack_u16");
    return buffer;
}

// ---- decompression (src/simd_scan.hpp:51-73) ----------------------------------------------------------
inline void decompress_unvectorized(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_128_sweep(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_128_nosweep(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_128_9bit(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_128(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_128_unrolled(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_128_aligned(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_256(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }
inline void decompress_256_avx2(__m128i *input, size_t input_size, int *output) { mi355_dropin::decompress(input, input_size, output); }

// ---- equality scan (src/simd_scan.hpp:89-96): returns the number of hits --------------------------------
inline int scan_unvectorized(int predicate_key, __m128i *input, size_t input_size, std::vector<uint8_t> &output) { return mi355_dropin::scan_eq(predicate_key, input, input_size, output); }
inline int scan_128(int predicate_key, __m128i *input, size_t input_size, std::vector<uint8_t> &output) { return mi355_dropin::scan_eq(predicate_key, input, input_size, output); }
inline int scan_128_unrolled(int predicate_key, __m128i *input, size_t input_size, std::vector<uint8_t> &output) { return mi355_dropin::scan_eq(predicate_key, input, input_size, output); }
inline int scan_256(int predicate_key, __m128i *input, size_t input_size, std::vector<uint8_t> &output) { return mi355_dropin::scan_eq(predicate_key, input, input_size, output); }
inline int scan_256_unrolled(int predicate_key, __m128i *input, size_t input_size, std::vector<uint8_t> &output) { return mi355_dropin::scan_eq(predicate_key, input, input_size, output); }

// the range scan the reference declares but never implemented (src/simd_scan.hpp:76-84):
// predicate_low <= value <= predicate_high; returns the number of tuples in the range
inline int scan(int predicate_low, int predicate_high, __m128i *compressed_input, int input_size, std::vector<uint8_t> &output)
{
    uint64_t hits = 0;
    if (predicate_high < 0 || predicate_high < predicate_low) {
        std::memset(output.data(), 0, ((size_t)input_size + 7) / 8);
        return 0;
    }
    mi355_dropin::check(mi355_scan_range(nullptr, compressed_input, (uint64_t)input_size, BITS_NEEDED,
                                         (uint32_t)(predicate_low < 0 ? 0 : predicate_low), (uint32_t)predicate_high,
                                         output.data(), &hits),
                        "mi355_scan_range");
    return (int)hits;
}

// ---- shared scan, one bitmap per predicate (src/simd_scan.hpp:102-113) ---------------------------------
inline void shared_scan_128_sequential(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_128_sequential_unrolled(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_128_threaded(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_128_standard(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_128_standard_unrolled(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_128_parallel(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_256_sequential(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_256_standard(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }
inline void shared_scan_256_parallel(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<std::vector<uint8_t>> &outputs) { mi355_dropin::shared(predicate_keys, input, input_size, outputs); }

// ---- shared scan, one linear output vector (src/simd_scan.hpp:119-123): byte of 8-value group g and
// key k at outputs[g * P + k] ------------------------------------------------------------------------
inline void shared_scan_128_linear_standard(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<uint8_t> &outputs) { mi355_dropin::shared_linear(predicate_keys, input, input_size, outputs); }

template <size_t NUM>
void shared_scan_128_linear_static(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<uint8_t> &output)
{
    std::vector<int> keys(predicate_keys.begin(), predicate_keys.begin() + NUM); // the reference reads NUM keys
    mi355_dropin::shared_linear(keys, input, input_size, output);
}

// src/simd_scan_shared_linear.cpp:64-82: only P in {1,2,4,...,1024}; anything else prints the reference's
// diagnostic to cerr and returns without touching the output
inline void shared_scan_128_linear_simple(std::vector<int> const &predicate_keys, __m128i *input, size_t input_size, std::vector<uint8_t> &output)
{
    const size_t P = predicate_keys.size();
    if (P >= 1 && P <= 1024 && (P & (P - 1)) == 0) {
        mi355_dropin::shared_linear(predicate_keys, input, input_size, output);
    } else {
        std::cerr << "not supported for " << P << " predicate keys!" << std::endl;
    }
}

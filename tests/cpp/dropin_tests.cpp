// dropin_tests.cpp -- the reference's own unit tests (test/simd_scan_tests.cpp, test/util_tests.cpp),
// re-expressed against include/simd_scan.hpp (the MI355X drop-in).  Same cases, same checks; a tiny
// REQUIRE replaces the vendored Catch.  Built and run by tests/test_dropin_cpp.py (-m gpu).
#include <cstdio>
#include <cstdlib>

#include "simd_scan.hpp"

static int g_checks = 0, g_failed = 0;
#define REQUIRE(cond)                                                              \
    do {                                                                           \
        g_checks++;                                                                \
        if (!(cond)) {                                                             \
            g_failed++;                                                            \
            std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #cond); \
        }                                                                          \
    } while (0)

typedef void (*decomp_fn)(__m128i *, size_t, int *);
typedef int (*scan_fn)(int, __m128i *, size_t, std::vector<uint8_t> &);
typedef void (*shared_fn)(std::vector<int> const &, __m128i *, size_t, std::vector<std::vector<uint8_t>> &);

// test/simd_scan_tests.cpp:6-43 "Compress and decompress" (all nine decompress names)
static void compress_and_decompress()
{
    size_t input_size = (1 << BITS_NEEDED) - 3;
    std::vector<uint16_t> input_numbers(input_size);
    for (size_t i = 0; i < input_size; i++) input_numbers[i] = (uint16_t)i;
    auto compressed = compress_9bit_input(input_numbers);
    __m128i *compressed_ptr = (__m128i *)compressed.get();
    const decomp_fn fns[] = {decompress_unvectorized, decompress_128_sweep,    decompress_128_nosweep,
                             decompress_128_9bit,     decompress_128,          decompress_128_unrolled,
                             decompress_128_aligned,  decompress_256,          decompress_256_avx2};
    for (decomp_fn f : fns) {
        size_t output_buffer_size = decompression_output_buffer_size(input_size) / sizeof(int);
        auto result_buffer = std::make_unique<int[]>(output_buffer_size);
        f(compressed_ptr, input_numbers.size(), result_buffer.get());
        for (size_t i = 0; i < input_numbers.size(); i++) REQUIRE(input_numbers[i] == result_buffer[i]);
        for (size_t i = input_numbers.size(); i < output_buffer_size; i++) REQUIRE(result_buffer[i] == 0); // padding untouched
    }
}

// test/simd_scan_tests.cpp:45-82 "SIMD Scan" (all five scan names)
static void simd_scan()
{
    std::vector<uint16_t> input_numbers{1, 2, 3, 3, 2, 1, 1, 2, 3, 1, 2, 3};
    auto compressed = compress_9bit_input(input_numbers);
    __m128i *compressed_ptr = (__m128i *)compressed.get();
    const scan_fn fns[] = {scan_unvectorized, scan_128, scan_128_unrolled, scan_256, scan_256_unrolled};
    for (scan_fn f : fns) {
        auto output_buffer_size = scan_output_buffer_size(input_numbers.size());
        std::vector<uint8_t> output(output_buffer_size);
        int predicate_key = 3;
        int hits = f(predicate_key, compressed_ptr, input_numbers.size(), output);
        REQUIRE(hits == 4);
        for (size_t i = 0; i < input_numbers.size(); i++) REQUIRE(get_bit(output, i) == (input_numbers[i] == predicate_key));
    }
    // the range scan the reference only declares (src/simd_scan.hpp:76-84)
    std::vector<uint8_t> output(scan_output_buffer_size(input_numbers.size()));
    int hits = scan(2, 3, compressed_ptr, (int)input_numbers.size(), output);
    REQUIRE(hits == 8);
    for (size_t i = 0; i < input_numbers.size(); i++) REQUIRE(get_bit(output, i) == (input_numbers[i] >= 2 && input_numbers[i] <= 3));
}

// test/simd_scan_tests.cpp:84-106 "Shared SIMD Scan" (all nine per-predicate names)
static void shared_simd_scan()
{
    std::vector<uint16_t> input_numbers{1, 2, 3, 3, 2, 1, 1, 2, 3, 1, 2, 3};
    auto compressed = compress_9bit_input(input_numbers);
    __m128i *compressed_ptr = (__m128i *)compressed.get();
    std::vector<int> predicate_keys{1, 2, 3};
    const shared_fn fns[] = {shared_scan_128_sequential, shared_scan_128_sequential_unrolled, shared_scan_128_threaded,
                             shared_scan_128_standard,   shared_scan_128_standard_unrolled,   shared_scan_128_parallel,
                             shared_scan_256_sequential, shared_scan_256_standard,            shared_scan_256_parallel};
    for (shared_fn f : fns) {
        auto output_buffer_size = scan_output_buffer_size(input_numbers.size());
        std::vector<std::vector<uint8_t>> outputs(predicate_keys.size(), std::vector<uint8_t>(output_buffer_size));
        f(predicate_keys, compressed_ptr, input_numbers.size(), outputs);
        for (size_t key_id = 0; key_id < predicate_keys.size(); key_id++)
            for (size_t i = 0; i < input_numbers.size(); i++)
                REQUIRE(get_bit(outputs[key_id], i) == (input_numbers[i] == predicate_keys[key_id]));
    }
}

// test/simd_scan_tests.cpp:108-150 "Simple Shared SIMD Scan"
static void simple_shared_simd_scan()
{
    std::vector<uint16_t> input_numbers{1, 2, 3, 3, 2, 1, 1, 2, 3, 1, 2, 3};
    auto compressed = compress_9bit_input(input_numbers);
    __m128i *compressed_ptr = (__m128i *)compressed.get();
    std::vector<int> predicate_keys1{1};
    std::vector<int> predicate_keys2{2, 3};
    size_t output_buffer_size = scan_output_buffer_size(input_numbers.size());
    std::vector<uint8_t> outputs1(predicate_keys1.size() * output_buffer_size);
    std::vector<uint8_t> outputs2(predicate_keys2.size() * output_buffer_size);
    std::vector<uint8_t> compare_output(output_buffer_size);
    {
        shared_scan_128_linear_simple(predicate_keys1, compressed_ptr, input_numbers.size(), outputs1);
        int hits = scan_128(predicate_keys1[0], compressed_ptr, input_numbers.size(), compare_output);
        REQUIRE(hits == 4);
        REQUIRE(outputs1 == compare_output);
    }
    {
        shared_scan_128_linear_simple(predicate_keys2, compressed_ptr, input_numbers.size(), outputs2);
        int hits = scan_128(predicate_keys2[0], compressed_ptr, input_numbers.size(), compare_output);
        REQUIRE(hits == 4);
        for (size_t i = 0; i < compare_output.size(); ++i) REQUIRE(outputs2[i * 2] == compare_output[i]);
        hits = scan_128(predicate_keys2[1], compressed_ptr, input_numbers.size(), compare_output);
        REQUIRE(hits == 4);
        for (size_t i = 0; i < compare_output.size(); ++i) REQUIRE(outputs2[i * 2 + 1] == compare_output[i]);
        // linear_standard and the static template produce the same bytes
        std::vector<uint8_t> o3(outputs2.size()), o4(outputs2.size());
        shared_scan_128_linear_standard(predicate_keys2, compressed_ptr, input_numbers.size(), o3);
        shared_scan_128_linear_static<2>(predicate_keys2, compressed_ptr, input_numbers.size(), o4);
        REQUIRE(o3 == outputs2);
        REQUIRE(o4 == outputs2);
    }
    {
        // unsupported key count: diagnostic on cerr, output untouched (src/simd_scan_shared_linear.cpp:79-80)
        std::vector<int> three{1, 2, 3};
        std::vector<uint8_t> o(3 * output_buffer_size, 0x77);
        shared_scan_128_linear_simple(three, compressed_ptr, input_numbers.size(), o);
        for (uint8_t b : o) REQUIRE(b == 0x77);
    }
}

// test/util_tests.cpp
static void util_tests()
{
    REQUIRE(next_multiple(0, 8) == 0);
    REQUIRE(next_multiple(5, 8) == 8);
    REQUIRE(next_multiple(8, 8) == 8);
    REQUIRE(next_multiple(15, 8) == 16);
    REQUIRE(next_multiple(16, 8) == 16);
    REQUIRE(next_multiple(17, 8) == 24);
    REQUIRE(next_multiple(17, 9) == 18);
    std::vector<uint8_t> vec{5, 5};
    const bool expect[16] = {true, false, true, false, false, false, false, false, true, false, true, false, false, false, false, false};
    for (size_t i = 0; i < 16; i++) REQUIRE(get_bit(vec, i) == expect[i]);
}

// the shape of bench_scan (src/benchmark.cpp:165-194) at 1 MB: v = i % 5, key 3, check_scan_result
static void bench_shape()
{
    size_t data_size = 1 << 20, compression = 9;
    size_t input_size = data_size * 8 / compression;
    std::vector<uint16_t> input(input_size);
    for (size_t i = 0; i < input_size; i++) input[i] = (uint16_t)(i % 5);
    auto compressed = compress_9bit_input(input);
    std::vector<uint8_t> out(scan_output_buffer_size(input_size));
    int hits = scan_256_unrolled(3, (__m128i *)compressed.get(), input_size, out);
    size_t expect = 0;
    bool ok = true;
    for (size_t i = 0; i < input_size; i++) {
        expect += input[i] == 3;
        ok = ok && (get_bit(out, i) == (input[i] == 3));
    }
    REQUIRE(ok);
    REQUIRE((size_t)hits == expect);
    std::vector<int> dec(decompression_output_buffer_size(input_size) / sizeof(int));
    decompress_256_avx2((__m128i *)compressed.get(), input_size, dec.data());
    ok = true;
    for (size_t i = 0; i < input_size; i++) ok = ok && dec[i] == input[i];
    REQUIRE(ok);
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "--compile-check") return 0; // CPU: proves the header compiles and links
    try {
        compress_and_decompress();
        simd_scan();
        shared_simd_scan();
        simple_shared_simd_scan();
        util_tests();
        bench_shape();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s (%d assertions in 6 test cases)\n", g_failed ? "FAILED" : "All tests passed", g_checks);
    return g_failed ? 1 : 0;
}

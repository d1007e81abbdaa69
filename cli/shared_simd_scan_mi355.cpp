// shared_simd_scan_mi355 -- benchmark CLI with the reference's command line and output format
// (src/main.cpp:12-73, src/benchmark.cpp:14-36,73-108,165-194,263-306), driving the MI355X engine through the
// C ABI with columns resident in HBM.
//
//   ./shared_simd_scan_mi355 <MB|_> <reps|_> decompression|scan|sharedscan [P]
//
// Same inputs as the reference's harness (v = i & 511 / i % 5 / i % P % 512, key 3 / keys 0..P-1), synthesised on the
// device; same `* name: avg ms; [a, b, ...] ms` lines (fractional ms: the reference prints whole ms, which would be
// 0 here; scripts/prepare_shared_scan_results.py:14-20 copies the field verbatim); same self-checks
// (check_decompression_result / check_scan_result, src/benchmark.cpp:38-49,110-121) on the downloaded result.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "mi355_scan.h"

static const size_t default_data_size = 500u << 20;     // src/benchmark.hpp:4
static const size_t default_benchmark_repetitions = 5;  // src/benchmark.hpp:5

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != MI355_OK) {                                                   \
            std::cerr << #call << " failed: " << mi355_last_error() << std::endl; \
            std::exit(2);                                                        \
        }                                                                        \
    } while (0)

static void print_cmd_help()
{
    std::cout << "Format: ./shared_simd_scan_mi355 data_size repetitions bench_name [bench_args...]" << std::endl;
    std::cout << "data_size = _ (for default) | number (in megabytes)" << std::endl;
    std::cout << "repetitions = _ (for default) | number (for number of repetitions" << std::endl;
    std::cout << "bench_name = decompression | scan | sharedscan [predicate_count] " << std::endl;
}

// src/benchmark.cpp:14-36, with fractional milliseconds
static void print_numbers(const std::string &name, const std::vector<double> &ms)
{
    double sum = 0;
    for (double v : ms) sum += v;
    std::printf("* %s: %.4f ms; [", name.c_str(), sum / ms.size());
    for (size_t i = 0; i < ms.size(); i++) std::printf("%s%.4f", i ? ", " : "", ms[i]);
    std::printf("] ms\n");
}

template <typename F> static std::vector<double> time_reps(mi355_ctx *ctx, size_t reps, F &&launch)
{
    std::vector<double> ms(reps);
    launch(); // first-launch overheads (module load) are not part of any rep
    CHECK(mi355_ctx_synchronize(ctx));
    for (size_t i = 0; i < reps; i++) {
        auto t0 = std::chrono::steady_clock::now();
        launch();
        CHECK(mi355_ctx_synchronize(ctx));
        ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return ms;
}

static bool get_bit(const std::vector<uint8_t> &v, size_t i) { return (v[i / 8] & (1 << (i % 8))) > 0; }

static void bench_decompression(mi355_ctx *ctx, size_t data_size, size_t reps)
{
    const unsigned c = 9;
    const size_t n = data_size * 8 / c;
    void *packed, *out;
    CHECK(mi355_dev_alloc(ctx, mi355_compressed_buffer_size(c, n), &packed));
    CHECK(mi355_dev_alloc(ctx, mi355_decompression_output_buffer_size(n), &out));
    CHECK(mi355_generate_dev(ctx, MI355_GEN_INDEX, 0, n, c, 0, packed)); // input[i] = i & 511 (src/benchmark.cpp:81)
    std::cout << "## decompression benchmarks ##" << std::endl;
    std::cout << "compressed input: " << n << " (" << data_size << " bytes)" << std::endl;
    auto ms = time_reps(ctx, reps, [&] { CHECK(mi355_decompress_dev(ctx, packed, n, c, (int32_t *)out)); });
    print_numbers("mi355x hip (lds-dma + alignbit)", ms);
    std::vector<int32_t> host(n);
    CHECK(mi355_dev_download(ctx, host.data(), out, n * 4));
    for (size_t i = 0; i < n; i++)
        if (host[i] != (int32_t)(i & 511)) {
            std::cout << "first mismatch at index " << i << std::endl;
            break;
        }
    std::cout << "finished benchmark" << std::endl;
    mi355_dev_free(ctx, packed);
    mi355_dev_free(ctx, out);
}

static void bench_scan(mi355_ctx *ctx, size_t data_size, size_t reps)
{
    const unsigned c = 9;
    const size_t n = data_size * 8 / c;
    const int predicate_key = 3; // src/benchmark.cpp:150
    void *packed, *bitmap, *hits;
    CHECK(mi355_dev_alloc(ctx, mi355_compressed_buffer_size(c, n), &packed));
    CHECK(mi355_dev_alloc(ctx, mi355_scan_output_buffer_size(n), &bitmap));
    CHECK(mi355_dev_alloc(ctx, 8, &hits));
    CHECK(mi355_generate_dev(ctx, MI355_GEN_MOD, 0, n, c, 5, packed)); // input[i] = i % 5 (src/benchmark.cpp:173)
    std::cout << "## scan benchmarks ##" << std::endl;
    std::cout << "compressed input: " << n << " (" << data_size << " bytes)" << std::endl;
    auto ms = time_reps(ctx, reps, [&] { CHECK(mi355_scan_eq_dev(ctx, packed, n, c, predicate_key, bitmap, (uint64_t *)hits)); });
    print_numbers("mi355x hip (lds-dma + v_bfe/v_cmp/v_addc)", ms);
    std::vector<uint8_t> host((n + 7) / 8);
    uint64_t h = 0;
    CHECK(mi355_dev_download(ctx, host.data(), bitmap, host.size()));
    CHECK(mi355_dev_download(ctx, &h, hits, 8));
    size_t expect = 0;
    for (size_t i = 0; i < n; i++) {
        const bool e = (i % 5) == (size_t)predicate_key;
        expect += e;
        if (get_bit(host, i) != e) {
            std::cout << "first mismatch at index " << i << std::endl;
            break;
        }
    }
    if (h != expect) std::cout << "hit count mismatch: " << h << " != " << expect << std::endl;
    ms = time_reps(ctx, reps, [&] { CHECK(mi355_scan_range_dev(ctx, packed, n, c, 2, 3, bitmap, (uint64_t *)hits)); });
    print_numbers("mi355x hip, range 2..3", ms);
    std::cout << "finished benchmark" << std::endl;
    mi355_dev_free(ctx, packed);
    mi355_dev_free(ctx, bitmap);
    mi355_dev_free(ctx, hits);
}

static void bench_shared_scan(mi355_ctx *ctx, size_t data_size, size_t reps, int P, bool want_hits)
{
    const unsigned c = 9;
    const size_t n = data_size * 8 / c;
    if (P < 1 || P > 1024) {
        std::cerr << "predicate_count must be 1..1024" << std::endl;
        std::exit(1);
    }
    // (the reference's shared scans return no hit counts, src/simd_scan.hpp:102-120: none are requested unless the
    // optional 5th argument is "hits")
    std::vector<int32_t> keys(P);
    for (int i = 0; i < P; i++) keys[i] = i; // src/benchmark.cpp:205-209
    const size_t nb = (n + 7) / 8, stride = mi355_bitmap_stride(n);
    void *packed, *out, *hits;
    CHECK(mi355_dev_alloc(ctx, mi355_compressed_buffer_size(c, n), &packed));
    CHECK(mi355_dev_alloc(ctx, stride * P + 64, &out));
    CHECK(mi355_dev_alloc(ctx, 8 * P, &hits));
    // input[i] = i % P % 512 (src/benchmark.cpp:277): generate i % P, then the 9-bit mask of the packer does % 512
    CHECK(mi355_generate_dev(ctx, MI355_GEN_MOD, 0, n, c, (uint64_t)P, packed));
    std::cout << "## shared scan benchmarks ##" << std::endl;
    std::cout << "compressed input: " << n << " (" << data_size << " bytes)" << std::endl;
    std::cout << "predicate key count: " << P << std::endl;
    auto ms = time_reps(ctx, reps, [&] {
        CHECK(mi355_shared_scan_eq_dev(ctx, packed, n, c, keys.data(), P, MI355_LAYOUT_PER_PREDICATE, out, stride, want_hits ? (uint64_t *)hits : nullptr));
    });
    print_numbers("mi355x hip, standard", ms);
    // check_scan_result per predicate (the reference has these checks commented out, src/benchmark.cpp:227)
    std::vector<uint8_t> host(nb);
    for (int k = 0; k < P; k += (P > 16 ? P / 8 : 1)) {
        CHECK(mi355_dev_download(ctx, host.data(), (uint8_t *)out + (size_t)k * stride, nb));
        for (size_t i = 0; i < n; i++)
            if (get_bit(host, i) != (((i % P) & 511) == (size_t)k)) {
                std::cout << "first mismatch at index " << i << " (key " << k << ")" << std::endl;
                break;
            }
    }
    ms = time_reps(ctx, reps, [&] {
        CHECK(mi355_shared_scan_eq_dev(ctx, packed, n, c, keys.data(), P, MI355_LAYOUT_LINEAR, out, 0, want_hits ? (uint64_t *)hits : nullptr));
    });
    print_numbers("mi355x hip, linear, standard", ms);
    std::cout << "finished benchmark" << std::endl;
    mi355_dev_free(ctx, packed);
    mi355_dev_free(ctx, out);
    mi355_dev_free(ctx, hits);
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        print_cmd_help();
        return 1;
    }
    size_t data_size = default_data_size;
    if (strcmp(argv[1], "_") != 0) data_size = (size_t)atoi(argv[1]) << 20; // src/main.cpp:31
    size_t repetitions = default_benchmark_repetitions;
    if (strcmp(argv[2], "_") != 0) repetitions = atoi(argv[2]);
    mi355_ctx *ctx = nullptr;
    CHECK(mi355_ctx_create(0, nullptr, &ctx));
    const char *bench_name = argv[3];
    if (strcmp(bench_name, "decompression") == 0) {
        bench_decompression(ctx, data_size, repetitions);
    } else if (strcmp(bench_name, "scan") == 0) {
        bench_scan(ctx, data_size, repetitions);
    } else if (strcmp(bench_name, "sharedscan") == 0) {
        bench_shared_scan(ctx, data_size, repetitions, argc > 4 ? atoi(argv[4]) : 8, argc > 5 && strcmp(argv[5], "hits") == 0);
    } else if (strcmp(bench_name, "memory") == 0) {
        std::cout << "memory: host DRAM copy probes (src/benchmark_misc.cpp) are not part of the scan path" << std::endl;
    } else {
        print_cmd_help();
        return 1;
    }
    mi355_ctx_destroy(ctx);
    return 0;
}

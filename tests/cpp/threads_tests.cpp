// threads_tests.cpp -- the threading contract of the drop-in boundary (include/mi355_scan.h "Threading contract").
//
// The reference's functions are stateless and re-entrant; its own shared_scan_128_threaded calls scan_128 from an
// OpenMP loop over the keys (src/simd_scan_shared.cpp:25-32).  Through include/simd_scan.hpp every call passes
// ctx == NULL, so this checks that concurrent calls from several host threads never see each other's state:
//   1. 8 std::threads x scan_128 with 8 different keys on one packed column, repeated, every bitmap and every
//      returned hit count compared with a host-side evaluation of the predicate;
//   2. the same through the C ABI with ONE explicit context shared by all threads (calls serialise on its lock);
//   3. threads mixing shared scans of 16 keys (the key ring), scans and decompression.
// Built and run by tests/test_dropin_cpp.py (-m gpu).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "simd_scan.hpp"

static std::atomic<int> g_checks{0}, g_failed{0};
#define CHECK(cond)                                                                       \
    do {                                                                                  \
        g_checks++;                                                                       \
        if (!(cond)) {                                                                    \
            g_failed++;                                                                   \
            std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond); \
        }                                                                                 \
    } while (0)

static const int kThreads = 8;

struct Column {
    std::vector<uint16_t> values;
    std::unique_ptr<uint64_t[]> packed;
    size_t n;
};

static Column make_column(size_t n)
{
    Column col;
    col.n = n;
    col.values.resize(n);
    uint32_t x = 12345u;
    for (size_t i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        col.values[i] = (uint16_t)((x >> 16) % 23); // 23 distinct values: every key below has ~4 % selectivity
    }
    col.packed = compress_9bit_input(col.values);
    return col;
}

static bool bitmap_matches(const Column &col, int key, const std::vector<uint8_t> &out, size_t *expect_hits)
{
    size_t hits = 0;
    bool same = true;
    for (size_t i = 0; i < col.n; i++) {
        const bool want = col.values[i] == key;
        hits += want;
        same = same && (get_bit(out, i) == want);
    }
    *expect_hits = hits;
    return same;
}

// 1. drop-in names from 8 threads
static void scans_from_threads(const Column &col)
{
    std::vector<std::thread> pool;
    for (int t = 0; t < kThreads; t++) {
        pool.emplace_back([&col, t] {
            const int key = 2 * t + 1;
            for (int rep = 0; rep < 12; rep++) {
                std::vector<uint8_t> out(scan_output_buffer_size(col.n));
                const int hits = (rep & 1) ? scan_128(key, (__m128i *)col.packed.get(), col.n, out)
                                           : scan_256_unrolled(key, (__m128i *)col.packed.get(), col.n, out);
                size_t expect = 0;
                CHECK(bitmap_matches(col, key, out, &expect));
                CHECK((size_t)hits == expect);
            }
        });
    }
    for (auto &th : pool) th.join();
    // the reference's own threaded variant (one scan per key)
    std::vector<int> keys{0, 3, 5, 7, 11, 13, 17, 22};
    std::vector<std::vector<uint8_t>> outs(keys.size(), std::vector<uint8_t>(scan_output_buffer_size(col.n)));
    shared_scan_128_threaded(keys, (__m128i *)col.packed.get(), col.n, outs);
    for (size_t k = 0; k < keys.size(); k++) {
        size_t expect = 0;
        CHECK(bitmap_matches(col, keys[k], outs[k], &expect));
    }
}

// 2. one explicit context shared by all threads
static void shared_context(const Column &col)
{
    mi355_ctx *ctx = nullptr;
    CHECK(mi355_ctx_create(0, nullptr, &ctx) == MI355_OK);
    if (!ctx) return;
    std::vector<std::thread> pool;
    for (int t = 0; t < kThreads; t++) {
        pool.emplace_back([&col, ctx, t] {
            const int key = t + 3;
            for (int rep = 0; rep < 8; rep++) {
                std::vector<uint8_t> out(scan_output_buffer_size(col.n));
                uint64_t hits = ~0ull;
                CHECK(mi355_scan_eq(ctx, col.packed.get(), col.n, BITS_NEEDED, key, out.data(), &hits) == MI355_OK);
                size_t expect = 0;
                CHECK(bitmap_matches(col, key, out, &expect));
                CHECK(hits == expect);
            }
        });
    }
    for (auto &th : pool) th.join();
    CHECK(mi355_ctx_destroy(ctx) == MI355_OK);
}

// 3. mixed work: 16-key shared scans (key lists travel through the per-context ring), scans, decompression
static void mixed_work(const Column &col)
{
    std::vector<std::thread> pool;
    for (int t = 0; t < kThreads; t++) {
        pool.emplace_back([&col, t] {
            for (int rep = 0; rep < 4; rep++) {
                if (t % 3 == 0) {
                    std::vector<int> keys(16);
                    for (int k = 0; k < 16; k++) keys[k] = (k + t) % 23;
                    std::vector<std::vector<uint8_t>> outs(16, std::vector<uint8_t>(scan_output_buffer_size(col.n)));
                    shared_scan_128_standard(keys, (__m128i *)col.packed.get(), col.n, outs);
                    for (int k = 0; k < 16; k++) {
                        size_t expect = 0;
                        CHECK(bitmap_matches(col, keys[k], outs[k], &expect));
                    }
                } else if (t % 3 == 1) {
                    std::vector<int> dec(decompression_output_buffer_size(col.n) / sizeof(int));
                    decompress_128(((__m128i *)col.packed.get()), col.n, dec.data());
                    bool same = true;
                    for (size_t i = 0; i < col.n; i++) same = same && dec[i] == col.values[i];
                    CHECK(same);
                } else {
                    std::vector<uint8_t> out(scan_output_buffer_size(col.n));
                    const int hits = scan_128_unrolled(t, (__m128i *)col.packed.get(), col.n, out);
                    size_t expect = 0;
                    CHECK(bitmap_matches(col, t, out, &expect));
                    CHECK((size_t)hits == expect);
                }
            }
        });
    }
    for (auto &th : pool) th.join();
}

int main()
{
    try {
        const Column col = make_column(3 * 8192 * 40 + 77); // ~1e6 rows, ragged last tile
        scans_from_threads(col);
        shared_context(col);
        mixed_work(col);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s (%d checks, %d threads)\n", g_failed.load() ? "FAILED" : "All thread tests passed", g_checks.load(), kThreads);
    return g_failed.load() ? 1 : 0;
}

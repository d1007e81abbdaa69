// tests/cpp/next_tests.cpp -- a plain C++ host (g++, no HIP, no Python) driving the device-pointer entry points that go
// beyond the reference's surface: resident columns, count-only and fused-mask scans, two columns in one call, the fused
// selection vector and the values of another column at those rows, IN-lists, shared scans of 2 / 5 / 37 keys in both
// layouts, load-time tuning.  Expected results are
// computed here with scalar loops over the generator's closed form (v[i] = (first + i) % m), not by the oracle.
//   next_tests --compile-check   : exit 0 without touching a device (CPU build check)
#include "mi355_scan.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define REQUIRE(x)                                                                  \
    do {                                                                            \
        if (!(x)) {                                                                 \
            std::printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #x, mi355_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#define OK(call) REQUIRE((call) == MI355_OK)

static bool bit(const std::vector<uint8_t> &bm, uint64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }

int main(int argc, char **argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "--compile-check")) return 0;
    const uint64_t n = 3 * 8192 * 16 + 4099; // full chunks of the select kernel, full tiles, a ragged tail
    const unsigned c = 9;
    const uint64_t m1 = 7, m2 = 11, first = 1000;
    mi355_ctx *ctx = nullptr;
    OK(mi355_ctx_create(0, nullptr, &ctx));
    void *col1 = nullptr, *col2 = nullptr, *bm_dev = nullptr, *mask_dev = nullptr, *ids_dev = nullptr, *out_dev = nullptr;
    uint64_t *hits_dev = nullptr;
    OK(mi355_dev_alloc(ctx, mi355_compressed_buffer_size(c, n), &col1));
    OK(mi355_dev_alloc(ctx, mi355_compressed_buffer_size(c, n), &col2));
    OK(mi355_dev_alloc(ctx, mi355_scan_output_buffer_size(n), &bm_dev));
    OK(mi355_dev_alloc(ctx, mi355_scan_output_buffer_size(n), &mask_dev));
    OK(mi355_dev_alloc(ctx, n * sizeof(uint64_t), &ids_dev));
    OK(mi355_dev_alloc(ctx, 64 * sizeof(uint64_t), (void **)&hits_dev));
    OK(mi355_generate_dev(ctx, MI355_GEN_MOD, first, n, c, m1, col1));
    OK(mi355_generate_dev(ctx, MI355_GEN_MOD, first, n, c, m2, col2));
    auto v1 = [&](uint64_t i) { return (first + i) % m1; };
    auto v2 = [&](uint64_t i) { return (first + i) % m2; };
    const size_t nb = (n + 7) / 8;
    std::vector<uint8_t> bm(nb), mask(nb);
    uint64_t hits = 0;

    // tuning is a no-op below 5e7 rows and must say so
    OK(mi355_tune_dev(ctx, col1, n, c, MI355_TUNE_ALL));
    REQUIRE(mi355_tuned_blocks_per_cu(ctx, c, MI355_TUNE_SCAN, 0) == 0);
    REQUIRE(mi355_tune_dev(ctx, col1, n, c, 0) == MI355_E_INVALID);

    // mask = (v1 <= 2); count-only scan of v1 == 5; fused OR
    OK(mi355_scan_combine_dev(ctx, col1, n, c, MI355_CMP_LE, 2, 0, MI355_BITMAP_AND, nullptr, mask_dev, hits_dev));
    OK(mi355_dev_download(ctx, mask.data(), mask_dev, nb));
    OK(mi355_dev_download(ctx, &hits, hits_dev, sizeof hits));
    uint64_t want = 0;
    for (uint64_t i = 0; i < n; i++) {
        REQUIRE(bit(mask, i) == (v1(i) <= 2));
        want += v1(i) <= 2;
    }
    REQUIRE(hits == want);
    OK(mi355_scan_combine_dev(ctx, col1, n, c, MI355_CMP_EQ, 5, 0, MI355_BITMAP_AND, nullptr, nullptr, hits_dev)); // count only
    OK(mi355_dev_download(ctx, &hits, hits_dev, sizeof hits));
    want = 0;
    for (uint64_t i = 0; i < n; i++) want += v1(i) == 5;
    REQUIRE(hits == want);
    OK(mi355_scan_combine_dev(ctx, col1, n, c, MI355_CMP_EQ, 5, 0, MI355_BITMAP_OR, mask_dev, bm_dev, hits_dev));
    OK(mi355_dev_download(ctx, bm.data(), bm_dev, nb));
    for (uint64_t i = 0; i < n; i++) REQUIRE(bit(bm, i) == (v1(i) <= 2 || v1(i) == 5));

    // two columns in one call: v1 BETWEEN 2 AND 4, AND NOT v2 >= 6
    OK(mi355_scan2_dev(ctx, col1, c, MI355_CMP_BETWEEN, 2, 4, col2, c, MI355_CMP_GE, 6, 0, n, MI355_BITMAP_ANDNOT, bm_dev, hits_dev));
    OK(mi355_dev_download(ctx, bm.data(), bm_dev, nb));
    OK(mi355_dev_download(ctx, &hits, hits_dev, sizeof hits));
    want = 0;
    for (uint64_t i = 0; i < n; i++) {
        const bool e = (v1(i) >= 2 && v1(i) <= 4) && !(v2(i) >= 6);
        REQUIRE(bit(bm, i) == e);
        want += e;
    }
    REQUIRE(hits == want);

    // IN-list
    const int32_t in_keys[3] = {1, 6, 300};
    OK(mi355_scan_in_dev(ctx, col1, n, c, in_keys, 3, 0, nullptr, bm_dev, hits_dev));
    OK(mi355_dev_download(ctx, bm.data(), bm_dev, nb));
    for (uint64_t i = 0; i < n; i++) REQUIRE(bit(bm, i) == (v1(i) == 1 || v1(i) == 6));

    // fused selection: rows with v2 == 3 AND the mask, global row ids
    OK(mi355_scan_select_dev(ctx, col2, n, c, MI355_CMP_EQ, 3, 0, MI355_BITMAP_AND, mask_dev, /*first_row*/ first, (uint64_t *)ids_dev, n, hits_dev));
    OK(mi355_dev_download(ctx, &hits, hits_dev, sizeof hits));
    std::vector<uint64_t> expect_ids;
    for (uint64_t i = 0; i < n; i++)
        if (v2(i) == 3 && v1(i) <= 2) expect_ids.push_back(first + i);
    REQUIRE(hits == expect_ids.size());
    std::vector<uint64_t> ids(expect_ids.size());
    OK(mi355_dev_download(ctx, ids.data(), ids_dev, ids.size() * sizeof(uint64_t)));
    REQUIRE(ids == expect_ids);
    // ... and the values of the OTHER column at those rows ("take"), the count still on the device
    {
        void *taken_dev = nullptr;
        OK(mi355_dev_alloc(ctx, n * sizeof(int32_t), &taken_dev));
        OK(mi355_gather_dev(ctx, col1, n, c, first, (const uint64_t *)ids_dev, hits_dev, n, (int32_t *)taken_dev));
        std::vector<int32_t> taken(expect_ids.size());
        OK(mi355_dev_download(ctx, taken.data(), taken_dev, taken.size() * sizeof(int32_t)));
        for (size_t k = 0; k < taken.size(); k++) REQUIRE((uint64_t)taken[k] == v1(expect_ids[k] - first));
        OK(mi355_dev_free(ctx, taken_dev));
    }

    // aggregates of column 2 over the rows of the mask (v1 <= 2), and over all rows
    {
        uint64_t agg[4], sum = 0, cnt = 0, mn = ~0ull, mx = 0;
        OK(mi355_aggregate_dev(ctx, col2, n, c, mask_dev, hits_dev));
        OK(mi355_dev_download(ctx, agg, hits_dev, sizeof agg));
        for (uint64_t i = 0; i < n; i++)
            if (v1(i) <= 2) {
                sum += v2(i), cnt++;
                if (v2(i) < mn) mn = v2(i);
                if (v2(i) > mx) mx = v2(i);
            }
        REQUIRE(agg[0] == sum && agg[1] == cnt && agg[2] == mn && agg[3] == mx);
        OK(mi355_aggregate_dev(ctx, col2, n, c, nullptr, hits_dev));
        OK(mi355_dev_download(ctx, agg, hits_dev, sizeof agg));
        sum = 0;
        for (uint64_t i = 0; i < n; i++) sum += v2(i);
        REQUIRE(agg[0] == sum && agg[1] == n && agg[2] == 0 && agg[3] == m2 - 1);
    }

    // shared scans: 2 keys (compare kernel), 5 keys (one-pass LUT, packed rows), 37 keys (32 keys per lookup), both layouts
    for (unsigned P : {2u, 5u, 37u}) {
        std::vector<int32_t> keys(P);
        for (unsigned k = 0; k < P; k++) keys[k] = (int32_t)((3 * k + 1) % 9); // 7, 8 match nothing; duplicates from k = 9 on
        const size_t stride = mi355_bitmap_stride(n);
        if (out_dev) OK(mi355_dev_free(ctx, out_dev));
        OK(mi355_dev_alloc(ctx, stride * P + 64, &out_dev));
        for (int layout : {MI355_LAYOUT_PER_PREDICATE, MI355_LAYOUT_LINEAR}) {
            OK(mi355_dev_memset(ctx, out_dev, 0xEE, stride * P + 64));
            OK(mi355_shared_scan_eq_dev(ctx, col1, n, c, keys.data(), P, layout, out_dev, layout == MI355_LAYOUT_PER_PREDICATE ? stride : 0, hits_dev));
            std::vector<uint8_t> out(stride * P + 64);
            std::vector<uint64_t> h(P);
            OK(mi355_dev_download(ctx, out.data(), out_dev, out.size()));
            OK(mi355_dev_download(ctx, h.data(), hits_dev, P * sizeof(uint64_t)));
            for (unsigned k = 0; k < P; k++) {
                uint64_t cnt = 0;
                for (uint64_t i = 0; i < n; i++) {
                    const bool e = v1(i) == (uint64_t)keys[k];
                    const uint8_t byte = layout == MI355_LAYOUT_PER_PREDICATE ? out[k * stride + (i >> 3)] : out[(i >> 3) * P + k];
                    REQUIRE((((byte >> (i & 7)) & 1) != 0) == e);
                    cnt += e;
                }
                REQUIRE(h[k] == cnt);
            }
            // nothing behind the output is touched (linear: nb * P bytes; per-predicate: the last bitmap's nb bytes)
            const size_t end = layout == MI355_LAYOUT_PER_PREDICATE ? (P - 1) * stride + nb : nb * P;
            for (size_t b = end; b < out.size(); b++) REQUIRE(out[b] == 0xEE);
        }
    }

    OK(mi355_dev_free(ctx, out_dev));
    OK(mi355_dev_free(ctx, ids_dev));
    OK(mi355_dev_free(ctx, mask_dev));
    OK(mi355_dev_free(ctx, bm_dev));
    OK(mi355_dev_free(ctx, col2));
    OK(mi355_dev_free(ctx, col1));
    OK(mi355_dev_free(ctx, hits_dev));
    OK(mi355_ctx_destroy(ctx));
    std::printf("All next-row tests passed\n");
    return 0;
}

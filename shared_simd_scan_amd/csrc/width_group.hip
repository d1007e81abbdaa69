// width_group.hip -- instantiates the width-templated kernels for widths MI355_WLO..MI355_WHI and
// exports one launcher per group.  Compiled 8 times (4 widths each) so the build parallelises.
#include <atomic>
#include <type_traits>

#include "dispatch.hpp"
#include "kernels.hpp"

#ifndef MI355_WLO
#error "compile with -DMI355_WLO=<first width> -DMI355_WHI=<last width> -DMI355_GROUP=<index>"
#endif

namespace mi355 {

namespace {

template <typename K> int blocks_per_cu(K kernel)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlockThreads, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

// Kernels that take their lookup tables as dynamic LDS may need more than the default 64 KiB: raise the limit once
// per kernel AND device (the attribute is per device; a process may hold contexts on several GPUs).
template <auto Kernel> void allow_dynamic_lds(int max_bytes, int device)
{
    static std::atomic<unsigned long long> done{0};
    const unsigned long long bit = 1ull << (device & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
        done.fetch_or(bit, std::memory_order_release);
    }
}

inline int cap_bpc(int bpc, const LaunchReq &r)
{
    return (r.max_blocks_per_cu > 0 && r.max_blocks_per_cu < bpc) ? r.max_blocks_per_cu : bpc;
}

// Resident blocks per CU for the streaming scans.  Measured on MI355X (tools/sweep.py, 1e9 rows): the scans
// run fastest with ~36-48 KiB of LDS-DMA in flight per CU -- one 4-wave block at c=9 (4 x 9 KiB tiles) --
// and lose 3-6 % at the occupancy limit (more concurrent streams, same bytes).  So: the number of blocks
// whose tiles add up to ~40 KiB, at least 1, at most what the occupancy query admits.
inline int scan_bpc(int occ_bpc, int tile_bytes, const LaunchReq &r)
{
    if (r.max_blocks_per_cu > 0) return r.max_blocks_per_cu < occ_bpc ? r.max_blocks_per_cu : occ_bpc;
    int want = (40 * 1024 + 2 * tile_bytes) / (kWavesPerBlock * tile_bytes); // rounded
    if (want < 1) want = 1;
    if (want > 4) want = 4; // c = 1, 2 (1-2 KiB tiles): four blocks per CU beat eight by 20 % / 6 % (launches back to back)
    return want < occ_bpc ? want : occ_bpc;
}

// static LDS of the multi-pass LUT kernel: four tiles, the per-block hit counters, ticket word and slack
template <int C, int VPL> constexpr size_t lut_static_lds()
{
    // + the hit-count histogram
    return 4 * ScanGeom<C, VPL>::LDS_BYTES + kMaxKeys * 4 + 512 + (C <= 12 ? (size_t)(4u << C) : 16);
}

// the 32-keys-per-lookup kernel needs ceil(P/32) tables next to that in the CU's 160 KiB of LDS
template <int C, int VPL> bool lut_fits(uint32_t P)
{
    const size_t tables = (size_t)((P + 31) / 32) * WideLutGeom<C>::TABLE_BYTES;
    return tables + lut_static_lds<C, VPL>() <= 160 * 1024;
}

// ... and the byte-entry multi-pass kernel ceil(P/8) tables
template <int C, int VPL> bool lut8_fits(uint32_t P)
{
    const size_t tables = ((size_t)((P + 7) / 8) * LutGeom<C, true>::TABLE_BYTES + 15) / 16 * 16;
    return tables + lut_static_lds<C, VPL>() <= 160 * 1024;
}

// tiles per store burst of scan_burst_kernel at width C.  Same-process A/B on four MI355X boxes (tools/ab_opts.py
// --opt scan_burst=..., 1e9 rows, launches back to back; profiles/r02_burst_*.txt): K = 4 is 0-4 % faster than K = 1 at
// c = 9 (never slower), +1-2 % at c = 5, 6, 0-2 % at c = 10..16; it LOSES at c = 7 (0.171 against 0.149 ms) and c = 8
// (0.184 against 0.177), is neutral at c <= 4, and -2 % at 64 values per lane (c >= 17).
constexpr int burst_k(int c) { return (c == 5 || c == 6 || (c >= 9 && c <= 16)) ? 4 : 1; }

template <int C, int MODE> void launch_scan(const LaunchReq &r)
{
    constexpr int VPL = scan_vpl(C, MODE);
    using G = ScanGeom<C, VPL>;
    const uint64_t ntiles = (r.scan.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
    // dma_aux: cache policy of the HBM->LDS stream; 2 (non-temporal: the column is read once) is the default.
    // Bitmap stores, measured with launches back to back (bench.py --store-policy, same box, 1e9 x 9 bit unless noted):
    // write-through (sc1) 0.201 ms, plain 0.207, non-temporal 0.216 -- dirty bitmap lines do not pile up in L2 to be
    // written back under the next launch's read stream; c = 21: 0.438 / 0.467 / 0.457; c = 5: 0.127 / 0.129 / 0.136.
    // Bitmaps far beyond the 256 MiB Infinity Cache prefer non-temporal stores: 4e9 rows sc1 0.82 ms / nt 0.85,
    // 8e9 rows (1 GB of bitmap) 1.74 / 1.72.
    const int policy = r.scan_nt_stores < 0 ? (r.scan.n / 8 > (768ull << 20) ? 1 : 2) : r.scan_nt_stores; // 0 plain, 1 nt, 2 sc1
    auto go = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        static const int bpcK = blocks_per_cu(scan_burst_kernel<C, MODE, 34, VPL, K>);
        const dim3 grid(grid_for((ntiles + K - 1) / K, scan_bpc(bpcK, G::TILE_BYTES, r), r.num_cus));
        if (r.dma_aux == 0)
            hipLaunchKernelGGL((scan_burst_kernel<C, MODE, 0, VPL, K>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
        else if (policy == 1)
            hipLaunchKernelGGL((scan_burst_kernel<C, MODE, 18, VPL, K>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
        else if (policy == 2)
            hipLaunchKernelGGL((scan_burst_kernel<C, MODE, 34, VPL, K>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
        else
            hipLaunchKernelGGL((scan_burst_kernel<C, MODE, 2, VPL, K>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
    };
    // "scan_burst" option: 0 = the width's default, 1 = one tile per burst (A/B)
    if (burst_k(C) > 1 && r.scan_burst != 1)
        go(std::integral_constant<int, burst_k(C)>{});
    else
        go(std::integral_constant<int, 1>{});
}

// shared scan, P <= 8: LDS lookup table, one pass, deferred stores
template <int C, int VPL> void launch_lut8(const LaunchReq &r, uint32_t P, bool linear)
{
    using G = ScanGeom<C, VPL>;
    const uint64_t ntiles = (r.scan.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
    // measured (tools/tune_scan.hip, 1e9 x 9 bit, P = 8): one block per CU 0.41 ms, two 0.46, three 0.49
    // (tools/sweep.py: c = 5, 2.5 KiB tiles, is the exception -- two blocks 0.30 ms against 0.37)
    // The linear layout (word-wise transposition + LDS row stage) wants a second block per CU on random data:
    // launches back to back, 1e9 x 9 bit, P = 8, random column 0.36-0.37 ms against 0.417 with one block; equal on
    // the i % 8 column; per-predicate prefers one (0.35-0.38 against 0.37-0.40).
    auto lut_bpc = [&](int occ) {
        const int want = r.max_blocks_per_cu > 0 ? r.max_blocks_per_cu : ((G::TILE_BYTES < 4096 || linear) ? 2 : 1);
        return want < occ ? want : occ;
    };
    static const int bpc_lin = blocks_per_cu(shared_lut_kernel<C, 2, VPL, 1, false>);
    static const int bpc_pp = blocks_per_cu(shared_lut_kernel<C, 2, VPL, 0, false>);
    const dim3 grid(grid_for(ntiles, lut_bpc(linear ? bpc_lin : bpc_pp), r.num_cus));
    // one pass: write-through below 768 MiB of output, non-temporal beyond, as in launch_scan (launches back to
    // back, P = 8: 1e8 rows sc1 0.046 ms / plain 0.047 / nt 0.049; 1e9 rows nt 0.353-0.383 / sc1 0.347-0.393 / plain 0.40)
    const int spol = r.scan_nt_stores < 0 ? ((r.scan.n / 8) * P > (768ull << 20) ? 1 : 2) : r.scan_nt_stores; // 0 plain, 1 nt, 2 sc1
    if (linear && spol == 1)
        hipLaunchKernelGGL((shared_lut_kernel<C, 18, VPL, 1, false>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
    else if (linear && spol == 2)
        hipLaunchKernelGGL((shared_lut_kernel<C, 34, VPL, 1, false>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
    else if (linear)
        hipLaunchKernelGGL((shared_lut_kernel<C, 2, VPL, 1, false>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
    else if (spol == 1)
        hipLaunchKernelGGL((shared_lut_kernel<C, 18, VPL, 0, false>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
    else if (spol == 2)
        hipLaunchKernelGGL((shared_lut_kernel<C, 34, VPL, 0, false>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
    else
        hipLaunchKernelGGL((shared_lut_kernel<C, 2, VPL, 0, false>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
}

// which shared scans of <= 8 keys run with 128 values per lane by default (A/B on MI355X: see DESIGN.md section 3.1b)
inline bool shared_lut_prefers_vpl128(int c, uint32_t P, bool linear)
{
    // launches back to back, 1e9 x 9 bit, same box (tools/sweep_p.py --vpl 64,128): per-predicate P = 2 0.296 -> 0.265 ms,
    // P = 4 0.301 -> 0.272 (16-byte stores, 1 KiB per wave and key), P = 8 equal (0.349); linear LOSES (P = 2 0.244 ->
    // 0.367, P = 8 0.365 -> 0.470: twice the row stage, one wave per SIMD)
    (void)c;
    return !linear && P <= 4;
}

// linear rows: does the short last table (R = P mod 32 keys behind Tf = P / 32 full ones) ride on the lane of the row's last full
// piece (shared_linear2_kernel's attached mode) instead of getting a lane of its own (shared_linear_kernel)?  Attached, a
// wave-step covers 64 / Tf rows instead of 64 / (Tf + 1) and pays the short piece's instructions with 1 / Tf of the lanes in
// use.  Measured at every Tf = 2 .. 8, 12, 15 and R = 1 .. 31 (2.5e8 x 9 bit, profiles/r03_linear_attach_ab.txt): where the row
// gain is >= 1.19 x it wins at (almost) every R -- 1.0 - 1.4 x; where it is 1.10 .. 1.18 x only for R <= 8; where the row
// count does not change (Tf = 11, 13 .. 15, ...) it loses 10 - 20 %.  (flags bit 16: never, bit 18: always, for A/B)
static inline bool attach_short(unsigned P, unsigned flags, bool hits)
{
    const unsigned Tf = P / 32, R = P % 32;
    if (Tf < 2 || R == 0) return false;
    if (flags & 0x40000u) return true;
    if (flags & 0x10000u) return false;
    const unsigned rows_attached = 64 / Tf, rows_own_lane = 64 / (Tf + 1);
    if (rows_attached * 100 >= rows_own_lane * 119) return !(Tf == 3 && R > 24);
    // (Tf = 7 with hit counts: the old mapping's eight lanes per row count one value each with a single LDS atomic)
    if (rows_attached * 100 >= rows_own_lane * 110) return R <= 8 && !(Tf == 7 && hits);
    return false;
}

template <int C> hipError_t launch_width(const LaunchReq &r)
{
    switch (r.op) {
    case kOpScanEq: launch_scan<C, kModeEq>(r); break;
    case kOpScanRange: launch_scan<C, kModeRange>(r); break;
    case kOpSharedScan: {
        constexpr int VPL = scan_vpl(C, kModeShared);
        using G = ScanGeom<C, VPL>;
        const uint64_t ntiles = (r.scan.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        const uint32_t P = r.scan.nkeys;
        const bool linear = r.scan.layout != 0;
        auto lut_bpc = [&](int occ) { // multi-pass LUT kernel (see launch_lut8 for the one-pass kernels)
            const int want = r.max_blocks_per_cu > 0 ? r.max_blocks_per_cu : ((G::TILE_BYTES < 4096 || linear) ? 2 : 1);
            return want < occ ? want : occ;
        };
        // result stores: non-temporal unless the P bitmaps together are small.  Measured (tools/sweep.py --nts 0,1, P = 8,
        // c = 9): 1e8 rows (100 MB of bitmaps) 0.0540 -> 0.0525 ms, 5e8 0.220 -> 0.214, 1e9 0.409 -> 0.372; c = 17: -1..-4 %.
        // Unlike the single bitmap of launch_scan, these outputs gain nothing from staying in the Infinity Cache.
        const bool nt_stores = r.scan_nt_stores < 0 ? (r.scan.n / 8) * P > (64ull << 20) : r.scan_nt_stores != 0;
        // linear rows of 9 .. 1024 keys: lanes in memory order (shared_linear_kernel).  It needs two blocks per CU to hide its
        // lookups: tables too big for that -- P = 1024 at c <= 10 -- stay on the per-group kernel unless hit counts are
        // wanted (2.5e8 x 9 bit, P = 1024: 13.5 against 10.2 ms without, 15.6 against 17.8 with).  (flags bit 1: the older kernels, A/B)
        // Digit-table widths (c > 10) leave it to the per-group kernel beyond 320 keys (beyond 160 without hit counts at c > 16):
        // every lane of a row decodes the row again and looks up ceil(c/8) digits, and the tables leave room for two blocks
        // per CU only (2.5e8 rows, with / without hit counts, TB/s, shared_linear_kernel against the per-group kernel: c = 13,
        // P = 300: 2.6 / 2.9 against 2.2 / 2.2, P = 400: 2.1 / 2.3 against 2.4 / 2.7, P = 600: 1.7 / 1.8 against 2.4 / 2.6;
        // c = 17, P = 150: 2.6 / 2.8 against 1.4 / 1.8, P = 200: 2.6 / 2.8 against 1.5 / 3.0, P = 300: 2.1 / 2.2 against 1.5 / 2.5;
        // c = 9, P = 300: 3.6 / 4.1 against 2.2 / 2.3).
        const bool lin_pays = C <= 10 || (C <= 16 ? P <= 320 : P <= (r.scan.hits ? 320u : 160u)) || (r.scan.flags & 128u); // (bit 7: always, A/B)
        const bool lin_rows = linear && P > 8 && lut_fits<C, VPL>(P) && !(r.scan.flags & 2u) && lin_pays &&
                              (2 * ((size_t)((P + 31) / 32) * WideLutGeom<C>::TABLE_BYTES + lut_static_lds<C, VPL>()) <= 160 * 1024 ||
                               (r.scan.hits && WideLutGeom<C>::SINGLE));
        // Per-predicate bitmaps at the widths of three or four table digits with few keys: round 2 sent them to the compare chain
        // (16 v_cmp + v_addc per value beat three or four lookups + ANDs per value in shared_wide2_kernel at ONE wave per SIMD).
        // shared_wide3_kernel turns that around (2.5e8 rows, with hit counts, TB/s, tables against chain: c = 17, P = 16: 4.39
        // against 2.92; c = 21: 3.61 against 3.27; c = 29: 4.77 against 3.56, P = 24: 4.74), so the chain keeps only the key
        // counts whose tables do not fit (flags bit 10: round 2's rule, for A/B).
        const bool chain_pays = !linear && C >= 17 && P <= (C >= 25 ? 24u : 16u) && (r.scan.flags & 0x400u);
        if (r.choice_out) { // introspection (mi355_shared_scan_kernel): which kernel family would run, nothing is launched
            *r.choice_out = (P == 2 && !(r.scan.flags & 32u)) ? 5 : P <= 8 ? 0 : lin_rows ? 4 : (linear && !r.scan.hits && P < 192 && lut8_fits<C, VPL>(P)) ? 1 : (lut_fits<C, VPL>(P) && !chain_pays) ? 2 : 3;
            break;
        }
        if (P == 2 && !(r.scan.flags & 32u)) { // two keys: the equality scan's decode twice (flags bit 5: the LUT kernel, A/B)
            // per-predicate: the scan's geometry (128 values per lane at c <= 16: a 16-byte store per key and lane); linear:
            // 64 values per lane, so that the lane's 16 row bytes are ONE store and an instruction writes 1 KiB of whole
            // lines (with 128 the lane's 32 bytes left as two instructions of half lines: write-through stores turned
            // them into partial-line writes -- c = 12: 4.7 TB/s against 5.5 for the LUT kernel it was to replace)
            auto go = [&](auto vpl_c) {
                constexpr int PVPL = decltype(vpl_c)::value;
                using PG = ScanGeom<C, PVPL>;
                const uint64_t ptiles = (r.scan.n + PG::TILE_VALUES - 1) / PG::TILE_VALUES;
                static const int pbpc = blocks_per_cu(shared_pair_kernel<C, 34, PVPL>);
                const dim3 pgrid(grid_for(ptiles, scan_bpc(pbpc, PG::TILE_BYTES, r), r.num_cus));
                // result stores as in launch_lut8: write-through below 768 MiB of output, non-temporal beyond
                const int spol = r.scan_nt_stores < 0 ? ((r.scan.n / 8) * P > (768ull << 20) ? 1 : 2) : r.scan_nt_stores;
                if (spol == 1)
                    hipLaunchKernelGGL((shared_pair_kernel<C, 18, PVPL>), pgrid, dim3(kBlockThreads), 0, r.stream, r.scan);
                else
                    hipLaunchKernelGGL((shared_pair_kernel<C, 34, PVPL>), pgrid, dim3(kBlockThreads), 0, r.stream, r.scan);
            };
            if (linear)
                go(std::integral_constant<int, 64>{});
            else
                go(std::integral_constant<int, scan_vpl(C, kModeEq)>{});
        } else if (P <= 8) { // LDS lookup table, one pass, deferred stores
            // 128 values per lane (16-byte result stores, 1 KiB per wave and key) where the tile, the table and the linear
            // stage fit in LDS and the registers hold 2 x 32 result dwords: c <= 12
            if constexpr (C <= 12) {
                if (r.shared_vpl == 128 || (r.shared_vpl == 0 && shared_lut_prefers_vpl128(C, P, linear))) {
                    launch_lut8<C, 128>(r, P, linear);
                    break;
                }
            }
            launch_lut8<C, 64>(r, P, linear);
        } else if (linear && P >= 32 && P <= 40 && !(r.scan.flags & 0x2000u) &&
                   2 * ((size_t)((P + 31) / 32) * WideLutGeom<C, false>::TABLE_BYTES + 4 * ScanGeom<C, 64>::LDS_BYTES + 33 * 1024) <= 160 * 1024) {
            // linear rows of 32 .. 40 keys: the per-predicate machinery + an LDS stage (shared_linear3_kernel; flags bit 13: the
            // row-per-lane kernels below, for A/B).  Hit counts in registers: one round (P <= 32) or two packed.  Where it pays
            // (2.5e8 rows, TB/s with hit counts, against the row-per-lane kernels on the same box): c = 9, P = 32 / 33 / 40:
            // 4.67 / 3.54 / 3.52 against 4.05 / 3.29 / 3.24; c = 5, P = 32: 4.49 against 3.28; c = 12: 4.54 against 4.06; c = 17:
            // 4.64 against 4.32.  Where it does not: fewer keys (no VALU to save: P = 9 2.55 against 2.92, P = 16 3.94 against
            // 4.33, P = 24 / 31 equal), a long second round (its 32-byte pieces complete the first round's half-written lines a
            // whole round later: P = 48 3.01 against 3.41, P = 64 2.29 against 3.94), and widths whose tiles leave room for one
            // block per CU only (c = 25, P = 32: 2.95 against 4.40).
            constexpr bool kBigWidth = (C >= 17 && C <= 20) || (C >= 25 && C <= 30);
            const int rc = r.scan.hits ? (P <= 32 ? 1 : 2) : 0;
            const size_t fixed = 4 * ScanGeom<C, 64>::LDS_BYTES + 33 * 1024;
            bool big = false;
            if constexpr (kBigWidth)
                big = !(r.scan.flags & 0x200u) && 2 * ((size_t)((P + 31) / 32) * WideLutGeom<C, true>::TABLE_BYTES + fixed) <= 160 * 1024;
            auto go3 = [&](auto rc_c, auto big_c) {
                constexpr int RC = decltype(rc_c)::value;
                constexpr bool BIG = decltype(big_c)::value;
                const size_t bdyn = (size_t)((P + 31) / 32) * WideLutGeom<C, BIG>::TABLE_BYTES;
                allow_dynamic_lds<shared_linear3_kernel<C, 2, RC, BIG>>((int)(160 * 1024 - fixed), r.device);
                int fit = (int)((160 * 1024) / (bdyn + fixed));
                fit = fit > 2 ? 2 : (fit < 1 ? 1 : fit);
                const dim3 g3(grid_for(ntiles, r.max_blocks_per_cu > 0 && r.max_blocks_per_cu < fit ? r.max_blocks_per_cu : fit, r.num_cus));
                hipLaunchKernelGGL((shared_linear3_kernel<C, 2, RC, BIG>), g3, dim3(kBlockThreads), bdyn, r.stream, r.scan);
            };
            auto with_big3 = [&](auto rc_c) {
                if constexpr (kBigWidth) {
                    if (big) {
                        go3(rc_c, std::true_type{});
                        return;
                    }
                }
                go3(rc_c, std::false_type{});
            };
            if (rc == 1)
                with_big3(std::integral_constant<int, 1>{});
            else if (rc == 2)
                with_big3(std::integral_constant<int, 2>{});
            else
                with_big3(std::integral_constant<int, 0>{});
        } else if (linear && !r.scan.hits && P < 192 && lut8_fits<C, VPL>(P) && !lin_rows) {
            // linear rows of fewer than ~200 keys without hit counts: byte-entry tables, 16 output bytes per round
            // (measured, tools/sweep_p.py, 2.5e8 x 9 bit: P = 16 / 32 / 64 / 128 0.21 / 0.43 / 0.72 / 1.45 ms against
            // 0.41 / 0.58 / 0.91 / 1.50 for the dword-entry kernel, which wins from P = 256: 2.80 against 3.16 ms)
            const size_t dyn = ((size_t)((P + 7) / 8) * LutGeom<C, true>::TABLE_BYTES + 15) / 16 * 16;
            allow_dynamic_lds<shared_lut_kernel<C, 2, VPL, 1, true>>((int)(160 * 1024 - lut_static_lds<C, VPL>()), r.device);
            hipLaunchKernelGGL((shared_lut_kernel<C, 2, VPL, 1, true>), dim3(grid_for(ntiles, lut_bpc(8), r.num_cus)),
                               dim3(kBlockThreads), dyn, r.stream, r.scan);
        } else if (lut_fits<C, VPL>(P) && !(r.scan.flags & 64u) && !chain_pays) { // one dword-entry lookup table per 32 keys, in dynamic LDS
            // (flags bit 6: the compare chain, for A/B)
            const size_t dyn = (size_t)((P + 31) / 32) * WideLutGeom<C>::TABLE_BYTES;
            const int max_dyn = (int)(160 * 1024 - lut_static_lds<C, VPL>());
            allow_dynamic_lds<shared_wide_kernel<C, 2, VPL, 1>>(max_dyn, r.device);
            allow_dynamic_lds<shared_wide_kernel<C, 2, VPL, 0>>(max_dyn, r.device);
            allow_dynamic_lds<shared_wide_kernel<C, 18, VPL, 0>>(max_dyn, r.device);
            const int want = r.max_blocks_per_cu > 0 ? r.max_blocks_per_cu : 2;
            const dim3 grid(grid_for(ntiles, want, r.num_cus));
            if (lin_rows) {
                allow_dynamic_lds<shared_linear_kernel<C, 2, 1>>(max_dyn, r.device);
                allow_dynamic_lds<shared_linear_kernel<C, 2, 2>>(max_dyn, r.device);
                const dim3 lgrid(grid_for(ntiles, r.max_blocks_per_cu > 0 ? r.max_blocks_per_cu : 4, r.num_cus));
                // P = 16: two rows per 32-byte piece only with the digit tables (c > 10: 4.0 / 4.8 TB/s against 3.2 / 4.2 with one
                // row per piece at c = 12); at c <= 10 one row per piece wins (c = 5: 3.0 / 4.7 against 2.0 / 4.1, c = 9: 3.9 /
                // 4.8 against 3.5 / 4.9 with / without hit counts).  (flags bit 4: one row per piece everywhere, for A/B)
                // everything else: full tables in memory order, the short last table on its own (shared_linear2_kernel; flags
                // bit 8: round 2's kernel, which gives the short table a whole lane per row, for A/B)
                if (P == 16 && C > 10 && !(r.scan.flags & 16u))
                    hipLaunchKernelGGL((shared_linear_kernel<C, 2, 2>), lgrid, dim3(kBlockThreads), dyn, r.stream, r.scan);
                // shared_linear2_kernel (the short last table on the full piece's lane / in steps of its own) is the product only for
                // rows below 32 keys without hit counts (2.5e8 x 9 bit, same box: P = 12: 4.09 against 3.77 TB/s; with hit counts
                // 2.87 against 3.43).  For rows of 33 .. 63 keys it beat round 2's kernel (P = 33: 3.10 against 2.59) until that kernel
                // learnt to write such rows through an aligned LDS image (P = 47 / 52 / 56: 3.02 / 3.30 / 3.46 against 2.82 / 2.84 /
                // 2.85; flags bit 15 brings it back for A/B), and its short-table steps LOSE behind two or more full tables -- P = 100:
                // 2.26 against 3.18, P = 300: 2.45 against 3.58: a step that writes 4 bytes of each of 64 rows is 64 partial-line
                // transactions, where the old mapping's short lane sits in the same store instruction as its row's full pieces.
                // Rows of 65 and more keys with a short last table: attach_short() above decides between the two mappings.
                else if ((r.scan.flags & 256u) ||
                         !(((r.scan.flags & 0x8000u) && P / 32 == 1 && P % 32 >= 1 && P % 32 <= 24) || (P < 32 && !r.scan.hits) || attach_short(P, r.scan.flags, r.scan.hits != nullptr)))
                {
                    // rows of 33 .. 63 keys whose length is not a multiple of 16 bytes, single-table widths: through the wave-private
                    // aligned output image (dynamic LDS behind the tables; flags bit 14: never, bit 19: at every row length, A/B)
                    ScanArgs a1 = r.scan;
                    size_t dyn1 = dyn;
                    if (C <= 10 && (P & 15u) != 0 && !(r.scan.flags & 0x4000u) && ((P + 31) / 32 == 2 || (r.scan.flags & 0x80000u)) &&
                        dyn + (size_t)kWavesPerBlock * kLinearImageBytes <= (size_t)max_dyn) {
                        a1.flags |= 0x100000u;
                        dyn1 += (size_t)kWavesPerBlock * kLinearImageBytes;
                    }
                    hipLaunchKernelGGL((shared_linear_kernel<C, 2, 1>), lgrid, dim3(kBlockThreads), dyn1, r.stream, a1);
                }
                else {
                    ScanArgs a2 = r.scan;
                    if (attach_short(P, r.scan.flags, r.scan.hits != nullptr)) a2.flags |= 0x20000u;
                    allow_dynamic_lds<shared_linear2_kernel<C, 2>>(max_dyn, r.device);
                    hipLaunchKernelGGL((shared_linear2_kernel<C, 2>), lgrid, dim3(kBlockThreads), dyn, r.stream, a2);
                }
            } else if (!linear && !(r.scan.flags & 2u)) { // (flags bit 1: the per-group kernel, for A/B)
                // Hit counts in registers (flags bit 3: per-tile wave reductions / the histogram instead, for A/B): one 32-key
                // round in 32-bit registers (P <= 32: 2.5e8 x 9 bit, same box, P = 16 0.200 -> 0.158 ms, P = 32 0.298 -> 0.249),
                // two rounds in packed 16-bit halves (P <= 64, round 3: c = 9, P = 33 / 40 / 48 3.50 / 3.96 / 4.12 -> 4.44 / 4.81 /
                // 4.87 TB/s; not where the histogram counts -- c <= 12, P >= 64: 4.99 against 4.66).
                // Wider digits (BIG; flags bit 9: byte digits, for A/B) at the widths of three or four byte digits while two
                // blocks per CU still fit.
                constexpr bool kBigWidth = (C >= 17 && C <= 20) || (C >= 25 && C <= 30);
                const bool hist_counts = C <= 12 && P >= 64;
                const int rc = (r.scan.hits && !(r.scan.flags & 8u)) ? (P <= 32 ? 1 : ((P <= 64 && !hist_counts) ? 2 : 0)) : 0;
                bool big = false;
                if constexpr (kBigWidth)
                    big = !(r.scan.flags & 0x200u) &&
                          2 * ((size_t)((P + 31) / 32) * WideLutGeom<C, true>::TABLE_BYTES + lut_static_lds<C, VPL>()) <= 160 * 1024;
                // a 32-value word at a time (shared_wide3_kernel: half the registers, several waves per SIMD -- what the digit-table
                // widths need, and 0-20 % ahead at c <= 10 too: 2.5e8 x 9 bit, P = 9 / 24 / 63, TB/s with / without hit counts:
                // 3.63 / 4.41, 5.13 / 5.43, 5.21 / 5.56 against 3.44 / 3.58, 4.49 / 4.58, 4.39 / 4.49) for every scan it can count:
                // without hit counts, or up to 64 keys (flags bit 11: shared_wide2_kernel, A/B)
                {
                    if ((!r.scan.hits || rc != 0) && !(r.scan.flags & 0x800u)) {
                        auto go3 = [&](auto rc_c, auto big_c) {
                            constexpr int RC = decltype(rc_c)::value;
                            constexpr bool BIG = decltype(big_c)::value;
                            const size_t bdyn = (size_t)((P + 31) / 32) * WideLutGeom<C, BIG>::TABLE_BYTES;
                            allow_dynamic_lds<shared_wide3_kernel<C, 2, RC, BIG>>(max_dyn, r.device);
                            allow_dynamic_lds<shared_wide3_kernel<C, 18, RC, BIG>>(max_dyn, r.device);
                            const size_t per_block = bdyn + 4 * ScanGeom<C, 64>::LDS_BYTES + 256;
                            int fit = (int)((160 * 1024) / per_block);
                            constexpr int kWaves = (RC == 0 || C <= 10) ? 3 : 2; // the kernel's launch bound
                            fit = fit > kWaves ? kWaves : (fit < 1 ? 1 : fit);
                            const dim3 g3(grid_for(ntiles, r.max_blocks_per_cu > 0 && r.max_blocks_per_cu < fit ? r.max_blocks_per_cu : fit, r.num_cus));
                            if (nt_stores)
                                hipLaunchKernelGGL((shared_wide3_kernel<C, 18, RC, BIG>), g3, dim3(kBlockThreads), bdyn, r.stream, r.scan);
                            else
                                hipLaunchKernelGGL((shared_wide3_kernel<C, 2, RC, BIG>), g3, dim3(kBlockThreads), bdyn, r.stream, r.scan);
                        };
                        auto with_big3 = [&](auto rc_c) {
                            if constexpr (kBigWidth) {
                                if (big) {
                                    go3(rc_c, std::true_type{});
                                    return;
                                }
                            }
                            go3(rc_c, std::false_type{});
                        };
                        if (rc == 1)
                            with_big3(std::integral_constant<int, 1>{});
                        else if (rc == 2)
                            with_big3(std::integral_constant<int, 2>{});
                        else
                            with_big3(std::integral_constant<int, 0>{});
                        break;
                    }
                }
                auto go = [&](auto rc_c, auto big_c) {
                    constexpr int RC = decltype(rc_c)::value;
                    constexpr bool BIG = decltype(big_c)::value;
                    const size_t bdyn = (size_t)((P + 31) / 32) * WideLutGeom<C, BIG>::TABLE_BYTES;
                    allow_dynamic_lds<shared_wide2_kernel<C, 2, VPL, RC, BIG>>(max_dyn, r.device);
                    allow_dynamic_lds<shared_wide2_kernel<C, 18, VPL, RC, BIG>>(max_dyn, r.device);
                    if (nt_stores)
                        hipLaunchKernelGGL((shared_wide2_kernel<C, 18, VPL, RC, BIG>), grid, dim3(kBlockThreads), bdyn, r.stream, r.scan);
                    else
                        hipLaunchKernelGGL((shared_wide2_kernel<C, 2, VPL, RC, BIG>), grid, dim3(kBlockThreads), bdyn, r.stream, r.scan);
                };
                // (register counters in shared_wide2_kernel only at the single-table widths: the digit-table widths that come here
                // -- more than 64 keys with hit counts, or the A/B switch -- have no registers to spare for them)
                if (rc == 1 && C <= 10)
                    go(std::integral_constant<int, (C <= 10 ? 1 : 0)>{}, std::false_type{});
                else if (rc == 2 && C <= 10)
                    go(std::integral_constant<int, (C <= 10 ? 2 : 0)>{}, std::false_type{});
                else {
                    if constexpr (kBigWidth) {
                        if (big) {
                            go(std::integral_constant<int, 0>{}, std::true_type{});
                            break;
                        }
                    }
                    go(std::integral_constant<int, 0>{}, std::false_type{});
                }
            } else if (linear)
                hipLaunchKernelGGL((shared_wide_kernel<C, 2, VPL, 1>), grid, dim3(kBlockThreads), dyn, r.stream, r.scan);
            else if (nt_stores)
                hipLaunchKernelGGL((shared_wide_kernel<C, 18, VPL, 0>), grid, dim3(kBlockThreads), dyn, r.stream, r.scan);
            else
                hipLaunchKernelGGL((shared_wide_kernel<C, 2, VPL, 0>), grid, dim3(kBlockThreads), dyn, r.stream, r.scan);
        } else { // more keys than the tables hold: compare chain, ceil(P/8) passes over the registers
            static const int bpc = blocks_per_cu(shared_general_kernel<C, 2, VPL>);
            hipLaunchKernelGGL((shared_general_kernel<C, 2, VPL>), dim3(grid_for(ntiles, cap_bpc(bpc, r), r.num_cus)), dim3(kBlockThreads), 0,
                               r.stream, r.scan);
        }
        break;
    }
    case kOpScanIn: {
        constexpr int VPL = scan_vpl(C, kModeEq);
        using G = ScanGeom<C, VPL>;
        static const int bpc = blocks_per_cu(in_kernel<C, 2, VPL>);
        const uint64_t ntiles = (r.scan.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        // one LDS lookup per value: a second wave per SIMD hides the lookup latency (measured, 1e9 rows, P = 40:
        // c = 9 0.31 -> 0.23 ms, c = 12 0.32 -> 0.29, c = 16 0.51 -> 0.45 with two blocks per CU instead of one)
        int want = scan_bpc(bpc, G::TILE_BYTES, r);
        if (r.max_blocks_per_cu <= 0 && want < 2 && bpc >= 2) want = 2;
        const int ipol = r.scan_nt_stores < 0 ? (r.scan.n / 8 > (768ull << 20) ? 1 : 2) : r.scan_nt_stores; // as in launch_scan
        if (ipol == 2)
            hipLaunchKernelGGL((in_kernel<C, 34, VPL>), dim3(grid_for(ntiles, want, r.num_cus)), dim3(kBlockThreads), 0, r.stream, r.scan);
        else if (ipol == 1)
            hipLaunchKernelGGL((in_kernel<C, 18, VPL>), dim3(grid_for(ntiles, want, r.num_cus)), dim3(kBlockThreads), 0, r.stream, r.scan);
        else
            hipLaunchKernelGGL((in_kernel<C, 2, VPL>), dim3(grid_for(ntiles, want, r.num_cus)), dim3(kBlockThreads), 0, r.stream, r.scan);
        break;
    }
    case kOpSelect: {
        // predicate -> row ids: one block per CU (LDS: tiles + mask image + 16 KiB id stage per wave), every wave on a
        // chunk of select_tiles(C) tiles.  Chunks are claimed from a ticket counter, so the grid need not be resident
        // as a whole (another context's kernel may hold CUs): a look-back only ever waits for running waves
        constexpr int VPL = scan_vpl(C, kModeRange);
        using G = ScanGeom<C, VPL>;
        const uint64_t ntiles = (r.scan.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        const uint64_t nchunks = (ntiles + select_tiles(C) - 1) / select_tiles(C);
        // (flags bit 5 = option bit 13: round 2's single-role kernel, for A/B)
        if ((r.scan.flags & 32u) || r.select_single)
            hipLaunchKernelGGL((select_kernel<C, kModeRange, VPL>), dim3(grid_for(nchunks, 1, r.num_cus)), dim3(kBlockThreads), 0, r.stream, r.scan);
        else
            hipLaunchKernelGGL((select2_kernel<C, kModeRange, VPL>), dim3(grid_for(nchunks, 1, r.num_cus)), dim3(kSel2Waves * 64), 0, r.stream, r.scan);
        break;
    }
    case kOpScan2: {
        // two columns of this width in one launch: twice the DMA per tile, so half the blocks per CU of the plain scan
        constexpr int VPL = scan_vpl(C, kModeRange);
        using G = ScanGeom<C, VPL>;
        static const int bpc = blocks_per_cu(scan2_kernel<C, 34, VPL>);
        const uint64_t ntiles = (r.scan.n + G::TILE_VALUES - 1) / G::TILE_VALUES;
        const dim3 grid(grid_for(ntiles, scan_bpc(bpc, 2 * G::TILE_BYTES, r), r.num_cus));
        const int policy = r.scan_nt_stores < 0 ? (r.scan.n / 8 > (768ull << 20) ? 1 : 2) : r.scan_nt_stores;
        if (policy == 1)
            hipLaunchKernelGGL((scan2_kernel<C, 18, VPL>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
        else
            hipLaunchKernelGGL((scan2_kernel<C, 34, VPL>), grid, dim3(kBlockThreads), 0, r.stream, r.scan);
        break;
    }
    case kOpDecompress: {
        static const int bpc = blocks_per_cu(decompress_kernel<C, 18>);
        const uint64_t ntiles = (r.decomp.n + DecompGeom<C>::TILE_VALUES - 1) / DecompGeom<C>::TILE_VALUES;
        // Blocks per CU, launches back to back, 1e9 rows.  Round 1 (one box): four blocks 0.863-0.871 ms at c = 9 against
        // 0.903-0.907 with one.  Round 2, every width 1..32 on two boxes (profiles/r02_decompress_bpc_sweep.txt): the best
        // count differs between boxes and widths -- four blocks cost up to 16 % at c <= 8 on one box (c = 8: 0.909 / 0.948 /
        // 1.055 ms with 1 / 2 / 4), one block costs 6-12 % at c <= 5 on the other -- and TWO is within 2-4 % of the best
        // almost everywhere on both.
        const int want = bpc < 2 ? bpc : 2;
        const unsigned grid = grid_for(ntiles, r.max_blocks_per_cu > 0 ? cap_bpc(bpc, r) : want, r.num_cus);
        if (r.dma_aux == 0)
            hipLaunchKernelGGL((decompress_kernel<C, 0>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.decomp);
        else if (r.dma_aux == 2) // nt DMA loads only (tools/sweep.py --aux 2)
            hipLaunchKernelGGL((decompress_kernel<C, 2>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.decomp);
        else if (r.dma_aux == 34) // nt loads + write-through stores
            hipLaunchKernelGGL((decompress_kernel<C, 34>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.decomp);
        else // default (dma_aux 18): nt loads + nt stores -- the 4 B/value output is written once (+1-2 %)
            hipLaunchKernelGGL((decompress_kernel<C, 18>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.decomp);
        break;
    }
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int C> hipError_t launch_from(const LaunchReq &r)
{
    if (r.c == C) return launch_width<C>(r);
    if constexpr (C < MI355_WHI)
        return launch_from<C + 1>(r);
    else
        return hipErrorInvalidValue;
}

} // namespace

#define MI355_CAT2(a, b) a##b
#define MI355_CAT(a, b) MI355_CAT2(a, b)

hipError_t MI355_CAT(launch_group_, MI355_GROUP)(const LaunchReq &r) { return launch_from<MI355_WLO>(r); }

} // namespace mi355

// dispatch.hpp -- host-side launch request shared by capi.hip and the width-group translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mi355 {

enum Op { kOpScanEq = 0, kOpScanRange = 1, kOpSharedScan = 2, kOpDecompress = 3, kOpScanIn = 4, kOpSelect = 5, kOpScan2 = 6 };

struct LaunchReq {
    int op;
    unsigned c;
    hipStream_t stream;
    int device;            // the context's device (per-device kernel attributes)
    int num_cus;
    int max_blocks_per_cu; // 0 = what the occupancy query allows
    int dma_aux;           // cache policy of the HBM->LDS loads: 0 default, 2 nt
    int scan_nt_stores;    // bitmap stores of the eq / range scan: -1 by size, 0 plain, 1 non-temporal
    int scan_burst;        // eq / range scan: 0 = tiles per store burst chosen by width (burst_k), 1 = one tile per burst
    int select_single;     // kOpSelect: 1 = the older single-role kernel (option "select_kernel" = 1, A/B), 0 = decoder / expander roles
    int shared_vpl;        // shared scans of <= 8 keys: values per lane and tile, 0 = the engine's choice, 64, 128 (c <= 12)
    int *choice_out;       // kOpSharedScan: non-null = only report the kernel family that would run (0 one-pass LUT,
                           // 1 byte-entry multi-pass LUT, 2 dword-entry LUT, 3 compare chain), launch nothing
    ScanArgs scan;
    DecompArgs decomp;
};

// Persistent grid: at most (resident blocks per CU) x (CUs) blocks of 4 waves; each wave strides
// over the wave tiles.  Small inputs get one wave per tile.
inline unsigned grid_for(uint64_t ntiles, int blocks_per_cu, int num_cus)
{
    uint64_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    uint64_t cap = (uint64_t)blocks_per_cu * (uint64_t)num_cus;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

constexpr int kNumGroups = 8; // widths 1..32, 4 per group
hipError_t launch_group_0(const LaunchReq &);
hipError_t launch_group_1(const LaunchReq &);
hipError_t launch_group_2(const LaunchReq &);
hipError_t launch_group_3(const LaunchReq &);
hipError_t launch_group_4(const LaunchReq &);
hipError_t launch_group_5(const LaunchReq &);
hipError_t launch_group_6(const LaunchReq &);
hipError_t launch_group_7(const LaunchReq &);

} // namespace mi355

// width_group.hip -- instantiates the width-templated kernels for widths MI355_WLO..MI355_WHI and
// exports one launcher per group.  Compiled 8 times (4 widths each) so the build parallelises.
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

inline int cap_bpc(int bpc, const LaunchReq &r)
{
    return (r.max_blocks_per_cu > 0 && r.max_blocks_per_cu < bpc) ? r.max_blocks_per_cu : bpc;
}

template <int C> hipError_t launch_width(const LaunchReq &r)
{
    switch (r.op) {
    case kOpScanEq: {
        static const int bpc = blocks_per_cu(scan_kernel<C, kModeEq, 0>);
        const uint64_t ntiles = (r.scan.n + ScanGeom<C>::TILE_VALUES - 1) / ScanGeom<C>::TILE_VALUES;
        const unsigned grid = grid_for(ntiles, cap_bpc(bpc, r), r.num_cus);
        if (r.dma_aux == 2)
            hipLaunchKernelGGL((scan_kernel<C, kModeEq, 2>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.scan);
        else
            hipLaunchKernelGGL((scan_kernel<C, kModeEq, 0>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.scan);
        break;
    }
    case kOpScanRange: {
        static const int bpc = blocks_per_cu(scan_kernel<C, kModeRange, 0>);
        const uint64_t ntiles = (r.scan.n + ScanGeom<C>::TILE_VALUES - 1) / ScanGeom<C>::TILE_VALUES;
        const unsigned grid = grid_for(ntiles, cap_bpc(bpc, r), r.num_cus);
        hipLaunchKernelGGL((scan_kernel<C, kModeRange, 0>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.scan);
        break;
    }
    case kOpSharedScan: {
        static const int bpc = blocks_per_cu(scan_kernel<C, kModeShared, 0>);
        const uint64_t ntiles = (r.scan.n + ScanGeom<C>::TILE_VALUES - 1) / ScanGeom<C>::TILE_VALUES;
        const unsigned grid = grid_for(ntiles, cap_bpc(bpc, r), r.num_cus);
        hipLaunchKernelGGL((scan_kernel<C, kModeShared, 0>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.scan);
        break;
    }
    case kOpDecompress: {
        static const int bpc = blocks_per_cu(decompress_kernel<C, 0>);
        const uint64_t ntiles = (r.decomp.n + DecompGeom<C>::TILE_VALUES - 1) / DecompGeom<C>::TILE_VALUES;
        const unsigned grid = grid_for(ntiles, cap_bpc(bpc, r), r.num_cus);
        hipLaunchKernelGGL((decompress_kernel<C, 0>), dim3(grid), dim3(kBlockThreads), 0, r.stream, r.decomp);
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

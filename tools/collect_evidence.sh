#!/bin/bash
# after `gpurun -- bash tools/evidence.sh [tag]`: summarise the rocprofv3 passes and copy the sweeps into profiles/ (tag r03)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
for w in scan_eq scan_range shared_scan decompress; do python tools/summarize_profile.py gpurun_out/prof_${TAG}_$w $w $TAG 1000000000 9 > /dev/null; done
O=gpurun_out/${TAG}h
grep -v amdgpu.ids $O/width_sweep.txt > profiles/${TAG}_width_sweep_1e9.txt
grep -v amdgpu.ids $O/p_sweep.txt > profiles/${TAG}_shared_scan_P_sweep.txt
grep -v amdgpu.ids $O/p_all.txt > profiles/${TAG}_shared_scan_all_P.txt
grep -v amdgpu.ids $O/bench_next.txt > profiles/${TAG}_next_rows_1e9x9.txt
grep -v amdgpu.ids $O/pcie.txt > profiles/${TAG}_host_pointer_pcie.txt
grep -v amdgpu.ids $O/shard_sizes.txt > profiles/${TAG}_shard_sizes_1gpu.txt
grep -v amdgpu.ids $O/select_ab.txt > profiles/${TAG}_select_ab.txt
grep -v amdgpu.ids $O/select_widths.txt > profiles/${TAG}_select_widths.txt
cp $O/wide_widths.txt profiles/${TAG}_wide_widths_after.txt
cp $O/bench.json profiles/${TAG}_bench_line.json
cp $O/bench_random.json profiles/${TAG}_bench_line_random_column.json
for w in scan_eq scan_range shared_scan decompress; do grep -E "^\| \`|bench.py in the same|total =" profiles/${TAG}_${w}_1e09x9.md; done

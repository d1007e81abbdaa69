#!/bin/bash
# after `gpurun -- bash tools/evidence.sh`: summarise the rocprofv3 passes and copy the sweeps into profiles/ (tag r02)
set -e
cd "$(dirname "$0")/.."
for w in scan_eq scan_range shared_scan decompress; do python tools/summarize_profile.py gpurun_out/prof_r02_$w $w r02 1000000000 9 > /dev/null; done
O=gpurun_out/r2h
grep -v amdgpu.ids $O/width_sweep.txt > profiles/r02_width_sweep_1e9.txt
grep -v amdgpu.ids $O/p_sweep.txt > profiles/r02_shared_scan_P_sweep.txt
grep -v amdgpu.ids $O/p_all.txt > profiles/r02_shared_scan_all_P.txt
grep -v amdgpu.ids $O/bench_next.txt > profiles/r02_next_rows_1e9x9.txt
grep -v amdgpu.ids $O/pcie.txt > profiles/r02_host_pointer_pcie.txt
for w in scan_eq scan_range shared_scan decompress; do grep -E "^\| \`|bench.py in the same|total =" profiles/r02_${w}_1e09x9.md; done

#!/bin/bash
# round 2, GPU session e: blocks-per-CU x burst sweep of the equality scan, all widths
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; mkdir -p $O
for b in 1 2 3 4; do
python tools/ab_opts.py --workload scan_eq --bits 9 --column mod --rounds 3 --fixed max_blocks_per_cu=$b --opt scan_burst=0,1,4 2>&1 | grep -v amdgpu | sed "s/^/bpc=$b /" >> $O/bpc.txt
python tools/ab_opts.py --workload scan_eq --bits 9,12,16 --rounds 3 --fixed max_blocks_per_cu=$b --opt scan_burst=0,1,4 2>&1 | grep -v amdgpu | sed "s/^/bpc=$b /" >> $O/bpc.txt
done
python tools/ab_opts.py --workload scan_eq --bits 1,2,3,4,5,6,7,8,10,11,13,14,15 --rounds 3 --opt scan_burst=0,1,2,4 2>&1 | grep -v amdgpu > $O/widths.txt
cat $O/bpc.txt $O/widths.txt

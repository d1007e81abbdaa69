#!/bin/bash
# round 2, GPU session d: burst kernel (inline-asm mask loads) parity + A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O
for k in 2 4 1; do
  MI355_SCAN_BURST=$k python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_burst$k.log 2>&1; echo "pytest burst=$k rc=$?"; tail -n 2 $O/pytest_burst$k.log
done
python tools/ab_opts.py --workload scan_eq --bits 9 --column mod --opt scan_burst=0,1,2,4 > $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_eq --bits 9,5,7,12,16,17,21 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_range --bits 9,5,21 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_eq --bits 9 --fixed max_blocks_per_cu=2 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_and --bits 9,17 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload count --bits 9 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_eq --rows 1e8 --bits 9 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_eq --rows 1e7 --bits 9 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
grep -v amdgpu $O/ab_burst.txt

#!/bin/bash
# round 2, GPU session c: burst kernel parity (whole suite under MI355_SCAN_BURST=1,2,4), in-process A/B of the burst
# length, the mask prefetch, the counted vmcnt of the wide shared kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O
for k in 2 4 1; do
  MI355_SCAN_BURST=$k python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_burst$k.log 2>&1; echo "pytest burst=$k rc=$?"; tail -2 $O/pytest_burst$k.log
done
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python tools/ab_opts.py --workload scan_eq --bits 9 --column mod --opt scan_burst=0,1,2,4 > $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_eq --bits 9,5,12,17,21 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_range --bits 9,5,21 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_eq --bits 9 --fixed max_blocks_per_cu=2 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload scan_and --bits 9 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
python tools/ab_opts.py --workload count --bits 9 --opt scan_burst=0,1,2,4 >> $O/ab_burst.txt 2>&1
for P in 16 64 512; do for L in per_predicate linear; do
python tools/ab_opts.py --workload shared --rows 2.5e8 --P $P --layout $L --hits 0 --burst 10 --opt kernel_flags=1,0 >> $O/ab_wide.txt 2>&1
python tools/ab_opts.py --workload shared --rows 2.5e8 --P $P --layout $L --hits 1 --burst 10 --opt kernel_flags=1,0 >> $O/ab_wide.txt 2>&1
done; done
cat $O/ab_burst.txt $O/ab_wide.txt

#!/bin/bash
# round 2, GPU session b: regression, burst-length experiment, counted-vmcnt A/B, VPL-128 shared scans, count-only scan
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 tools/burst > $O/burst.txt 2>&1; echo "burst rc=$?"
timeout -k 10 300 python tools/count_only.py > $O/count_only.txt 2>&1; echo "count_only rc=$?"
timeout -k 10 600 python tools/sweep_p.py --rows 1000000000 --P 2,4,8 --vpl 64,128 --burst 20 --reps 5 > $O/vpl.txt 2>&1; echo "vpl rc=$?"
timeout -k 10 900 bash tools/ab_run.sh 'python tools/sweep_p.py --P 16,64,512 --burst 10 --reps 5' 2 > $O/ab_wide.txt 2>&1; echo "ab rc=$?"
cat $O/burst.txt $O/count_only.txt $O/vpl.txt

#!/bin/bash
# tools/profile.sh [workload] [tag] -- rocprofv3 passes for bench.py on the GPU box (run through gpurun).
# Pass 1: --kernel-trace --stats (timing).  Passes 2..: one --pmc pass per counter group (never combined with
# tracing domains other than the kernel trace), as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes.
set -o pipefail
W=${1:-scan_eq}
TAG=${2:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${W}
mkdir -p $OUT
ARGS="bench.py --workload $W --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS --steps 50 --warmup 10 > $OUT/stats.log 2>&1
echo "stats rc=$?" >> $OUT/stats.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ARGS --steps 6 --warmup 2 > $OUT/pmc_$name.log 2>&1
  echo "pmc $grp rc=$?" >> $OUT/stats.log
done
tail -3 $OUT/stats.log
find $OUT -name "*.csv" | head -40

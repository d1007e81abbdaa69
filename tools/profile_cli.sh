#!/bin/bash
# tools/profile_cli.sh <tag> <cli args...> -- PMC passes over the benchmark CLI (same rules as tools/profile.sh)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profcli_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- cli/shared_simd_scan_mi355 "$@" > $OUT/stats.log 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- cli/shared_simd_scan_mi355 "$@" > $OUT/pmc_$name.log 2>&1
done

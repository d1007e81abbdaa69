#!/bin/bash
# tools/ab_run.sh '<command>' [rounds=2] -- on the GPU box: run <command> with the old and the new library in turn
# (old, new, old, new ...), same process environment, same box.  See tools/ab_build.sh.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
CMD=${1:?usage: tools/ab_run.sh '<command>' [rounds]}
ROUNDS=${2:-2}
for r in $(seq 1 "$ROUNDS"); do
  for v in old new; do
    cp tools/_ab/libmi355scan_$v.so shared_simd_scan_amd/libmi355scan.so
    echo "== $v (round $r)"
    bash -c "$CMD"
  done
done
cp tools/_ab/libmi355scan_new.so shared_simd_scan_amd/libmi355scan.so

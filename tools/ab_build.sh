#!/bin/bash
# tools/ab_build.sh <git-ref> -- same-box A/B of two library builds (box-to-box variation on the pool is +-5 %, so
# before/after numbers from different gpurun calls cannot resolve a few per cent).  Run HERE (no GPU needed): builds
# libmi355scan.so of <git-ref> in a scratch worktree into tools/_ab/libmi355scan_old.so and copies the working tree's
# current build to tools/_ab/libmi355scan_new.so.  Then, through gpurun:  bash tools/ab_run.sh '<command>'
set -e
cd "$(dirname "$0")/.."
REF=${1:?usage: tools/ab_build.sh <git-ref>}
WT=$(mktemp -d /tmp/abwt.XXXXXX)
git worktree add -q --detach "$WT" "$REF"
trap 'git worktree remove --force "$WT"' EXIT
make -C "$WT/shared_simd_scan_amd/csrc" -j8 > /dev/null
make -C shared_simd_scan_amd/csrc -j8 > /dev/null
mkdir -p tools/_ab
cp "$WT/shared_simd_scan_amd/libmi355scan.so" tools/_ab/libmi355scan_old.so
cp shared_simd_scan_amd/libmi355scan.so tools/_ab/libmi355scan_new.so
ls -la tools/_ab/*.so

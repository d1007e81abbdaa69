#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "select or fused_mask or compare_chain or one_tile" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -n 15 $O/pytest_new.log

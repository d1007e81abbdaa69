#!/bin/bash
# run selected GPU tests: bash tools/run_one_test.sh "<-k expression>"
mkdir -p gpurun_out/one
python -m pytest tests/test_gpu_parity.py -q -x -k "$1" > gpurun_out/one/test.log 2>&1
rc=$?
tail -15 gpurun_out/one/test.log
exit $rc

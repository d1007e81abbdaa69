#!/bin/bash
set -e
out=gpurun_out/shits; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "shared or linear" > $out/test.log 2>&1
python tools/sweep_p.py --P 1,2,3,4,5,6,7,8 --layouts linear --burst 10 > $out/lin_small.log 2>&1
python tools/sweep_p.py --P 3,5,6,7 --bits 5 --layouts linear --burst 10 > $out/lin_small_c5.log 2>&1
python tools/sweep_p.py --P 3,5,6,7 --bits 17 --layouts linear --burst 10 > $out/lin_small_c17.log 2>&1

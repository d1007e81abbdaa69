#!/bin/bash
set -e
out=gpurun_out/shits; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "shared or linear" > $out/test.log 2>&1
python tools/sweep_p.py --P 9,11,13,15,17,23,27,31,32,33,40,47,63,64,65,95,127,128,257,511 --layouts linear --hits 1,0 --burst 4 --reps 3 > $out/lin_odd.log 2>&1

#!/bin/bash
set -e
out=gpurun_out/shits; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "shared or linear" > $out/test.log 2>&1
python tools/sweep_p.py --P 9,12,15,17,24,33,48,65,96,100,160,300,500 --layouts linear --burst 10 > $out/sweep_new.log 2>&1
MI355_KERNEL_FLAGS=2 python tools/sweep_p.py --P 9,12,15,17,24,33,48,65,96,100,160,300,500 --layouts linear --burst 10 > $out/sweep_old.log 2>&1
python tools/sweep_p.py --P 12,24,100 --bits 12 --layouts linear --burst 10 > $out/sweep_c12.log 2>&1
MI355_KERNEL_FLAGS=2 python tools/sweep_p.py --P 12,24,100 --bits 12 --layouts linear --burst 10 > $out/sweep_c12_old.log 2>&1

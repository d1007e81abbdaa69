#!/bin/bash
set -e
out=gpurun_out/shits; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "shared" > $out/test.log 2>&1
python tools/sweep_p.py --P 9,16,24,32,64 --burst 10 > $out/sweep_new.log 2>&1
MI355_KERNEL_FLAGS=8 python tools/sweep_p.py --P 16,32 --layouts per_predicate --burst 10 > $out/sweep_oldcounts.log 2>&1
python tools/sweep_p.py --P 16,32 --bits 12 --burst 10 > $out/sweep_c12.log 2>&1
python tools/sweep_p.py --P 16,32 --bits 16 --burst 10 > $out/sweep_c16.log 2>&1

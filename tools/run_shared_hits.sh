#!/bin/bash
set -e
out=gpurun_out/shits; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "shared or linear" > $out/test.log 2>&1
python tools/sweep_p.py --P 9,16,32,40,64,128 --layouts per_predicate --burst 10 > $out/pp_c9.log 2>&1
python tools/sweep_p.py --P 9,16,32,40,64,128 --bits 12 --layouts per_predicate --burst 10 > $out/pp_c12.log 2>&1
python tools/sweep_p.py --P 16,32,64 --bits 16 --layouts per_predicate --burst 10 > $out/pp_c16.log 2>&1
python tools/sweep_p.py --P 16,32,64 --bits 21 --layouts per_predicate --burst 10 > $out/pp_c21.log 2>&1

#!/bin/bash
set -e
out=gpurun_out/shits; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "shared or linear or golden or fuzz or cfg4 or 2_pow_32 or store_policy" > $out/test.log 2>&1
for c in 3 5 7 9 12 16 17 25; do
  python tools/sweep_p.py --bits $c --P 2 --burst 10 --reps 5 >> $out/pair.log 2>&1
  MI355_KERNEL_FLAGS=32 python tools/sweep_p.py --bits $c --P 2 --burst 10 --reps 5 >> $out/lut.log 2>&1
done
python tools/sweep_p.py --rows 1000000000 --bits 9 --P 2 --burst 10 --reps 5 >> $out/pair_1e9.log 2>&1
MI355_KERNEL_FLAGS=32 python tools/sweep_p.py --rows 1000000000 --bits 9 --P 2 --burst 10 --reps 5 >> $out/lut_1e9.log 2>&1

#!/bin/bash
out=gpurun_out/cliffs; mkdir -p $out; rm -f $out/*
for c in 9 13 17 21 25 32; do
  python tools/sweep_p.py --bits $c --P 9,16,24,32,64,128 --layouts per_predicate,linear --hits 1,0 --burst 4 --reps 3 > $out/lut_c$c.txt 2>&1
  MI355_KERNEL_FLAGS=64 python tools/sweep_p.py --bits $c --P 9,16,24,32,64,128 --layouts per_predicate,linear --hits 1,0 --burst 4 --reps 3 > $out/chain_c$c.txt 2>&1
done

#!/bin/bash
out=gpurun_out/cliffs; mkdir -p $out; rm -f $out/*
for c in 11 13 16 17 21 25; do
  MI355_KERNEL_FLAGS=128 python tools/sweep_p.py --bits $c --P 129,200,257,300,400,600,1000 --layouts linear --hits 1,0 --burst 4 --reps 3 > $out/new_c$c.txt 2>&1
  MI355_KERNEL_FLAGS=2 python tools/sweep_p.py --bits $c --P 129,200,257,300,400,600,1000 --layouts linear --hits 1,0 --burst 4 --reps 3 > $out/old_c$c.txt 2>&1
done

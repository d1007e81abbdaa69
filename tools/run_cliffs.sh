#!/bin/bash
out=gpurun_out/cliffs; mkdir -p $out; rm -f $out/*
for c in 11 13 17 25; do
  python tools/sweep_p.py --bits $c --P 129,150,257,300,350,400,600,1000 --layouts linear --hits 1,0 --burst 4 --reps 3 > $out/new_c$c.txt 2>&1
  MI355_KERNEL_FLAGS=2 python tools/sweep_p.py --bits $c --P 129,150,257,300,350,400,600,1000 --layouts linear --hits 1,0 --burst 4 --reps 3 > $out/old_c$c.txt 2>&1
done
for c in 3 5 6 7 8 9 10 12 16; do
  python tools/sweep_p.py --bits $c --P 16 --layouts linear --hits 1,0 --burst 8 --reps 3 >> $out/rp2.txt 2>&1
  MI355_KERNEL_FLAGS=16 python tools/sweep_p.py --bits $c --P 16 --layouts linear --hits 1,0 --burst 8 --reps 3 >> $out/rp1.txt 2>&1
done

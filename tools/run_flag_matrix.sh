#!/bin/bash
# the shared-scan, fuzz and selection parity tests under every A/B switch (the alternative kernels stay reachable: they
# must stay right; select_kernel's timing ablations live in other bits -- 512, 1024 -- and are not part of this)
out=gpurun_out/flags; mkdir -p $out; rm -f $out/*
rc=0
for f in 0 1 2 4 8 16 32 64 128 2+64; do
  v=$(( ${f/+/ + } ))
  MI355_KERNEL_FLAGS=$v python -m pytest tests/test_gpu_parity.py -q -x -k "shared or linear or cfg4 or golden_shared or fuzz or select or 2_pow_32" > $out/flags_$v.log 2>&1 || rc=1
  echo "flags=$v: $(tail -1 $out/flags_$v.log)"
done
exit $rc

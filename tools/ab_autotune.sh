#!/bin/bash
# same-box check of mi355_tune_dev: what it keeps per width, and bench.py with / without it
set -e
out=gpurun_out/autotune; mkdir -p $out; rm -f $out/*
python -m pytest tests/test_gpu_parity.py -q -x -k "load_time_tuning or hip_graph" > $out/test.log 2>&1
python - > $out/kept.log 2>&1 <<'PY'
import time, torch
from shared_simd_scan_amd import ScanEngine
eng = ScanEngine(0)
for c in (1, 2, 5, 9, 12, 16, 21, 32):
    col = eng.generate("splitmix", 10**9, c, 42)
    torch.cuda.synchronize(); t = time.perf_counter()
    kept = eng.tune(col)
    print(f"c={c:2d} tune {1e3*(time.perf_counter()-t):7.1f} ms  kept {kept}", flush=True)
    del col; torch.cuda.empty_cache()
PY
for i in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline >> $out/bench_notune.jsonl 2>>$out/err.log
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --tune >> $out/bench_tune.jsonl 2>>$out/err.log
done
python bench.py --workload decompress --steps 50 --warmup 5 --no-cpu-baseline >> $out/bench_dec.jsonl 2>>$out/err.log
python bench.py --workload decompress --steps 50 --warmup 5 --no-cpu-baseline --tune >> $out/bench_dec.jsonl 2>>$out/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune > $out/bench_driver_shape.json 2>>$out/err.log

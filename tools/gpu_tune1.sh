cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
for aux in 0 2; do for bpc in 0 1 2 3; do
  echo "== aux=$aux bpc=$bpc" >> gpurun_out/tune1.log
  MI355_DMA_AUX=$aux MI355_MAX_BLOCKS_PER_CU=$bpc timeout -k 10 120 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel_ms'], d['roofline']['achieved'], d['value'])" >> gpurun_out/tune1.log
done; done
echo "== random column" >> gpurun_out/tune1.log
timeout -k 10 120 $B --column random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel_ms'], d['roofline']['achieved'], d['value'], d['hits'])" >> gpurun_out/tune1.log
for w in scan_range shared_scan decompress; do
echo "== $w" >> gpurun_out/tune1.log
timeout -k 10 120 $B --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel_ms'], d['roofline']['achieved'], d['value'], d['hits'])" >> gpurun_out/tune1.log
done
cat gpurun_out/tune1.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_r01 -name "*stats*" | head

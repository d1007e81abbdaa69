#!/bin/bash
# tools/evidence.sh -- the evidence set of a round (run through gpurun): rocprofv3 stats + PMC per workload (tools/profile.sh),
# width sweep, P sweep, next-row bench, host-path rates, bench lines.  Outputs under gpurun_out/; summarise with
# tools/summarize_profile.py gpurun_out/prof_<tag>_<workload> <workload> <tag> 1000000000 9 and copy into profiles/.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}
O=gpurun_out/${TAG}h; mkdir -p $O
# a fresh box runs its first process a few per cent slower (clocks, page-in): one discarded run first
python bench.py --no-cpu-baseline --steps 400 > /dev/null 2>&1; python bench.py --workload decompress --no-cpu-baseline --steps 100 > /dev/null 2>&1
for w in scan_eq scan_range shared_scan decompress; do
  bash tools/profile.sh $w $TAG > $O/profile_$w.log 2>&1; echo "profile $w rc=$?"
done
python tools/sweep.py --bits 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32 --bpc 0 --reps 8 > $O/width_sweep.txt 2>&1; echo "width sweep rc=$?"
python tools/sweep_p.py --burst 5 --reps 5 > $O/p_sweep.txt 2>&1; echo "p sweep rc=$?"
python tools/sweep_p.py --P $(python -c "print(','.join(str(p) for p in list(range(1, 65)) + [65, 66, 80, 95, 96, 97, 111, 127, 128, 129, 150, 191, 192, 193, 200, 255, 256, 257, 300, 383, 384, 385, 400, 447, 448, 449, 500, 511, 512]))") --hits 1 --burst 4 --reps 3 > $O/p_all.txt 2>&1; echo "all-P sweep rc=$?"
python tools/bench_next.py > $O/bench_next.txt 2>&1; echo "bench_next rc=$?"
python tools/pcie_rate.py > $O/pcie.txt 2>&1; echo "pcie rc=$?"
python tools/shard_sizes.py > $O/shard_sizes.txt 2>&1; echo "shard sizes rc=$?"
python tools/select_ablate.py > $O/select_ab.txt 2>&1; echo "select ablate rc=$?"
python tools/select_widths.py > $O/select_widths.txt 2>&1; echo "select widths rc=$?"
python tools/profile_shared.py --bits 17,21,25 --P 16,64 --out $O/wide_widths.txt > /dev/null 2>&1; echo "wide widths profile rc=$?"
# the bench line's roofline.traffic comes from THIS run's PMC passes: summarise them into the box's profiles/ first
for w in scan_eq scan_range shared_scan decompress; do python tools/summarize_profile.py gpurun_out/prof_${TAG}_$w $w $TAG 1000000000 9 > /dev/null 2>&1; done
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --column random > $O/bench_random.json 2>> $O/bench.err; echo "bench random rc=$?"

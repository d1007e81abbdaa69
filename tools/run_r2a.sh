#!/bin/bash
# round 2, first GPU session: parity suite, bench (1 GPU, 2-rank gloo self-launch), host path, RCCL probe
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench1.json 2> gpurun_out/r2a/bench1.err; echo "bench1 rc=$?"
BENCH_BACKEND=gloo python bench.py --gpus 2 --rows 100000000 --steps 50 --pipelined-gather > gpurun_out/r2a/bench2.json 2> gpurun_out/r2a/bench2.err; echo "bench2 rc=$?"
python tools/pcie_rate.py > gpurun_out/r2a/pcie.txt 2>&1; echo "pcie rc=$?"
timeout -k 10 200 python tools/rccl_probe.py > gpurun_out/r2a/rccl_probe.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/r2a/bench1.json gpurun_out/r2a/bench2.json gpurun_out/r2a/pcie.txt gpurun_out/r2a/rccl_probe.txt | cut -c1-1500

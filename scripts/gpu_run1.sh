#!/bin/bash
# round-2 GPU call 1: regression of the GPU suite after the ops refactor + correlation tile-variant sweep + bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( nproc; lscpu | grep "Model name" | head -1 ) > gpurun_out/device.txt
timeout 300 python scripts/ubench/corr_bench.py --out gpurun_out/corr_variants.json > gpurun_out/corr_bench.log 2>&1; echo "corr_bench exit $?" >> gpurun_out/corr_bench.log
cat gpurun_out/corr_bench.log
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 240 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nocpu.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_nocpu.log
tail -3 gpurun_out/bench_nocpu.log

#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench; logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep "Model name" >> gpurun_out/device.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -8 gpurun_out/smoke.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log

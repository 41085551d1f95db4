#!/bin/bash
# round 3, call q: expansion phase of the fused Bottleneck tail: weight prefetch depth (2 / 4 quads) x residual prefetch before the K loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="timeout 300 python scripts/ubench/conv_bench.py --iters 10 --shapes tail64_240x320 tail64_120x160 tail128_60x80 --out gpurun_out/c3x.jsonl"
$B --tag base 2>&1 | grep -v "Warn\|amdgpu.ids"
for v in 40 21 41; do
  RFX_LIB=ransac-flow_amd/librfx_c3x$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bottleneck_tail" 2>&1 | tail -1
  RFX_LIB=ransac-flow_amd/librfx_c3x$v.so $B --tag x$v 2>&1 | grep -v "Warn\|amdgpu.ids"
done
$B --tag base 2>&1 | grep -v "Warn\|amdgpu.ids"

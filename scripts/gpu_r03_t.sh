#!/bin/bash
# round 3, call t: late residual prefetch in the fused tail's expansion K loop (behind the last weight load): bit-identity + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bottleneck_tail" 2>&1 | tail -1
B="timeout 300 python scripts/ubench/conv_bench.py --iters 10 --shapes tail64_240x320 tail64_120x160 tail128_60x80 tail128_50x66 --out gpurun_out/latepre.jsonl"
RFX_LIB=ransac-flow_amd/librfx_lp0.so $B --tag lp0 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag lp1 2>&1 | grep -v "Warn\|amdgpu.ids"
RFX_LIB=ransac-flow_amd/librfx_lp0.so $B --tag lp0 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag lp1 2>&1 | grep -v "Warn\|amdgpu.ids"

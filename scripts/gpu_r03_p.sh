#!/bin/bash
# round 3, call p: 256-pixel patches (TN = 4) for the 64-channel direct 3x3 / fused tail kernels: bit-identity, microbench A/B, end to end
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direct_3x3 or bottleneck_tail or conv2d_matches" 2>&1 | tail -2
B="timeout 300 python scripts/ubench/conv_bench.py --iters 10 --shapes tail64_240x320 tail64_120x160 fe64_240x320 --out gpurun_out/wide.jsonl"
RFX_C3_WIDE=0 $B --tag tn2 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag tn4 2>&1 | grep -v "Warn\|amdgpu.ids"
RFX_C3_WIDE=0 $B --tag tn2 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag tn4 2>&1 | grep -v "Warn\|amdgpu.ids"

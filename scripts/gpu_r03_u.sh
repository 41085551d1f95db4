#!/bin/bash
# round 3, call u: three workgroups per CU for the 64-channel fused tail (168 VGPRs): bit-identity + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
RFX_LIB=ransac-flow_amd/librfx_w3.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bottleneck_tail" 2>&1 | tail -1
B="timeout 300 python scripts/ubench/conv_bench.py --iters 10 --shapes tail64_240x320 tail64_120x160 tail64_100x132 --out gpurun_out/w3.jsonl"
$B --tag w2 2>&1 | grep -v "Warn\|amdgpu.ids"
RFX_LIB=ransac-flow_amd/librfx_w3.so $B --tag w3 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag w2 2>&1 | grep -v "Warn\|amdgpu.ids"
RFX_LIB=ransac-flow_amd/librfx_w3.so $B --tag w3 2>&1 | grep -v "Warn\|amdgpu.ids"

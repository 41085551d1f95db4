#!/bin/bash
# round 3, call m: ResNet stem occupancy variants (tile height x workgroups per CU): bit-identity + timing against the shipped kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="timeout 200 python scripts/ubench/conv_bench.py --iters 10 --shapes stem7_960x1280 stem7_480x640 --out gpurun_out/stem7_occ.jsonl"
$B --tag base 2>&1 | grep -v "Warn\|amdgpu.ids"
for v in th4w2 th3w3 th2w4 th2w3; do
  RFX_LIB=ransac-flow_amd/librfx_stem_$v.so timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "resnet_stem" 2>&1 | tail -1
  RFX_LIB=ransac-flow_amd/librfx_stem_$v.so $B --tag $v 2>&1 | grep -v "Warn\|amdgpu.ids"
done
$B --tag base 2>&1 | grep -v "Warn\|amdgpu.ids"

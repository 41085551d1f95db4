#!/bin/bash
# round 3, call x: persistent direct 3x3 kernel (tile loop, next tile's first K step in flight under the epilogue): bit-identity + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
B="timeout 300 python scripts/ubench/conv_bench.py --iters 10 --shapes tail64_240x320 tail128_60x80 fe64_240x320 l3_256_60x80 c128_120x160 head49_512_60x80 --out gpurun_out/persist.jsonl"
RFX_C3_PERSIST=0 $B --tag one_tile_per_wg 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag persistent 2>&1 | grep -v "Warn\|amdgpu.ids"
RFX_C3_PERSIST=0 $B --tag one_tile_per_wg 2>&1 | grep -v "Warn\|amdgpu.ids"
$B --tag persistent 2>&1 | grep -v "Warn\|amdgpu.ids"

#!/bin/bash
# round 3, call l: ResNet stem with both channel groups per workgroup (balanced sub-tile walk, LDS-resident weight block): bit-identity + timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stem" 2>&1 | tail -2
for r in 1 2; do
RFX_LIB=ransac-flow_amd/librfx_stemold.so timeout 200 python scripts/ubench/conv_bench.py --iters 10 --tag stem7_old --shapes stem7_960x1280 stem7_480x640 --out gpurun_out/stem7.jsonl 2>&1 | grep -v "Warn\|amdgpu.ids"
timeout 200 python scripts/ubench/conv_bench.py --iters 10 --tag stem7_v2 --shapes stem7_960x1280 stem7_480x640 --out gpurun_out/stem7.jsonl 2>&1 | grep -v "Warn\|amdgpu.ids"
done

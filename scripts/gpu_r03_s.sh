#!/bin/bash
# round 3, call s: per-workgroup phase stamps of the fused Bottleneck tail (make trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
RFX_LIB=ransac-flow_amd/librfx_trace.so timeout 200 python scripts/dbg/fused_trace.py 2>&1 | grep -v "Warn\|amdgpu.ids"
NORES=1 RFX_LIB=ransac-flow_amd/librfx_trace.so timeout 200 python scripts/dbg/fused_trace.py 2>&1 | grep -v "Warn\|amdgpu.ids"

#!/bin/bash
# round 3, call w: how much of the fused tail's 8.8 us prologue is index set-up, how much the first tile's memory round trip
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "stamp 1 after the first tile is in LDS (prologue = set-up + first tile):"
RFX_LIB=ransac-flow_amd/librfx_trace.so timeout 200 python scripts/dbg/fused_trace.py 2>&1 | grep -v "Warn\|amdgpu.ids"
echo "stamp 1 after the index set-up, before the first loads (prologue = set-up only; 'main' then includes the first tile):"
RFX_LIB=ransac-flow_amd/librfx_trace_setup.so timeout 200 python scripts/dbg/fused_trace.py 2>&1 | grep -v "Warn\|amdgpu.ids"

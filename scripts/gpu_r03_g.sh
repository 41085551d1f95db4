#!/bin/bash
# round-3: PMC passes (FETCH / WRITE / matrix-pipe busy per kernel) of the KITTI-shaped config 5 and of quick_start
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CONFIGS="" PMC=1 PMC_CONFIG=5 bash scripts/gpu_profile_r02.sh 2>&1 | tail -3
CONFIGS="" PMC=1 PMC_CONFIG=qs bash scripts/gpu_profile_r02.sh 2>&1 | tail -3
ls -la gpurun_out/pmc_summary*.json

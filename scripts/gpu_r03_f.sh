#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -4
timeout 900 python tests/run_parity_sweep.py ev 64 | tail -c 1500

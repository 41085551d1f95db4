#!/bin/bash
# experiments: one library, several environments (ENVS = ';'-separated assignments lists)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
IFS=';' read -ra E <<< "$ENVS"
for e in "${E[@]}"; do
  echo "== $e"
  env $e timeout 300 python scripts/ubench/conv_bench.py --shapes $SH --tag "$e" 2>&1 | grep -v "Warning\|amdgpu.ids"
done

#!/bin/bash
# experiments: price the pieces of a conv main loop (make c3dbgN / c1dbgN) on the bench's own shapes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/conv_dbg.jsonl
rm -f $OUT
for lib in $LIBS; do
  RFX_LIB=$PWD/ransac-flow_amd/$lib timeout 300 python scripts/ubench/conv_bench.py --shapes $SH --out $OUT 2>&1 | grep -v "Warning\|amdgpu.ids"
done

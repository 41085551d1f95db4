#!/bin/bash
# experiments: price the pieces of the direct 3x3 main loop (make c3dbgN) on the bench's own shapes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/conv_dbg.jsonl
rm -f $OUT
SH="fe64_240x320 c128_120x160 l3_256_60x80 l3_256_30x40"
for lib in librfx.so librfx_c3dbg1.so librfx_c3dbg2.so librfx_c3dbg3.so librfx_c3dbg4.so librfx_c3dbg6.so librfx.so; do
  RFX_LIB=$PWD/ransac-flow_amd/$lib timeout 300 python scripts/ubench/conv_bench.py --shapes $SH --out $OUT 2>&1 | grep -v Warning
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -5
OUT=gpurun_out/conv_new.jsonl
rm -f $OUT
timeout 600 python scripts/ubench/conv_bench.py --out $OUT $CONV_BENCH_ARGS 2>&1 | grep -v Warning
if [ -n "$CONV_BENCH_OLD" ]; then RFX_CONV_1X1=0 timeout 600 python scripts/ubench/conv_bench.py --out $OUT --tag old1x1 $CONV_BENCH_ARGS 2>&1 | grep -v Warning; fi

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for WS in 0 1; do
  RFX_CONV_WS=$WS RFX_BENCH_DUMP=gpurun_out/conv_shapes_ws$WS.csv timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 > gpurun_out/bench_ws$WS.log 2>&1
  tail -1 gpurun_out/bench_ws$WS.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WS=$WS', d['value'], 'pairs/s conv', d['roofline']['achieved'], d['roofline']['all_conv_tflops'])" || tail -3 gpurun_out/bench_ws$WS.log
done

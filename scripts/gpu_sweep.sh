#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for B in 16 32 64; do
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > gpurun_out/bench_b$B.log 2>&1
  tail -1 gpurun_out/bench_b$B.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', 'conv', d['roofline']['achieved'], d['roofline']['all_conv_tflops'], 'share', d['roofline']['conv_time_share'], 'corr GB/s', d['roofline_corr']['achieved'])" || tail -5 gpurun_out/bench_b$B.log
done

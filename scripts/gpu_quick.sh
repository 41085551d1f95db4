#!/bin/bash
# quick perf probe: conv parity test + bench at two batch sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "conv or networks or corr" 2>&1 | tail -3
for B in ${BATCHES:-8 64}; do
  RFX_BENCH_DUMP=gpurun_out/conv_shapes_b$B.csv timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > gpurun_out/bench_b$B.log 2>&1
  tail -1 gpurun_out/bench_b$B.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', 'conv', d['roofline']['achieved'], d['roofline']['all_conv_tflops'], 'share', d['roofline']['conv_time_share'], 'corr GB/s', d['roofline_corr']['achieved'])" || tail -5 gpurun_out/bench_b$B.log
done

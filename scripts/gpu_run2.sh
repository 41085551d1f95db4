#!/bin/bash
# round-2 GPU call: full GPU suite + smoke + default bench (CPU legs, parity sweep) + configs 2..5 + 2-rank rehearsal
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( nproc; lscpu | grep "Model name" | head -1 ) > gpurun_out/device.txt
if [ "$SKIP_TESTS" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
fi
if [ "$SKIP_BENCH" != "1" ]; then
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -c 6000 gpurun_out/bench.log; tail -8 gpurun_out/bench.err
fi
for c in ${CONFIGS:-2 3 4 5}; do
timeout 600 python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/bench_c$c.log 2> gpurun_out/bench_c$c.err; echo "bench c$c exit $?" >> gpurun_out/bench_c$c.err
tail -c 2500 gpurun_out/bench_c$c.log; tail -3 gpurun_out/bench_c$c.err
done
if [ "$REHEARSE" = "1" ]; then
RFX_BENCH_DEVICE=0 RFX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 16 > gpurun_out/bench_2ranks_1gpu.log 2> gpurun_out/bench_2ranks_1gpu.err; echo "rehearsal exit $?" >> gpurun_out/bench_2ranks_1gpu.err
tail -c 1500 gpurun_out/bench_2ranks_1gpu.log; tail -5 gpurun_out/bench_2ranks_1gpu.err
fi

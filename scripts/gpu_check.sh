#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench, rocprof; logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( nproc; lscpu | grep "Model name" | head -1 ) > gpurun_out/device.txt
if [ "$SKIP_TESTS" != "1" ]; then
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -6 gpurun_out/smoke.log
fi
timeout ${BENCH_TIMEOUT:-240} python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_nocpu.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_nocpu.log
tail -12 gpurun_out/bench_nocpu.log
if [ "$FULL_BENCH" = "1" ]; then
timeout 500 python bench.py $BENCH_ARGS > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -12 gpurun_out/bench.log
fi
if [ "$PROF" = "1" ]; then
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OLDPWD/gpurun_out/prof.log 2>&1
cd $OLDPWD
ls -R gpurun_out/prof | head -20
fi

#!/bin/bash
# round 3, call h: the records -> assembly integration test (SURVEY 8f1 -> 8f3) + the per-shape table of the conv launches of a config-3 step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_assemble.py -x -q -m gpu 2>&1 | tail -5
RFX_BENCH_DUMP=gpurun_out/conv_shapes_config3.csv timeout 600 python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_h.log 2> gpurun_out/bench_h.err
tail -2 gpurun_out/bench_h.err; head -30 gpurun_out/conv_shapes_config3.csv

#!/bin/bash
# round-3 first GPU run: full GPU suite with the new round kernels, default bench (config 3 headline + quick_start leg + CPU
# legs), loop dumps for the offline oracle sweeps, correlation microbench incl. the two-direction kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2> $O/bench_default.err; echo "bench exit $?"
tail -c 3000 $O/bench_default.log; tail -12 $O/bench_default.err
timeout 600 python tests/run_loop_dumps.py r03 > $O/loop_dumps.log 2>&1; tail -4 $O/loop_dumps.log
timeout 300 python scripts/ubench/corr_bench.py --n 64 128 --variants 3 5 --pairs --out $O/corr_variants.json 2>&1 | grep -v amdgpu.ids

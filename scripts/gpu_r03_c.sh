#!/bin/bash
# round-3 third GPU run: correlation variants (DMA issued by the light tap-group waves, ring of 5) on the bench shape, copy /
# ATen launch attribution of one config-3 step, the GPU tests touched since the last run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multih.py -m gpu -x -q > $O/pytest_c.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_c.log; grep "det gate" $O/pytest_c.log
timeout 300 python scripts/ubench/corr_bench.py --n 64 128 --variants 3 5 10 11 5 10 --iters 40 --out $O/corr_variants_ldw.json 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/dbg/copy_sites.py 16 > $O/copy_sites.txt 2>&1; tail -45 $O/copy_sites.txt

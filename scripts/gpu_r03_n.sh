#!/bin/bash
# round 3, call n: can RCCL run two ranks on ONE GPU here?  (rehearsal of the real "nccl" all_gather between two processes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
RFX_BENCH_DEVICE=0 NCCL_DEBUG=WARN timeout 240 python bench.py --config 3 --gpus 2 --steps 2 --warmup 1 --batch 16 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_c3_2ranks_1gpu_rccl.log 2> gpurun_out/bench_c3_2ranks_1gpu_rccl.err; echo "exit $?"
tail -c 1500 gpurun_out/bench_c3_2ranks_1gpu_rccl.err; grep -c "^{" gpurun_out/bench_c3_2ranks_1gpu_rccl.log

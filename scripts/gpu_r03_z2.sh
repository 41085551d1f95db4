#!/bin/bash
# round 3, call z2: grouped chains with per-chain side-stream pools (RFX_CHAIN_SIDE=1) against serial chains
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cs in 0 1; do for k in 2 3; do
RFX_CHAIN_SIDE=$cs RFX_GROUP_CHAINS=$k timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_z.log 2> gpurun_out/bench_z.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_z.log') if l.startswith('{')][0]); print('chain side streams $cs chains $k', j['ms_per_step'], j['config']['aligned_ok_last_step'])" 2>&1 | tail -1
done; done
RFX_GROUP_CHAINS=1 timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_z.log 2> gpurun_out/bench_z.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_z.log') if l.startswith('{')][0]); print('one chain (side streams on)', j['ms_per_step'])" 2>&1 | tail -1

#!/bin/bash
# round 3, call y: single-pair latency with the pair's 8 images split into k grouped chains on k streams (RFX_GROUP_CHAINS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for gs in 1 0; do for k in 1 2 3 4; do
RFX_GROUP_STREAMS=$gs RFX_GROUP_CHAINS=$k timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_y.log 2> gpurun_out/bench_y.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_y.log') if l.startswith('{')][0]); print('side streams $gs chains $k', j['ms_per_step'], j['config']['aligned_ok_last_step'])" 2>&1 | tail -1
done; done
RFX_GROUP_CHAINS=2 timeout 300 python -m pytest tests/test_gpu_multih.py -x -q -m gpu -k "graphed or grouped" 2>&1 | tail -1

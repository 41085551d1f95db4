#!/bin/bash
# round 3, call z3: how the pair's images are dealt to the two grouped chains
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do for sp in greedy bigsmall alternate; do
RFX_CHAIN_SPLIT=$sp timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_z.log 2> gpurun_out/bench_z.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_z.log') if l.startswith('{')][0]); print('split $sp', j['ms_per_step'], j['config']['aligned_ok_last_step'])" 2>&1 | tail -1
done; done

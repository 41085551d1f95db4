#!/bin/bash
# round 3, call v: single-pair latency (config 2) with the conv3x3/conv objects of commit e34911a against the final tree, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do
for l in librfx_old.so librfx.so; do
RFX_LIB=ransac-flow_amd/$l timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_v.log 2> gpurun_out/bench_v.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_v.log') if l.startswith('{')][0]); print('$l', j['ms_per_step'])"
done; done

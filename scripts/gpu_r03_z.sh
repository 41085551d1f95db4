#!/bin/bash
# round 3, call z: two grouped chains for a single pair as the default (side streams off inside chains): tests + latency, batch 1 and 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multih.py tests/test_gpu_pipeline.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -2
for k in 1 2 3; do
RFX_GROUP_CHAINS=$k timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_z.log 2> gpurun_out/bench_z.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_z.log') if l.startswith('{')][0]); print('batch 1 chains $k', j['ms_per_step'], j['config']['aligned_ok_last_step'])" 2>&1 | tail -1
done
for k in 1 2; do
RFX_GROUP_CHAINS=$k timeout 200 python bench.py --config 2 --batch 2 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_z2.log 2> gpurun_out/bench_z2.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_z2.log') if l.startswith('{')][0]); print('batch 2 chains $k', j['ms_per_step'], j['config']['aligned_ok_last_step'])" 2>&1 | tail -1
done
timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_c2.log 2> gpurun_out/bench_c2.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_c2.log') if l.startswith('{')][0]); print('default', j['ms_per_step'])"

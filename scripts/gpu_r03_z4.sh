#!/bin/bash
# round 3, call z4: one graph from the raw images (pyramid + trunk) for small batches: bit-identity + single-pair latency
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multih.py -x -q -m gpu -k "graph" 2>&1 | tail -2
for r in 1 2; do
timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_c2.log 2> gpurun_out/bench_c2.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_c2.log') if l.startswith('{')][0]); print('graph from raw', j['ms_per_step'], j['config']['aligned_ok_last_step'])" 2>&1 | tail -1
done
timeout 200 python bench.py --config 2 --batch 2 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_z2.log 2> gpurun_out/bench_z2.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/bench_z2.log') if l.startswith('{')][0]); print('batch 2', j['ms_per_step'])" 2>&1 | tail -1

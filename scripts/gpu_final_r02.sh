#!/bin/bash
# round-2 final evidence run: default bench (driver command), per-config benches, rocprofv3 stats per config, PMC passes,
# full 64-pair EV parity sweep, corr microbench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -c 1500 gpurun_out/bench.log; tail -4 gpurun_out/bench.err
for c in 2 3 4 5; do
timeout 600 python bench.py --config $c --steps 5 --warmup 2 > gpurun_out/bench_c$c.log 2> gpurun_out/bench_c$c.err; echo "bench c$c exit $?" >> gpurun_out/bench_c$c.err
python -c "import json,sys; j=json.loads([l for l in open('gpurun_out/bench_c$c.log') if l.startswith('{')][0]); print('config $c', j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], (j.get('roofline_corr') or {}).get('frac'))"
done
timeout 900 python tests/run_parity_sweep.py ev 64 | tail -c 1500
timeout 900 python tests/run_parity_sweep.py qs 64 | tail -c 1500
timeout 300 python scripts/ubench/corr_bench.py --n 64 128 --variants 3 4 6 5 21 22 --out gpurun_out/corr_variants.json 2>&1 | grep -v amdgpu.ids
rm -f gpurun_out/conv_bench_all.jsonl; timeout 300 python scripts/ubench/conv_bench.py --out gpurun_out/conv_bench_all.jsonl 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/conv_bench_all.txt
[ -x scripts/ubench/mfma_mix.bin ] && ./scripts/ubench/mfma_mix.bin | grep -v "^\[" | tail -19 > gpurun_out/mfma_mix.txt
PMC=1 bash scripts/gpu_profile_r02.sh 2>&1 | tail -12

#!/bin/bash
# round-3 final evidence run: the driver's bench command, the other bench configs, the 2-rank rehearsal of the config-3 step on
# one GPU, rocprofv3 kernel stats per config, PMC passes of the headline config, full 64-pair first-homography sweeps (qs, ev),
# correlation microbench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
RFX_PARITY_RECORDS=gpurun_out/bench_parity_records timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -c 600 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
for c in 2 3 4 5 qs; do
timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-parity > gpurun_out/bench_c$c.log 2> gpurun_out/bench_c$c.err; echo "bench c$c exit $?" >> gpurun_out/bench_c$c.err
python -c "import json,sys; j=json.loads([l for l in open('gpurun_out/bench_c$c.log') if l.startswith('{')][0]); r=j['roofline']; print('config $c', j['value'], j['ms_per_step'], r['kernel'], r['frac'], r['all_conv_tflops'], r['conv_time_share'], (j.get('roofline_corr') or {}).get('frac'))"
done
RFX_BENCH_BACKEND=gloo RFX_BENCH_DEVICE=0 timeout 600 python bench.py --config 3 --gpus 2 --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/bench_c3_2ranks_1gpu_gloo.log 2> gpurun_out/bench_c3_2ranks_1gpu_gloo.err; echo "2-rank exit $?"
timeout 900 python tests/run_parity_sweep.py qs 64 | tail -c 1200
timeout 900 python tests/run_parity_sweep.py ev 64 | tail -c 1200
timeout 300 python scripts/ubench/corr_bench.py --n 64 128 --variants 3 4 6 5 10 12 --pairs --out gpurun_out/corr_variants.json 2>&1 | grep -v amdgpu.ids | cut -c1-200
CONFIGS="3 qs 2 4 5" PMC=1 PMC_CONFIG=3 bash scripts/gpu_profile_r02.sh 2>&1 | tail -12
du -sh gpurun_out

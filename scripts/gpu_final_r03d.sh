#!/bin/bash
# round-3 final evidence refresh on the final tree: driver bench command, the other configs, 2-rank rehearsal, rocprofv3 kernel stats
# per config, PMC passes of the headline config
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
RFX_PARITY_RECORDS=gpurun_out/bench_parity_records timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -2 gpurun_out/bench.err
for c in 2 3 4 5 qs; do
timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_c$c.log 2> gpurun_out/bench_c$c.err; echo "bench c$c exit $?" >> gpurun_out/bench_c$c.err
python -c "import json,sys; j=json.loads([l for l in open('gpurun_out/bench_c$c.log') if l.startswith('{')][0]); r=j['roofline']; print('config $c', j['value'], j['ms_per_step'], r['kernel'], r['frac'], r['all_conv_tflops'], r['conv_time_share'], (j.get('roofline_corr') or {}).get('frac'))"
done
RFX_BENCH_BACKEND=gloo RFX_BENCH_DEVICE=0 timeout 600 python bench.py --config 3 --gpus 2 --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/bench_c3_2ranks_1gpu_gloo.log 2> gpurun_out/bench_c3_2ranks_1gpu_gloo.err; echo "2-rank exit $?"
CONFIGS="3 qs 2 4 5" PMC=1 PMC_CONFIG=3 bash scripts/gpu_profile_r02.sh 2>&1 | tail -8
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['all_conv_tflops'], j['extra']['quick_start']['value'], j['cpu_baseline']['value'])
p=j['parity']; print(p['pairs'], p['rounds'], p['rounds_exact_given_state'], p['max_H_delta'], p['oracle_wall_s'])
q=j['extra']['quick_start']['parity']; print(q['pairs'], q['identical_lists'], q['downstream_exact_given_matches'], q['max_abs_H_delta_identical'], q['oracle_wall_s'])
PY
du -sh gpurun_out

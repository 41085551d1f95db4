#!/bin/bash
# round-3 closing refresh after the latency work (graph from raw images, two grouped chains, l2norm): full GPU suite, smoke, driver bench
# command, config 2, rocprofv3 kernel stats of configs 3 and 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
RFX_PARITY_RECORDS=gpurun_out/bench_parity_records timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -1 gpurun_out/bench.err
timeout 200 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_c2.log 2> gpurun_out/bench_c2.err
CONFIGS="3 2" bash scripts/gpu_profile_r02.sh 2>&1 | tail -3 | cut -c1-200
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['all_conv_tflops'], j['extra']['quick_start']['value'], j['cpu_baseline']['value'])
p=j['parity']; print(p['pairs'], p['rounds'], p['rounds_exact_given_state'], p['max_H_delta'], p['oracle_wall_s'])
q=j['extra']['quick_start']['parity']; print(q['pairs'], q['identical_lists'], q['downstream_exact_given_matches'], q['max_abs_H_delta_identical'], q['oracle_wall_s'])
c=json.loads([l for l in open('gpurun_out/bench_c2.log') if l.startswith('{')][0]); print('config 2', c['ms_per_step'])
PY

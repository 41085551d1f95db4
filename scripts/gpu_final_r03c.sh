#!/bin/bash
# round-3 closing check of the final tree: the full GPU test suite, smoke(), and the driver's bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
RFX_PARITY_RECORDS=gpurun_out/bench_parity_records timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -2 gpurun_out/bench.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['all_conv_tflops'], j['extra']['quick_start']['value'], j['cpu_baseline']['value'])
p=j['parity']; print(p['pairs'], p['rounds'], p['rounds_exact_given_state'], p['max_H_delta'], p['oracle_wall_s'])
q=j['extra']['quick_start']['parity']; print(q['pairs'], q['identical_lists'], q['downstream_exact_given_matches'], q['max_abs_H_delta_identical'], q['oracle_wall_s'])
PY

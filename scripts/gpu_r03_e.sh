#!/bin/bash
# round-3: after the cell-coordinate fix -- full GPU suite, fresh loop dumps (all 64 config-3 pairs, configs 4 / 5), default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
rm -rf $O/dumps; timeout 600 python tests/run_loop_dumps.py r03 ev_loop:64 c4:6 c5:6 > $O/loop_dumps.log 2>&1; tail -4 $O/loop_dumps.log
RFX_PARITY_RECORDS=$O/bench_parity_records timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2> $O/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r03/bench_default.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['cpu_baseline']['value'])
print(json.dumps(j['parity'])[:1800])
print(json.dumps(j['extra']['quick_start']['parity'])[:1500])
PY

#!/bin/bash
# round 3, call r: kernel + pipeline + drop-in suites and a short config-3 bench with the 256-pixel-patch kernel selected
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for w in 0 1; do
RFX_C3_WIDE=$w timeout 300 python bench.py --config 3 --steps 8 --warmup 3 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_r$w.log 2> gpurun_out/bench_r$w.err
python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/bench_r$w.log") if l.startswith("{")][0])
print("wide=$w", j["value"], j["ms_per_step"], j["roofline"]["all_conv_tflops"])
PY
done

# bash scripts/stream_sweep.sh [VAR "values"]: BASELINE config 3 (exact mode, 8 timed steps) over one environment knob
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06
VAR=${1:-RFX_TRUNK_CHUNK}; VALS=${2:-"0 8 16 32"}
for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exact-leg --no-qs-leg 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$VAR=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done

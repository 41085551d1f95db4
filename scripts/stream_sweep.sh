cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06
for ts in 1 2 3 4 8; do for sp in 1 2 3; do
  RFX_TRUNK_STREAMS=$ts timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exact-leg --no-qs-leg --split $sp 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('trunk_streams=$ts split=$sp', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done; done

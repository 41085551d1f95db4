#!/bin/bash
# round 3, call k: PCIe-inclusive variants of the config-3 and quick_start steps (inputs from pinned host memory, records back to it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in 3 qs; do
  timeout 300 python bench.py --config $c --pcie --steps 8 --warmup 3 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_pcie_$c.log 2> gpurun_out/bench_pcie_$c.err
  timeout 300 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_hbm_$c.log 2> gpurun_out/bench_hbm_$c.err
  python - <<PY
import json
for k in ("pcie", "hbm"):
    j = json.loads([l for l in open("gpurun_out/bench_%s_$c.log" % k) if l.startswith("{")][0])
    print("$c", k, j["value"], j["ms_per_step"], j["config"].get("pcie_inclusive", ""))
PY
done

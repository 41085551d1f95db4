#!/bin/bash
# round 3, call j: device traces of the multi-homography drivers at the config-4 / config-5 sizes for 20 pairs each (the CPU oracle
# replays every round offline: oracle/parity_sweep.py --config c4|c5 --dump ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03b; export TMPDIR=/tmp
timeout 900 python tests/run_loop_dumps.py r03b c4:20 c5:20 2>&1 | grep -v Warn | tail -4

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/dbg/copy_sites.py 16 > $O/copy_sites.txt 2>&1; tail -60 $O/copy_sites.txt

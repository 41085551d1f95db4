#!/bin/bash
# round 3, call i: de-phasing experiment (common.h rfx_stagger) on the fused Bottleneck tail, the k-major 1x1 and the plain 3x3 kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_assemble.py -x -q -m gpu 2>&1 | tail -3
B="timeout 200 python scripts/ubench/conv_bench.py --iters 10 --out gpurun_out/stagger.jsonl"
F="tail64_240x320 tail64_120x160 tail128_60x80"
P="pw256_1024_60x80_res pw256_1024_30x40_res pw1024_256_30x40 pw512_128_60x80"
C="fe64_240x320 l3_256_60x80"
$B --tag base --shapes $F $P $C 2>&1 | grep -v Warn
for m in 0 3; do for t in 1500 3000 4500; do
  RFX_C3F_STAGGER=$t RFX_C3F_STAGGER_MODE=$m $B --tag c3f_m${m}_t$t --shapes $F 2>&1 | grep -v Warn
done; done
for m in 0 3; do for t in 500 1000 2000; do
  RFX_C1_STAGGER=$t RFX_C1_STAGGER_MODE=$m $B --tag c1_m${m}_t$t --shapes $P 2>&1 | grep -v Warn
done; done
for m in 0 3; do
  RFX_C3_STAGGER=2500 RFX_C3_STAGGER_MODE=$m $B --tag c3_m${m}_t2500 --shapes $C 2>&1 | grep -v Warn
done

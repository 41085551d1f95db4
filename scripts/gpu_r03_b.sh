#!/bin/bash
# round-3 second GPU run: full GPU suite (det-gate fuzz, round kernels), loop dumps for the offline oracle sweeps (all 64
# bench pairs of config 3; configs 4 / 5), 2-rank rehearsal of the config-3 step on one GPU (gloo), the other bench configs,
# correlation microbench on the config-4 / 5 shapes, rocprofv3 kernel stats of the config-3 step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $O/lscpu.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
timeout 600 python tests/run_loop_dumps.py r03 ev_loop:64 c4:6 c5:6 > $O/loop_dumps.log 2>&1; tail -4 $O/loop_dumps.log
RFX_BENCH_BACKEND=gloo RFX_BENCH_DEVICE=0 timeout 600 python bench.py --config 3 --gpus 2 --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > $O/bench_c3_2ranks_1gpu_gloo.log 2> $O/bench_c3_2ranks_1gpu_gloo.err; echo "2-rank exit $?"; tail -c 600 $O/bench_c3_2ranks_1gpu_gloo.log
for c in 2 4 5; do
timeout 600 python bench.py --config $c --steps 5 --warmup 2 > $O/bench_c$c.log 2> $O/bench_c$c.err; echo "bench c$c exit $?"
python -c "import json,sys; j=json.loads([l for l in open('$O/bench_c$c.log') if l.startswith('{')][0]); print('config $c', j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['conv_time_share'], (j.get('roofline_corr') or {}).get('frac'), (j.get('roofline_corr') or {}).get('per_direction_accounting',{}).get('frac'))"
done
for w in 0 256 512; do
RFX_CORR_MIN_WGS=$w timeout 200 python scripts/ubench/corr_bench.py --n 16 --shape 256 90 120 --variants 3 --pairs --iters 20 2>&1 | grep pairs | sed "s/^/minwgs $w c4 /"
RFX_CORR_MIN_WGS=$w timeout 200 python scripts/ubench/corr_bench.py --n 8 --shape 256 81 268 --variants 3 --pairs --iters 20 2>&1 | grep pairs | sed "s/^/minwgs $w c5 /"
RFX_CORR_MIN_WGS=$w timeout 200 python scripts/ubench/corr_bench.py --n 8 --shape 256 41 136 --variants 3 --pairs --iters 20 2>&1 | grep pairs | sed "s/^/minwgs $w c5d2 /"
RFX_CORR_MIN_WGS=$w timeout 200 python scripts/ubench/corr_bench.py --n 24 --shape 256 60 80 --variants 3 --pairs --iters 20 2>&1 | grep pairs | sed "s/^/minwgs $w ev24 /"
done 2>&1 | tee $O/corr_small_launches.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-qs-leg > $GRAFT_REPO_ROOT/$O/prof_c3.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; find $O/prof_c3 -name "*kernel_stats.csv" | head -2; find $O/prof_c3 -name "*.db" -delete; find $O/prof_c3 -name "*kernel_trace.csv" -delete; du -sh $O

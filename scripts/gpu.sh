#!/bin/bash
# ONE parametrised GPU-box script (replaces the per-experiment gpu_r0N_*.sh wrappers of earlier rounds).
#   gpurun --timeout 1800 -- 'bash scripts/gpu.sh <step> [<step> ...]'     outputs under gpurun_out/$TAG (default r05)
# steps: exact_probe hosttest ref tests smoke bench sweeps_full kernels_ab newtests bench_nocpu feature_error[_qs|_ev] profile pmc c2 c4 c5 sweeps mmprobe ubench_corr ubench_conv ubench_mnn
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${TAG:-r06}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case $step in
    ref)        ls oracle/_ref oracle/_ref/utils oracle/_ref/_extract 2>&1 | head -20; python -c "import sys; sys.path.insert(0,'oracle'); import ref_loader; print(ref_loader.REF_ROOT, ref_loader.kind() if ref_loader.available() else 'ABSENT')" ;;
    tests)      timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -${TAIL:-25} ;;
    newtests)   timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multih.py "tests/test_gpu_dropin.py" -x -q -m gpu --durations=5 2>&1 | tail -${TAIL:-25} ;;
    kittitests) timeout 900 python -m pytest tests -x -q -m gpu -k "kitti or lock_step" --durations=5 2>&1 | tail -${TAIL:-12} ;;
    smoke)      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 ;;
    bench)      RFX_PARITY_RECORDS=$OUT/bench_parity_records timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err | cut -c1-300
                python scripts/bench_digest.py $OUT/bench.log ;;
    bench_nocpu) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg > $OUT/bench_nocpu.log 2> $OUT/bench_nocpu.err; python scripts/bench_digest.py $OUT/bench_nocpu.log ;;
    feature_error_qs) timeout 1500 python scripts/feature_error.py --config qs --pairs ${FE_PAIRS:-12} --threads 16 --out $OUT/feature_error_qs.json > $OUT/feature_error_qs.log 2>&1; tail -40 $OUT/feature_error_qs.log | grep -A3 "differing_lists\|pairs\"" | head -60 ;;
    feature_error_ev) timeout 1500 python scripts/feature_error.py --config ev --pairs ${FE_PAIRS:-6} --threads 16 --out $OUT/feature_error_ev.json > $OUT/feature_error_ev.log 2>&1; tail -5 $OUT/feature_error_ev.log ;;
    c2)         timeout 300 python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/bench_c2.log 2> $OUT/bench_c2.err; python scripts/bench_digest.py $OUT/bench_c2.log ;;
    c4)         timeout 600 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.log 2> $OUT/bench_c4.err; python scripts/bench_digest.py $OUT/bench_c4.log ;;
    c5)         timeout 600 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.log 2> $OUT/bench_c5.err; python scripts/bench_digest.py $OUT/bench_c5.log ;;
    profile)    for c in ${CONFIGS:-3}; do rm -rf $OUT/prof_$c; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o k -- python bench.py --config $c --steps 5 --warmup 2 --single-stream --no-cpu-baseline --no-exact-leg --no-qs-leg > $OUT/prof_$c.log 2>&1; f=$(find $OUT/prof_$c -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_config$c.csv; head -12 "$f" | cut -c1-150; find $OUT/prof_$c -name "*.csv" ! -name "*stats*" -delete; done ;;
    pmc)        for c in ${CONFIGS:-3}; do for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do n=$(echo $grp | tr ' ' '_' | cut -c1-24); rm -rf $OUT/pmc_${c}_$n; timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_${c}_$n -o p -- python bench.py --config $c --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-qs-leg --no-exact-leg > $OUT/pmc_${c}_$n.log 2>&1; tail -1 $OUT/pmc_${c}_$n.log | cut -c1-160; done; python scripts/pmc_summary.py --dirs $OUT/pmc_${c}_* --cmd "rocprofv3 --pmc <group> --kernel-trace -- python bench.py --config $c --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-qs-leg --no-exact-leg" --out $OUT/pmc_summary_config$c.json 2>&1 | tail -12; find $OUT -path "*pmc_${c}_*" -name "*.csv" -delete; done ;;
    ubench_mnn) timeout 600 python scripts/ubench/mnn_bench.py --out $OUT/mnn_forms.json 2>&1 | tail -5 ;;
    sweeps)     # full 64-pair first-homography sweeps (device vs reference) + the reference against itself on the same box
                for c in ${SWEEP_CFGS:-qs ev}; do timeout 1200 python tests/run_parity_sweep.py $c 64 2>&1 | tail -1 | cut -c1-900; cp gpurun_out/parity_sweep_${c}_64.json $OUT/parity_sweep_${c}_64pairs.json
                  timeout 900 python oracle/parity_sweep.py --config $c --stability --threads 8 --budget 300 --records $OUT/oracle_vs_oracle_${c}_64pairs.json 2>&1 | tail -1 | cut -c1-900; done ;;
    sweeps_full) # VERDICT r4 #1a: first-homography sweeps over ALL seeds (qs 0..127, ev 0..159) with the host-resolved score chunk, then the
                # reference against itself on the same seeds (last: cut first if the budget runs out)
                timeout 900 python tests/run_parity_sweep.py qs ${QS_N:-128} 2>&1 | tail -1 | cut -c1-1200; cp gpurun_out/parity_sweep_qs_${QS_N:-128}.json $OUT/parity_sweep_qs_${QS_N:-128}pairs.json
                timeout 1200 python tests/run_parity_sweep.py ev ${EV_N:-160} 2>&1 | tail -1 | cut -c1-1200; cp gpurun_out/parity_sweep_ev_${EV_N:-160}.json $OUT/parity_sweep_ev_${EV_N:-160}pairs.json
                timeout 900 python oracle/parity_sweep.py --config qs --stability --threads 8 --budget 500 --seeds $(seq 0 $((${QS_N:-128}-1))) --records $OUT/oracle_vs_oracle_qs_${QS_N:-128}pairs.json 2>&1 | tail -1 | cut -c1-900
                timeout 1200 python oracle/parity_sweep.py --config ev --stability --threads 8 --budget 800 --seeds $(seq 0 $((${EV_N:-160}-1))) --records $OUT/oracle_vs_oracle_ev_${EV_N:-160}pairs.json 2>&1 | tail -1 | cut -c1-900 ;;
    kernels_ab) # round 5 kernel A/B on ONE box: fused tails (chunked + interleaved epilogue vs the burst form vs round 4's chains), the
                # implicit-GEMM kernel with the k-major A image vs the transposed one
                T="tail64_120x160 tail128_60x80 tail64_240x320 tail64_100x132 tail128_50x66"
                G="ds_256_512_s2_120x160 s2_128_128_120x160 head49_512_60x80"
                timeout 300 python scripts/ubench/conv_bench.py --shapes $T $G pw256_1024_30x40_res fe64_240x320 --out $OUT/conv_ab_r5.json 2>&1 | tail -14
                RFX_LIB=$PWD/ransac-flow_amd/librfx_noint.so timeout 300 python scripts/ubench/conv_bench.py --shapes $T --out $OUT/conv_ab_burst_epilogue.json 2>&1 | tail -6
                RFX_C3_TAIL_CHUNK=0 RFX_LIB=$PWD/ransac-flow_amd/librfx_noint.so timeout 300 python scripts/ubench/conv_bench.py --shapes $T --out $OUT/conv_ab_r4_tails.json 2>&1 | tail -6
                RFX_LIB=$PWD/ransac-flow_amd/librfx_oldconv.so timeout 300 python scripts/ubench/conv_bench.py --shapes $G --out $OUT/conv_ab_transposed_A.json 2>&1 | tail -4 ;;
    sweeps_more) # further seeds for the flip statistics (qs 128..383, ev 160..319): device vs reference, then reference vs reference
                timeout 1200 python tests/run_parity_sweep.py qs 256 128 2>&1 | tail -1 | cut -c1-1200; cp gpurun_out/parity_sweep_qs_256_from128.json $OUT/parity_sweep_qs_seeds128_383.json
                timeout 1200 python tests/run_parity_sweep.py ev 160 160 2>&1 | tail -1 | cut -c1-1200; cp gpurun_out/parity_sweep_ev_160_from160.json $OUT/parity_sweep_ev_seeds160_319.json
                timeout 1200 python oracle/parity_sweep.py --config qs --stability --threads 8 --budget 900 --seeds $(seq 128 383) --records $OUT/oracle_vs_oracle_qs_seeds128_383.json 2>&1 | tail -1 | cut -c1-900
                timeout 1200 python oracle/parity_sweep.py --config ev --stability --threads 8 --budget 900 --seeds $(seq 160 319) --records $OUT/oracle_vs_oracle_ev_seeds160_319.json 2>&1 | tail -1 | cut -c1-900 ;;
    sweeps_dev) # device-vs-reference first-homography sweeps only (the reference-vs-reference figures of the same seeds are on file)
                timeout 900 python tests/run_parity_sweep.py qs ${QS_N:-128} 2>&1 | tail -1 | cut -c1-1200; cp gpurun_out/parity_sweep_qs_${QS_N:-128}.json $OUT/parity_sweep_qs_${QS_N:-128}pairs${SUFFIX:-}.json
                timeout 1200 python tests/run_parity_sweep.py ev ${EV_N:-160} 2>&1 | tail -1 | cut -c1-1200; cp gpurun_out/parity_sweep_ev_${EV_N:-160}.json $OUT/parity_sweep_ev_${EV_N:-160}pairs${SUFFIX:-}.json ;;
    exact_probe) timeout 900 python scripts/exact_mode_probe.py --steps ${PROBE_STEPS:-5} --split ${PROBE_SPLIT:-2} --out $OUT/exact_mode_probe${SUFFIX:-}.json 2>&1 | tail -12 | cut -c1-1500 ;;
    hosttest)   timeout 300 python -m pytest tests/test_oracle.py -x -q -k "native_lapack or exports" 2>&1 | tail -3 ;;
    mmprobe)    timeout 120 python scripts/mm_blocking_probe.py --out $OUT/mm_blocking_probe.json 2>&1 | tail -8 ;;
    ubench_corr) timeout 600 python scripts/ubench/corr_bench.py ${CORR_ARGS:-} 2>&1 | tail -30 ;;
    ubench_conv) timeout 600 python scripts/ubench/conv_bench.py ${CONV_ARGS:-} 2>&1 | tail -40 ;;
    *)          echo "unknown step $step" ;;
  esac
done
echo "=== done ($(date +%T))"

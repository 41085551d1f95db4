#!/bin/bash
# round 2: rocprofv3 kernel-trace stats per bench config + PMC passes (own runs, kernel-trace only) of the default config
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in ${CONFIGS:-qs 2 3 4 5}; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-config3-leg > $R/gpurun_out/prof_$c.log 2>&1
  f=$(find $R/gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/kernel_stats_$c.csv
  find $R/gpurun_out/prof_$c -name "*kernel_trace.csv" -delete
  grep "^{" $R/gpurun_out/prof_$c.log | head -c 600; echo
done
if [ "$PMC" = "1" ]; then
ARGS="--config ${PMC_CONFIG:-qs} --steps 1 --warmup 1 --no-cpu-baseline --no-config3-leg"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            for c, val in v.items():
                out[k][c] = val; out[k]["launches_" + c] = cnt[(k, c)]
        os.remove(f)
    for f in glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % d, recursive=True):
        if d == "pmc_sq":
            tot = collections.defaultdict(float); n = collections.Counter()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]; tot[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); n[k] += 1
            for k in tot:
                out[k]["pmc_run_total_ns"] = tot[k]; out[k]["pmc_run_launches"] = n[k]
        os.remove(f)
json.dump(out, open("gpurun_out/pmc_summary%s.json" % ("" if os.environ.get("PMC_CONFIG", "qs") == "qs" else "_config" + os.environ["PMC_CONFIG"]), "w"), indent=1)
print("pmc kernels:", len(out))
PY
fi
ls gpurun_out/kernel_stats_*.csv

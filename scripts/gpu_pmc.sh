#!/bin/bash
# PMC passes (own runs, kernel-trace only) over a short bench.  Output: gpurun_out/pmc_*/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --batch ${B:-8}"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
cd $R
ls -la gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write
tail -3 gpurun_out/pmc_sq.log
# shrink: aggregate counter csv per kernel name
python - <<'PY'
import csv, glob, collections, os
for d in ("pmc_sq","pmc_fetch","pmc_write"):
    for f in glob.glob("gpurun_out/%s/*counter_collection.csv"%d):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        with open("gpurun_out/%s_summary.csv"%d,"w") as o:
            for k,v in agg.items():
                for c,val in v.items(): o.write("%s,%s,%d,%.6g\n"%(k.replace(","," "),c,cnt[(k,c)],val))
        os.remove(f)
PY
head -50 gpurun_out/pmc_sq_summary.csv

#!/bin/bash
# correlation-kernel experiments: variant timing + PMC passes (own runs, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
V="${VARIANTS:-3 5 21 22 23}"
timeout 300 python scripts/ubench/corr_bench.py --n 64 --variants $V --out gpurun_out/corr_variants2.json > gpurun_out/corr_bench2.log 2>&1
cat gpurun_out/corr_bench2.log
cd /tmp
PV="${PMC_VARIANTS:-3 5}"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/cpmc_sq -o p -- python $R/scripts/ubench/corr_bench.py --n 64 --variants $PV --iters 6 > $R/gpurun_out/cpmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/cpmc_fetch -o p -- python $R/scripts/ubench/corr_bench.py --n 64 --variants $PV --iters 6 > $R/gpurun_out/cpmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/cpmc_write -o p -- python $R/scripts/ubench/corr_bench.py --n 64 --variants $PV --iters 6 > $R/gpurun_out/cpmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/cpmc_tcc -o p -- python $R/scripts/ubench/corr_bench.py --n 64 --variants $PV --iters 6 > $R/gpurun_out/cpmc_tcc.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in ("cpmc_sq","cpmc_fetch","cpmc_write","cpmc_tcc"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "corr7" not in r["Kernel_Name"]: continue
            k=r["Kernel_Name"][:80]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        with open("gpurun_out/%s_summary.csv"%d,"w") as o:
            for k,v in agg.items():
                for c,val in v.items(): o.write("%s|%s|%d|%.6g|%.6g\n"%(k.replace(","," "),c,cnt[(k,c)],val,val/cnt[(k,c)]))
        os.remove(f)
    print(open("gpurun_out/%s_summary.csv"%d).read() if os.path.exists("gpurun_out/%s_summary.csv"%d) else d+": no csv")
PY
tail -3 gpurun_out/cpmc_sq.log

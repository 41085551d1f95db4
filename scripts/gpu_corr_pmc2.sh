#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
PV="${PMC_VARIANTS:-46 48}"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/cpmc_sq -o p -- python $R/scripts/ubench/corr_bench.py --n 64 --variants $PV --iters 6 > $R/gpurun_out/cpmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/cpmc_sq2 -o p -- python $R/scripts/ubench/corr_bench.py --n 64 --variants $PV --iters 6 > $R/gpurun_out/cpmc_sq2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in ("cpmc_sq","cpmc_sq2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "corr7" not in r["Kernel_Name"]: continue
            k=r["Kernel_Name"][:80]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            for c,val in v.items(): print("%s|%s|%d|%.6g"%(k[40:75].replace(","," "),c,cnt[(k,c)],val/cnt[(k,c)]))
        os.remove(f)
PY
tail -2 gpurun_out/cpmc_sq2.log; head -c 3000 gpurun_out/sq_counters.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc2 -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch ${B:-64} > $R/gpurun_out/pmc2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
for f in glob.glob("gpurun_out/pmc2/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    with open("gpurun_out/pmc2_summary.csv","w") as o:
        for k,v in agg.items():
            for c,val in v.items(): o.write("%s,%s,%d,%.6g\n"%(k.replace(","," "),c,cnt[(k,c)],val))
    os.remove(f)
PY
grep -E "corr7|conv2d_mfma_kernel<2|mnn_tile" gpurun_out/pmc2_summary.csv

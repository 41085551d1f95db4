#!/bin/bash
# Final profile of the round: rocprofv3 kernel stats of the bench command + PMC passes (own runs, kernel-trace only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
B=${B:-64}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > $R/gpurun_out/prof.log 2>&1
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --batch $B"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os, json
out = {}
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_sq/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d = dur[r["Kernel_Name"]]; d[0] += 1; d[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob("gpurun_out/%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            e = out.setdefault(k, {})
            for c, val in v.items():
                e[c] = val; e["launches_" + c] = cnt[(k, c)]
        os.remove(f)
for k, e in out.items():
    if k in dur:
        e["pmc_run_launches"] = dur[k][0]; e["pmc_run_total_ns"] = dur[k][1]
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
keep = {k: v for k, v in out.items() if any(s in k for s in ("conv2d", "conv3x3", "stem", "corr7", "mnn_tile", "l2norm", "maxpool", "maxblur", "lanczos"))}
for k, v in keep.items():
    print(k[:80], {c: v[c] for c in v if not c.startswith("launches") and not c.startswith("pmc_run")})
PY
head -12 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-160
tail -1 gpurun_out/prof.log | cut -c1-400

#!/usr/bin/env python
"""Per-kernel, per-launch summary of rocprofv3 --pmc passes (evidence helper; scripts/gpu.sh step `pmc`).

    python scripts/pmc_summary.py --dirs gpurun_out/rNN/pmc_3_* --cmd "python bench.py --config 3 ..." --out profiles/rNN_pmc_summary_config3.json

Every directory holds ONE pass (--pmc <group> --kernel-trace: counters are collected in their own runs, as gpurun requires).
FETCH_SIZE / WRITE_SIZE arrive in KB; bytes = x 1024.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports
exactly half of the bytes of wide coalesced streaming reads (16 B / lane, global_load and buffer_load..lds alike) -> the
`fetch_bytes_corrected` column doubles it for the kernels whose loads are 16-byte (listed in WIDE); dword readers and WRITE_SIZE
stay raw ("uncalibrated").  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): rocprofv3 sums
GRBM_GUI_ACTIVE over the 8 XCDs (checked against the kernel duration x 2.4 GHz)."""
import argparse
import collections
import csv
import glob
import json
import os

WIDE = ("corr7_", "conv3x3_direct", "conv1x1_kmajor", "mnn_tile", "conv2d_mfma", "stem")


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dirs", nargs="+", required=True)
    ap.add_argument("--cmd", default="")
    ap.add_argument("--out", required=True)
    ap.add_argument("--top", type=int, default=24)
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (r.get("Dispatch_Id"), r["Counter_Name"])
                if key not in seen:
                    seen.add(key)
                    cnt[k][r["Counter_Name"]] += 1
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            if dur and any(v[0] for v in dur.values()):
                continue                                     # durations from the first pass only
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                dur[k][0] += 1
                dur[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for k, c in agg.items():
        o = {}
        for name, v in c.items():
            n = max(cnt[k][name], 1)
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                o[name.lower() + "_bytes_per_launch_raw"] = v * 1024.0 / n
            else:
                o[name + "_per_launch"] = v / n
        if "fetch_size_bytes_per_launch_raw" in o:
            wide = any(w in k for w in WIDE)
            o["fetch_bytes_per_launch_corrected"] = o["fetch_size_bytes_per_launch_raw"] * (2.0 if wide else 1.0)
            o["fetch_correction"] = "x2 (16-byte/lane loads, gfx950 FETCH_SIZE counts them at half)" if wide else "none (dword loads: uncalibrated)"
        if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in o and o.get("GRBM_GUI_ACTIVE_per_launch"):
            o["mfma_busy"] = o["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / (o["GRBM_GUI_ACTIVE_per_launch"] / 8.0 * 1024.0)
        if "SQ_LDS_BANK_CONFLICT_per_launch" in o and o.get("SQ_ACTIVE_INST_ANY_per_launch"):
            o["lds_conflict_over_active_inst"] = o["SQ_LDS_BANK_CONFLICT_per_launch"] / o["SQ_ACTIVE_INST_ANY_per_launch"]
        if dur[k][0]:
            o["launches"] = dur[k][0]
            o["avg_us"] = dur[k][1] / dur[k][0] / 1e3
        out[k] = o
    ranked = sorted(out.items(), key=lambda kv: -(kv[1].get("avg_us", 0) * kv[1].get("launches", 0)))
    top = dict(ranked[:a.top] + [kv for kv in ranked[a.top:] if any(w in kv[0] for w in ("corr7", "mnn_", "ransac_", "stem"))])
    json.dump(dict(note=__doc__.split("\n\n")[1], command=a.cmd, passes=[os.path.basename(d.rstrip("/")) for d in a.dirs], kernels=top),
              open(a.out, "w"), indent=1)
    for k, o in list(top.items())[:10]:
        print("%-60s %5s x %9.1f us  fetch %8.1f MB (corr)  write %8.1f MB  mfma_busy %.3f  ldsconf/inst %.3f" % (
            k[:60], o.get("launches"), o.get("avg_us", 0), o.get("fetch_bytes_per_launch_corrected", 0) / 1e6,
            o.get("write_size_bytes_per_launch_raw", 0) / 1e6, o.get("mfma_busy", 0), o.get("lds_conflict_over_active_inst", 0)))


if __name__ == "__main__":
    main()

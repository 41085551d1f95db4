"""Turns the scratch outputs of scripts/gpu_profile_final.sh (gpurun_out/) into the committed, judged artefacts:
    profiles/r01_rocprofv3_kernel_stats_b<B>.csv   rocprofv3 --kernel-trace --stats summary of the bench command
    profiles/r01_pmc_summary_b<B>.json             per-kernel PMC counters (separate --pmc passes), per launch
usage: python scripts/make_profile_summary.py [B]"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
src = os.path.join(ROOT, "gpurun_out")
stats = open(os.path.join(src, "prof", "bench_kernel_stats.csv")).read().replace('"', "")
hdr = ("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch %d"
       "  (MI355X, round 1, final)\n" % B)
with open(os.path.join(ROOT, "profiles", "r01_rocprofv3_kernel_stats_b%d.csv" % B), "w") as f:
    f.write(hdr)
    for line in stats.splitlines():
        # kernel names contain commas: rocprofv3 quotes them; keep the file a plain CSV by using two spaces instead
        name, rest = line.rsplit(",", 7)[0], line.rsplit(",", 7)[1:]
        f.write(name.replace(",", " ") + "," + ",".join(rest[:6]) + "\n")
raw = json.load(open(os.path.join(src, "pmc_summary.json")))
kern = {}
for name, e in raw.items():
    n = e.get("pmc_run_launches")
    if not n:
        continue
    o = {"launches_in_pmc_run": n, "avg_ns_in_pmc_run": e["pmc_run_total_ns"] / n}
    if "FETCH_SIZE" in e:
        o["FETCH_SIZE_bytes_per_launch_raw"] = e["FETCH_SIZE"] * 1024.0 / e["launches_FETCH_SIZE"]
    if "WRITE_SIZE" in e:
        o["WRITE_SIZE_bytes_per_launch_raw"] = e["WRITE_SIZE"] * 1024.0 / e["launches_WRITE_SIZE"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        # summed over the 256 CUs x 4 SIMDs; normalised by the kernel's wall time at the nominal 2.4 GHz
        o["mfma_busy_frac_at_2.4GHz"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["pmc_run_total_ns"] * 2.4 * 1024)
    for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU",
              "SQ_LDS_BANK_CONFLICT"):
        if c in e:
            o[c] = e[c]
    kern[name] = o
note = ("rocprofv3 --pmc passes (separate runs, kernel-trace only) over `python bench.py --steps 1 --warmup 1 "
        "--no-cpu-baseline --batch %d` on MI355X, round 1. FETCH_SIZE/WRITE_SIZE in bytes (counter x 1024) per launch, "
        "RAW: on gfx950 FETCH_SIZE under-counts wide (16 B/lane) coalesced reads by exactly 2x (MI355X_MICROARCH.md); "
        "dword (4 B/lane) reads are uncalibrated." % B)
json.dump({"note": note, "kernels": kern}, open(os.path.join(ROOT, "profiles", "r01_pmc_summary_b%d.json" % B), "w"), indent=1)
print("kernels:", len(kern))
for k, o in sorted(kern.items(), key=lambda kv: -kv[1]["avg_ns_in_pmc_run"] * kv[1]["launches_in_pmc_run"])[:10]:
    print("%-60s launches %4d avg %8.1f us  mfma %.2f  fetch %.1f MB write %.1f MB" % (
        k[:60], o["launches_in_pmc_run"], o["avg_ns_in_pmc_run"] / 1e3, o.get("mfma_busy_frac_at_2.4GHz", 0),
        o.get("FETCH_SIZE_bytes_per_launch_raw", 0) / 1e6, o.get("WRITE_SIZE_bytes_per_launch_raw", 0) / 1e6))

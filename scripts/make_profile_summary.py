"""Turns the scratch outputs of scripts/gpu_final_r02.sh (gpurun_out/) into the committed, judged artefacts of a round:
    profiles/rNN_bench_default.log / rNN_bench_config{2,3,4,5}.log   bench JSON lines (+ stderr) as run
    profiles/rNN_rocprofv3_kernel_stats_<config>.csv                 rocprofv3 --kernel-trace --stats summary per bench config
    profiles/rNN_pmc_summary_qs.json                                 per-kernel PMC counters (separate --pmc passes), per launch
    profiles/rNN_parity_sweep_{qs,ev}_64pairs.json                   end-to-end parity sweeps (summary + per-pair records)
    profiles/rNN_corr_variants.json                                  correlation-kernel microbench incl. DMA-only / compute-only
usage: python scripts/make_profile_summary.py [round]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = "r%02d" % (int(sys.argv[1]) if len(sys.argv) > 1 else 2)
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")


def bench_log(name, cmd, out):
    log, err = os.path.join(src, name + ".log"), os.path.join(src, name + ".err")
    if not os.path.exists(log):
        return
    with open(os.path.join(dst, out), "w") as f:
        f.write("# %s   (MI355X, round %s)\n" % (cmd, RND[1:]))
        f.write(open(log).read())
        if os.path.exists(err):
            f.write("---- stderr ----\n" + "".join(l for l in open(err) if "amdgpu.ids" not in l))


bench_log("bench", "python bench.py --gpus 1 --steps 20 --warmup 5", RND + "_bench_default.log")
for c in ("2", "3", "4", "5", "qs"):
    bench_log("bench_c" + c, ("python bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline" if c == "2" else
                              "python bench.py --config %s --steps 5 --warmup 2 --no-cpu-baseline --no-qs-leg" % c), RND + "_bench_config%s.log" % c)
bench_log("bench_c3_2ranks_1gpu_gloo", "RFX_BENCH_BACKEND=gloo RFX_BENCH_DEVICE=0 python bench.py --config 3 --gpus 2 --steps 3 --warmup 1 --batch 32 "
          "--no-cpu-baseline  (two ranks rehearsed on ONE GPU)", RND + "_bench_config3_2ranks_on_1gpu_gloo.log")

for c in ("qs", "2", "3", "4", "5"):
    p = os.path.join(src, "kernel_stats_%s.csv" % c)
    if not os.path.exists(p):
        continue
    with open(os.path.join(dst, "%s_rocprofv3_kernel_stats_%s.csv" % (RND, "config" + c if c != "qs" else "qs")), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline "
                "--no-config3-leg  (MI355X, round %s; 3 steps in the trace incl. the warm-up)\n" % (c, RND[1:]))
        for line in open(p).read().replace('"', "").splitlines():
            name, rest = line.rsplit(",", 7)[0], line.rsplit(",", 7)[1:]
            f.write(name.replace(",", " ") + "," + ",".join(rest[:6]) + "\n")

for pm_name, pm_cfg in (("pmc_summary.json", "qs"), ("pmc_summary_config3.json", "config3"), ("pmc_summary_config5.json", "config5")):
  pm = os.path.join(src, pm_name)
  if os.path.exists(pm):
      raw = json.load(open(pm))
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
          for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT"):
              if c in e:
                  o[c] = e[c]
          kern[name] = o
      note = ("rocprofv3 --pmc passes (separate runs, kernel-trace only) over `python bench.py --steps 1 --warmup 1 --no-cpu-baseline "
              "--no-config3-leg` (config " + pm_cfg + ") on MI355X, round %s. FETCH_SIZE/WRITE_SIZE in bytes (counter x 1024) per launch, RAW: on gfx950 FETCH_SIZE "
              "under-counts wide (16 B/lane) coalesced reads by exactly 2x (MI355X_MICROARCH.md) -- double it for the LDS-DMA correlation "
              "kernel and the 16-byte conv loads; dword (4 B/lane) reads are uncalibrated." % RND[1:])
      json.dump({"note": note, "kernels": kern}, open(os.path.join(dst, RND + "_pmc_summary_" + pm_cfg + ".json"), "w"), indent=1)
      for k, o in sorted(kern.items(), key=lambda kv: -kv[1]["avg_ns_in_pmc_run"] * kv[1]["launches_in_pmc_run"])[:8]:
          print("%-70s launches %4d avg %8.1f us  mfma %.2f  fetch %.1f MB write %.1f MB" % (
              k[:70], o["launches_in_pmc_run"], o["avg_ns_in_pmc_run"] / 1e3, o.get("mfma_busy_frac_at_2.4GHz", 0),
              o.get("FETCH_SIZE_bytes_per_launch_raw", 0) / 1e6, o.get("WRITE_SIZE_bytes_per_launch_raw", 0) / 1e6))

for cfg, n in (("ev", 64), ("qs", 64)):
    p = os.path.join(src, "parity_sweep_%s_%d.json" % (cfg, n))
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_parity_sweep_%s_%dpairs.json" % (RND, cfg, n)))
for name, out in (("bench_parity_records_ev_loop.json", "_bench_parity_records_config3.json"), ("bench_parity_records_qs.json", "_bench_parity_records_qs.json")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, RND + out))
for name, out in (("conv_bench_all.txt", "_conv_bench.txt"), ("mfma_mix.txt", "_mfma_mix.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, RND + out))
p = os.path.join(src, "corr_variants.json")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, RND + "_corr_variants.json"))
print(sorted(f for f in os.listdir(dst) if f.startswith(RND)))

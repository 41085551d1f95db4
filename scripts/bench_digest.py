#!/usr/bin/env python
"""Prints the figures of one bench.py JSON line that a round's notes quote (evidence helper, no product code)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        j = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
    except (IndexError, OSError) as e:
        print(path, "no JSON line:", e)
        continue
    r = j.get("roofline", {})
    print("%s: %.2f pairs/s, %.2f ms/step | dominant %s %.1f TF frac %.3f share %.3f | all conv %.1f TF share %.3f"
          % (path, j["value"], j["ms_per_step"], r.get("kernel"), r.get("achieved", 0), r.get("frac", 0), r.get("time_share", 0),
             r.get("all_conv_tflops", 0), r.get("conv_time_share", 0)))
    c = j.get("roofline_corr")
    if c:
        print("  corr: frac %.3f (%.1f us)%s" % (c["frac"], c["avg_launch_us"], (" full-batch frac %.3f" % c["full_batch_launches"]["frac"]) if "full_batch_launches" in c else ""))
    for k, g in sorted(r.get("conv_kernels", {}).items(), key=lambda kv: -kv[1]["time_share"])[:8]:
        print("    kid %s: %.1f TF, share %.3f, %d launches" % (k, g["tflops"], g["time_share"], g["launches"]))
    cb = j.get("cpu_baseline")
    if cb:
        print("  cpu_baseline: %s pairs/s kind=%s cores=%s scan=%s" % (cb.get("value"), cb.get("kind"), cb.get("cores"), cb.get("threads_scan_s_per_pair")))
    for name, p in (("parity", j.get("parity")), ("qs parity", j.get("extra", {}).get("quick_start", {}).get("parity"))):
        if p:
            keys = ("oracle", "pairs", "identical_lists", "total_flipped_matches", "rounds", "rounds_exact_given_state", "degenerate_winner_rounds",
                    "max_H_delta", "max_flow12_delta", "free_run_same_nbH", "downstream_exact_given_matches", "max_flow_delta_e2e_identical",
                    "max_flow_delta_e2e", "oracle_wall_s", "error", "errors")
            print("  %s: %s" % (name, {k: p[k] for k in keys if k in p}))
    q = j.get("extra", {}).get("quick_start")
    if q:
        print("  quick_start leg: %.1f pairs/s, corr frac %.3f" % (q["value"], (q.get("roofline_corr") or {}).get("frac", 0)))

#!/usr/bin/env python
"""Test utility (GPU box): device-side traces of the multi-homography drivers at the sizes bench.py --config 3 / 4 / 5 times
(oracle/parity_sweep.py "ev_loop" / "c4" / "c5") written to gpurun_out/<tag>/dumps/<cfg>/pair_<seed>.npz, small enough to
travel back, so that the CPU oracle can replay every round OFFLINE (authoring container: `python oracle/parity_sweep.py
--config c4 --dump gpurun_out/<tag>/dumps/c4 --seeds 0 1 2 3 --records profiles/...`) instead of on GPU-box time.

    python tests/run_loop_dumps.py <tag> [cfg:n_pairs ...]        default: ev_loop:16 c4:4 c5:4
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ransac-flow_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    import torch
    import parity_sweep
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    jobs = [a.split(":") for a in sys.argv[2:]] or [("ev_loop", "16"), ("c4", "4"), ("c5", "4")]
    dev = torch.device("cuda:0")
    for cfg, n in jobs:
        d = os.path.join(ROOT, "gpurun_out", tag, "dumps", cfg)
        t0 = time.perf_counter()
        # every 8th pixel of the composed flow for the 64-pair sweep (the dump must travel back: <= 64 MiB in total)
        parity_sweep.dump_gpu_loop(cfg, list(range(int(n))), dev, d, batch=(16 if cfg == "ev_loop" else 4) if cfg != "c5" else 2,
                                   sub=8 if cfg == "ev_loop" else 4)
        sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        print("%s: %s pairs dumped to %s in %.1f s, %.1f MB" % (cfg, n, d, time.perf_counter() - t0, sz / 1e6), flush=True)


if __name__ == "__main__":
    main()

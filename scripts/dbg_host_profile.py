#!/usr/bin/env python
"""Experiments: where does the HOST time of a bench step go?  cProfile over a few steps of one bench config.
    python scripts/dbg_host_profile.py 4 [steps]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    sys.argv = ["bench.py", "--config", cfg]
    args = bench.parse_args() if hasattr(bench, "parse_args") else None
    dev = torch.device("cuda:0")
    step, meta, extra = bench.build_workload(args, dev, 0, 1)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    dt = (time.perf_counter() - t0) / steps
    print("config %s: %.1f ms per step" % (cfg, dt * 1e3))
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()

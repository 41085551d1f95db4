#!/usr/bin/env python
"""Samples the GPU clock / power (rocm-smi) while one convolution of the trunk runs back to back for a few seconds:
tells a matrix-pipe number measured under sustained load from the 2.4 GHz the 157.3 TFLOP/s peak assumes.

    python scripts/ubench/clock_probe.py [shape] [seconds]
"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts", "ubench"))
import torch  # noqa: E402
from rfx import ops  # noqa: E402
from conv_bench import SHAPES  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c128_120x160"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    N, Cin, H, W, Cout, k, stride, _ = SHAPES[name][:8]
    dev = torch.device("cuda:0")
    w = torch.randn(Cout, Cin, k, k) * 0.05
    plan = ops.ConvPlan(w, None, stride=stride, pad=k // 2, act=ops.ACT_RELU, device=dev)
    x = torch.randn(N, Cin, H, W, device=dev)
    plan(x)
    torch.cuda.synchronize()
    samples = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True,
                                     timeout=10).stdout
                keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "junction" in l.lower()) and "GPU[0]" in l]
                samples.append((time.time(), keep))
            except Exception as e:  # noqa: BLE001
                samples.append((time.time(), [repr(e)]))
            time.sleep(0.5)

    print("idle:")
    th = threading.Thread(target=poll)
    th.start()
    time.sleep(1.2)
    t0 = time.time()
    print("load starts")
    flops = 2.0 * N * (H // stride) * (W // stride) * Cout * Cin * k * k
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            plan(x)
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    stop.set()
    th.join()
    for ts, keep in samples:
        print("%+6.1f s  %s" % (ts - t0, " | ".join(keep)))
    print("%s: %d launches in %.1f ms -> %.1f TFLOP/s sustained" % (name, n, ms, flops * n / ms / 1e9))


if __name__ == "__main__":
    main()

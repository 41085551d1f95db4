#!/usr/bin/env python
"""Test utility (GPU box): the FULL end-to-end parity sweep -- all 64 bench seeds at 480x640 -- for one configuration
("qs" = quick_start semantics, what bench.py's default run also does; "ev" = evaluation semantics, variant B, nA = 13 065),
written to gpurun_out/parity_sweep_<cfg>_64.json for profiles/.  The pytest version (tests/test_gpu_parity_sweep.py) runs a
subset by default; this is the same code on every seed.

    python tests/run_parity_sweep.py ev [n_pairs [first_seed]]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ransac-flow_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    import torch
    import parity_sweep
    cfg = sys.argv[1] if len(sys.argv) > 1 else "ev"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # further seeds: more pairs for the flip statistics
    seeds = list(range(first, first + n))
    d = tempfile.mkdtemp(prefix="rfx_parity_")
    parity_sweep.dump_gpu_pairs(cfg, seeds, 480, 640, torch.device("cuda:0"), d)
    rec = os.path.join(ROOT, "gpurun_out", "parity_sweep_%s_%d%s.json" % (cfg, n, "_from%d" % first if first else ""))
    os.makedirs(os.path.dirname(rec), exist_ok=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "parity_sweep.py"), "--config", cfg, "--dump", d, "--seeds"]
                         + [str(s) for s in seeds] + ["--records", rec], capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:])
    sys.exit(out.returncode)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Experiments: config-5 step with the small-component filter on the device vs injected on the host (scipy), same box."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
import bench  # noqa: E402


def host_cc(match, match_th, cc_th):
    from scipy import ndimage
    lab, n = ndimage.label(match > match_th, structure=np.ones((3, 3), dtype=np.int32))
    if n == 0:
        return match
    area = np.bincount(lab.ravel(), minlength=n + 1) / float(lab.size)
    small = np.flatnonzero(area <= cc_th)
    small = small[small != 0]
    if small.size:
        match = match.copy()
        match[np.isin(lab, small)] = 0
    return match


sys.argv = ["bench.py", "--config", "5"]
args = bench.parse_args()
dev = torch.device("cuda:0")
step, meta, extra = bench.build_workload(args, dev, 0, 1)
pipe = extra["pipe"]
from rfx import synth  # noqa: E402
raws = [pipe.upload_raw([synth.make_pair(args.height, args.width, seed=s, homography=True, amp=0.02)]) for s in extra["seeds"]]
ra = (torch.cat([r[0] for r in raws]), torch.cat([r[1] for r in raws]))
for name, fn in (("device", None), ("host", host_cc), ("device", None), ("host", host_cc)):
    for _ in range(2):
        pipe.multi_h_kitti_batched(ra[0], ra[1], fineSize=650, maskRegionTh=0.005, cc_th=0.01, remove_small_cc=fn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        torch.manual_seed(0)
        o = pipe.multi_h_kitti_batched(ra[0], ra[1], fineSize=650, maskRegionTh=0.005, cc_th=0.01, remove_small_cc=fn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print("%-7s cc: %.1f ms per step, %.2f pairs/s, homographies per pair %s" % (name, dt * 1e3, len(ra[0]) / dt, [len(x["H"]) for x in o]))

#!/usr/bin/env python
"""Debug utility (GPU box): which Python lines of one config-3 step issue device copies / ATen kernels?  torch.profiler with
stacks around ONE multi-homography step; prints the memcpy / ATen-kernel launch counts grouped by the innermost frame that
lies in this repository.  (rocprofv3 counts them -- 425 of 1841 launches in r03 -- but cannot say where they come from.)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from rfx import weights, synth, ops  # noqa: E402
from rfx.pipeline import AlignPipeline  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
           match=weights.net_matchability_sd(3, last_std=3.0))
pipe = AlignPipeline(sds, nbScale=7, nbIter=10000, tolerance=0.05, minSize=480, scaleR=2.0, variant="B", device=dev)
raw = pipe.upload_raw([synth.make_pair(480, 640, seed=s, homography=True) for s in range(B)])


def step():
    prep = pipe.prepare_device(*raw)
    R = ops.MultiHRecords(B, 60, 80, dev)
    pipe.multi_h_batched(prep, records=R, want_lists=False)
    return R.rec


step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
sites = collections.Counter()
kinds = collections.Counter()
for ev in prof.events():
    name = ev.name
    if not (name.startswith("aten::") or "Memcpy" in name or "memcpy" in name):
        continue
    if name.startswith("aten::") and ev.device_time_total == 0 and not ev.kernels:
        continue
    frame = next((f for f in (ev.stack or []) if "/ransac-flow_amd/" in f or "bench.py" in f), "?")
    sites[(name, frame.split("ransac-flow_amd/")[-1][:90])] += 1
    kinds[name] += 1
for (name, frame), c in sites.most_common(40):
    print("%4d  %-28s %s" % (c, name, frame))
print(kinds.most_common(20))

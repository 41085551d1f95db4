#!/usr/bin/env python
"""Times rfx_remove_small_cc_f32 on KITTI-shaped matchability maps (8 x 376 x 1242): mostly-foreground blobs (the real
case: the matched region is one large component with holes), sparse blobs, all foreground."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from rfx import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
N, H, W = 8, 376, 1242
n = torch.randn(N, 1, H, W, generator=g)
k = 25
ax = torch.arange(k) - k // 2
ker = torch.exp(-ax.float() ** 2 / (2 * 6.0 ** 2))
ker = (ker[:, None] * ker[None, :]) / ker.sum() ** 2
sm = F.conv2d(n, ker[None, None], padding=k // 2)[:, 0]
sm = sm / sm.std()
for name, m in (("mostly foreground", (sm > -1.0).float()), ("half", (sm > 0).float()), ("sparse", (sm > 1.5).float()),
                ("all foreground", torch.ones(N, H, W))):
    md = m.to(dev).contiguous()
    for _ in range(3):
        ops.remove_small_cc(md, 0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.remove_small_cc(md, 0.01)
    e1.record()
    torch.cuda.synchronize()
    print("%-20s fg %.2f  %.3f ms per call (8 x 376 x 1242)" % (name, float(m.mean()), e0.elapsed_time(e1) / 10))

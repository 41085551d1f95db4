"""Experiment: time individual non-conv ops at the shapes of the default bench (B=64)."""
import sys
sys.path.insert(0, "ransac-flow_amd")
import torch
from rfx import ops
dev = "cuda"

def timeit(name, fn, bytes_moved, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-44s %8.3f ms  %6.2f TB/s" % (name, ms, bytes_moved / ms / 1e9))

x = torch.randn(64, 64, 480, 640, device=dev)
timeit("maxblurpool 64x64x480x640 s2", lambda: ops.maxblurpool2d(x, 2), x.numel() * 4 * 1.25)
del x
x = torch.randn(64, 64, 288, 384, device=dev)
timeit("maxpool3x3s2 64x64x288x384", lambda: ops.maxpool2d(x, 3, 2, 1), x.numel() * 4 * 1.25)
del x
x = torch.randn(64, 1024, 36, 48, device=dev)
timeit("l2norm 64x1024x36x48", lambda: ops.l2norm(x), x.numel() * 8)
x = torch.randn(128, 256, 60, 80, device=dev)
timeit("l2norm 128x256x60x80", lambda: ops.l2norm(x), x.numel() * 8)
x = torch.randn(64, 128, 120, 160, device=dev)
timeit("blurpool s2 64x128x120x160", lambda: ops.blurpool2d(x, 2), x.numel() * 4 * 1.25)

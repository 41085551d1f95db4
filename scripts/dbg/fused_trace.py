"""Experiment: phase stamps of the fused bottleneck-tail kernel (library built with `make trace`)."""
import ctypes, sys
sys.path.insert(0, "ransac-flow_amd")
import torch
from rfx import _lib, ops
lib = _lib.load()
dev = "cuda"
for (N, Cin, H, W, Cmid, Cexp) in [(64, 64, 120, 160, 64, 256), (64, 128, 60, 80, 128, 512)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Cin, H, W, device=dev)
    p2 = ops.ConvPlan(torch.randn(Cmid, Cin, 3, 3) * 0.05, dict(weight=torch.ones(Cmid), bias=torch.zeros(Cmid), running_mean=torch.zeros(Cmid), running_var=torch.ones(Cmid)), 1, 1, ops.ACT_RELU, dev)
    p3 = ops.ConvPlan(torch.randn(Cexp, Cmid, 1, 1) * 0.05, dict(weight=torch.ones(Cexp), bias=torch.zeros(Cexp), running_mean=torch.zeros(Cexp), running_var=torch.ones(Cexp)), 1, 0, ops.ACT_RELU, dev)
    r = torch.randn(N, Cexp, H, W, device=dev)
    import os
    if os.environ.get("NORES"): r = None
    for _ in range(2): y = ops.bottleneck_tail(x, p2, p3, residual=r)
    torch.cuda.synchronize()
    trace = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    lib.rfx_debug_trace.argtypes = [ctypes.c_void_p]
    lib.rfx_debug_trace(ctypes.c_void_p(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = ops.bottleneck_tail(x, p2, p3, residual=r); e1.record(); torch.cuda.synchronize()
    lib.rfx_debug_trace(ctypes.c_void_p(0))
    t = trace.cpu().view(-1, 4); t = t[t[:, 0] > 0].double(); tick = 1e-2
    pro, main, epi = (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick, (t[:, 3] - t[:, 2]) * tick
    span = (t[:, 3].max() - t[:, 0].min()) * tick
    fl = 2.0 * N * H * W * (Cmid * Cin * 9 + Cexp * Cmid)
    print((N, Cin, H, W, Cmid, Cexp), "wgs", t.shape[0], "ms %.3f" % e0.elapsed_time(e1), "TF %.1f" % (fl / span / 1e6),
          "per-WG us: prologue %.1f main %.1f phaseB %.1f" % (pro.mean(), main.mean(), epi.mean()),
          "concurrency %.0f" % (t.shape[0] * (pro + main + epi).mean() / span))

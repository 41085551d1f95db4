"""Experiment: per-workgroup phase timing of conv2d_mfma_kernel (library built with `make trace`).
usage: RFX_LIB=ransac-flow_amd/librfx_trace.so python scripts/dbg/conv_trace.py"""
import ctypes, os, sys
sys.path.insert(0, "ransac-flow_amd")
import torch
from rfx import _lib, ops

lib = _lib.load()
dev = "cuda"
SHAPES = [  # N, Cin, H, W, Cout, k, stride, pad, residual
    (64, 128, 60, 80, 512, 1, 1, 0, True),
    (64, 256, 30, 40, 1024, 1, 1, 0, True),
    (64, 64, 120, 160, 256, 1, 1, 0, True),
    (64, 1024, 30, 40, 256, 1, 1, 0, False),
    (64, 512, 60, 80, 128, 1, 1, 0, False),
    (128, 128, 120, 160, 128, 3, 1, 1, False),
    (128, 64, 240, 320, 64, 3, 1, 1, True),
    (64, 256, 30, 40, 256, 3, 1, 1, False),
    (64, 256, 25, 33, 256, 3, 1, 1, False),
]
for (N, Cin, H, W, Cout, k, s, p, res) in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, k, k) * 0.05
    plan = ops.ConvPlan(w, None, s, p, ops.ACT_RELU, dev)
    Ho, Wo = plan.out_hw(H, W) if hasattr(plan, "out_hw") else ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    r = torch.randn(N, Cout, Ho, Wo, device=dev) if res else None
    for _ in range(2):
        y = plan(x, residual=r)
    torch.cuda.synchronize()
    trace = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    if not hasattr(lib, "rfx_debug_trace"):      # plain library: event timing only
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): y = plan(x, residual=r)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("shape", (N, Cin, H, W, Cout, k, s), "kid", lib.rfx_conv2d_kernel_id(N, Cin, Cout, k, k, s, p, Ho, Wo),
              "ms %.3f  TF %.1f" % (ms, 2.0 * N * Ho * Wo * Cout * Cin * k * k / ms / 1e9))
        continue
    lib.rfx_debug_trace.argtypes = [ctypes.c_void_p]
    lib.rfx_debug_trace(ctypes.c_void_p(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = plan(x, residual=r); e1.record()
    torch.cuda.synchronize()
    lib.rfx_debug_trace(ctypes.c_void_p(0))
    t = trace.cpu().view(-1, 4)
    t = t[t[:, 0] > 0].double()
    nwg = t.shape[0]
    tick = 1e-8  # wall_clock64: 100 MHz
    t0 = t[:, 0].min()
    pro, main, epi = (t[:, 1] - t[:, 0]) * tick * 1e6, (t[:, 2] - t[:, 1]) * tick * 1e6, (t[:, 3] - t[:, 2]) * tick * 1e6
    span = (t[:, 3].max() - t0) * tick * 1e6
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    kid = lib.rfx_conv2d_kernel_id(N, Cin, Cout, k, k, s, p, Ho, Wo)
    print("shape", (N, Cin, H, W, Cout, k, s), "kid", kid, "wgs", nwg, "event ms %.3f" % e0.elapsed_time(e1),
          "span us %.0f" % span, "TF %.1f" % (flops / (span * 1e-6) / 1e12))
    print("   per-WG us: prologue %.2f (p90 %.2f)  main %.2f (p90 %.2f)  epilogue %.2f (p90 %.2f)  total %.2f" % (
        pro.mean(), pro.quantile(0.9), main.mean(), main.quantile(0.9), epi.mean(), epi.quantile(0.9),
        (pro + main + epi).mean()))
    conc = nwg * (pro + main + epi).mean() / span
    print("   mean concurrency %.0f WGs (512 slots)" % conc)

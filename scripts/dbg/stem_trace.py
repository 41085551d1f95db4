"""Experiment: per-workgroup phase timing of stem7_conv_maxpool_kernel (library built with `make trace`): s_memtime stamps at
0 start | 1 patch in LDS | 2 after barrier | 3 MFMA phase + C tile done | 4 after barrier | 5 pooled + stored.
usage: RFX_LIB=ransac-flow_amd/librfx_trace.so python scripts/dbg/stem_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "ransac-flow_amd"))
import torch
from rfx import _lib, ops, weights
from rfx.ops import ConvPlan, ACT_RELU

lib = _lib.load()
dev = torch.device("cuda:0")
sd = weights.resnet50_trunk_sd(0, randomize_bn=True)
plan = ConvPlan(sd["conv1.weight"], {k: sd["bn1." + k] for k in ("weight", "bias", "running_mean", "running_var")}, 2, 3, ACT_RELU, dev)
lib.rfx_debug_trace.argtypes = [ctypes.c_void_p]
for (N, H, W) in ((64, 480, 640), (64, 960, 1280)):
    x = torch.randn(N, 3, H, W, device=dev)
    for _ in range(3):
        ops.stem_conv7_maxpool(x, plan)
    torch.cuda.synchronize()
    trace = torch.zeros(1 << 23, dtype=torch.int64, device=dev)
    lib.rfx_debug_trace(ctypes.c_void_p(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.stem_conv7_maxpool(x, plan); e1.record()
    torch.cuda.synchronize()
    lib.rfx_debug_trace(ctypes.c_void_p(0))
    t = trace.cpu().view(-1, 8)
    t = t[t[:, 5] > 0].double()
    d = t[:, 1:6] - t[:, 0:5]
    names = ["load+stage", "barrier1", "mfma+ctile", "barrier2", "pool+store"]
    tot = (t[:, 5] - t[:, 0])
    print("%dx%dx%d: %d workgroups, event %.3f ms, ticks per workgroup: total %.0f (p90 %.0f)" % (N, H, W, t.shape[0], e0.elapsed_time(e1), tot.mean(), tot.quantile(0.9)))
    for k, nm in enumerate(names):
        print("   %-12s mean %8.0f  p10 %8.0f  p90 %8.0f  (%.1f %%)" % (nm, d[:, k].mean(), d[:, k].quantile(0.1), d[:, k].quantile(0.9), 100 * d[:, k].mean() / tot.mean()))
    span = t[:, 5].max() - t[:, 0].min()
    print("   span %.0f ticks -> %.3f GHz-equivalent tick rate; concurrency %.1f workgroups in flight (512 slots)" % (span, span / (e0.elapsed_time(e1) * 1e6), float(tot.sum() / span)))

#!/usr/bin/env python
"""bench.py -- aligned image pairs / s at 480x640 on N MI355X (one process per GPU).

A "step" = one pass of the whole hot path over one batch of synthetic pairs already resident in HBM:
ResNet-50 conv4 features of the 7-level source pyramid + target -> L2 norm -> all-pairs correlation +
mutual NN -> 4-point DLT RANSAC (nbIter hypotheses) -> homography grid -> warp -> FeatureExtractor x2 ->
7x7 local correlation -> NetFlowCoarse -> flow composition -> final warp  (the quick_start/align2images.py
path, BASELINE configs 2+3 at the metric's resolution).  float32 end to end.

N > 1: launched by torch.distributed.run, one rank per GPU over RCCL; every rank aligns its own batch per
step (pairs shard embarrassingly: weak scaling) and the per-pair result records (H + flowDown8) are
collected with ONE all_gather per step.

Prints one JSON line on rank 0 (see the driver's contract): value = pairs/s over all ranks, plus
  roofline     -- dominant kernel (conv2d_mfma_kernel<2,2>, fp32 MFMA bound): algorithmic FLOP per launch /
                  average launch duration, measured with HIP events on the launch stream inside the timed
                  region; roofline_corr -- same for the HBM-bound 7x7 correlation kernel;
  cpu_baseline -- the CPU oracle (oracle/restate.py, a port of the reference path) on this host's cores,
                  rank 0 / N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))

# the host driver supports only dmabuf IPC: without this RCCL / cross-process tensor sharing fails with
# "hipIpcGetMemHandle: invalid argument" (already exported on the build and GPU boxes; kept for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md (dense fp32 matrix peak)
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


def cpu_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, os.cpu_count() or 1, 64))


def cpu_baseline_subprocess(args):
    """Runs the CPU leg in a child process with a hard wall-clock limit so that it can never stall the bench."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--height", str(args.height), "--width",
           str(args.width), "--nb-scale", str(args.nb_scale), "--nb-iter", str(args.nb_iter), "--cpu-pairs",
           str(args.cpu_pairs)]
    if getattr(args, "parity_file", None):
        cmd += ["--parity-file", args.parity_file]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
        for ln in out.stdout.splitlines():
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "pairs/s", "cores": cpu_threads(), "kind": "port",
                "sample": "cpu leg failed: " + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "pairs/s", "cores": cpu_threads(), "kind": "port",
                "sample": "cpu leg exceeded its 150 s limit"}


def cpu_baseline(sds, args):
    """Bounded CPU sample: the oracle restatement on `cpu_pairs` pairs of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate
    from rfx import synth
    torch.set_num_threads(cpu_threads())
    H, W = args.height, args.width
    ca = restate.CoarseAlignOracle(sds["trunk"], args.nb_scale, args.nb_iter, 0.05, max(H, W), 1.2, variant="A")
    nets = dict(feat=sds["feat"], flow=sds["flow"])

    def one(seed):
        I1, I2 = synth.make_pair(H, W, seed=seed)
        ca.setSource(I1)
        ca.setTarget(I2)
        r = ca.getCoarse(np.zeros((ca.It.size[1], ca.It.size[0])))
        with torch.no_grad():
            fc = restate.warp_grid(torch.from_numpy(r["H"])[None], ca.It.size[1], ca.It.size[0])
            restate.fine_step_quickstart(nets, ca.IsTensor, ca.ItTensor, fc)

    t0 = time.perf_counter()
    one(1000)  # warm-up
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = 0
    while n < args.cpu_pairs:
        one(2000 + n)
        n += 1
        if time.perf_counter() - t0 > 25 or (n == 1 and warm > 30):
            break
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d synthetic %dx%d pairs, full coarse+fine path (oracle/restate.py), %.1f s" % (n, H, W, dt)}
    if getattr(args, "parity_file", None) and os.path.exists(args.parity_file):
        out["parity"] = parity_vs_oracle(ca, nets, args)
    return out


def parity_vs_oracle(ca, nets, args):
    """Full-size parity of ONE pair of the timed workload: the GPU leg left its results for pair 0 in an npz; the
    oracle aligns the same synthetic pair with the same RANSAC index draw (checker only, never timed)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate
    from rfx import synth
    g = np.load(args.parity_file)
    I1, I2 = synth.make_pair(args.height, args.width, seed=int(g["seed"]))
    ca.sample_fn = lambda n, it: torch.from_numpy(g["samples"])
    ca.setSource(I1)
    ca.setTarget(I2)
    r = ca.getCoarse(np.zeros((ca.It.size[1], ca.It.size[0])))
    same = bool(np.array_equal(r["index1"], g["index1"]) and np.array_equal(r["index2"], g["index2"]))
    ref_set = set(zip(r["index1"].tolist(), r["index2"].tolist()))
    got_set = set(zip(g["index1"].tolist(), g["index2"].tolist()))
    res = {"pair": "synthetic %dx%d seed %d" % (args.height, args.width, int(g["seed"])), "n_matches": int(len(r["index1"])),
           "match_list_identical": same, "n_matches_differing": len(ref_set ^ got_set)}
    if same:
        res["inlier_indices_bit_exact"] = bool(np.array_equal(r["inlier"], g["inlier"]))
        res["max_abs_H_delta"] = float(np.abs(r["H"] - g["H"]).max())
    with torch.no_grad():
        h, w = ca.It.size[1], ca.It.size[0]
        st = restate.fine_step_quickstart(nets, ca.IsTensor, ca.ItTensor, restate.warp_grid(torch.from_numpy(g["H"])[None], h, w))
    res["max_abs_flow_delta"] = float((st["flow12"].numpy() - g["flow12"]).__abs__().max())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="pairs per step per GPU (BASELINE config 3: batch of 64)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--nb-scale", type=int, default=7)
    ap.add_argument("--nb-iter", type=int, default=1000)
    ap.add_argument("--multi-h", action="store_true",
                    help="BASELINE config 3 as literally worded: variant B (evalHpatch) multi-homography loop, coarseIter "
                         "10 000, minSize 480, scaleR 2, PredFlowMask per homography (default: quick_start semantics, one H)")
    ap.add_argument("--cpu-pairs", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-prep", action="store_true",
                    help="build the LANCZOS pyramid with PIL on the host before the timed region (default: raw uint8 images "
                         "resident in HBM, pyramid + ToTensor + Normalize on the device inside the timed step)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--parity-file", type=str, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_only:
        from rfx import weights
        sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1),
                   flow=weights.net_flow_coarse_sd(2))
        print(json.dumps(cpu_baseline(sds, args)))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (there is no CPU fallback)")
    # RFX_BENCH_DEVICE / RFX_BENCH_BACKEND exist only to rehearse the N>1 control flow on a 1-GPU box (all ranks on
    # one device, gloo instead of RCCL); the driver's multi-GPU runs use the defaults: one GPU per rank, RCCL.
    dev_index = int(os.environ.get("RFX_BENCH_DEVICE", local_rank))
    backend = os.environ.get("RFX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from rfx import weights, synth, ops
    from rfx.pipeline import AlignPipeline
    from rfx import dist as rdist

    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1),
               flow=weights.net_flow_coarse_sd(2), match=weights.net_matchability_sd(3))
    H, W, B = args.height, args.width, args.batch
    if args.multi_h:
        sds["match"] = weights.net_matchability_sd(3, last_std=0.02)
        pipe = AlignPipeline(sds, nbScale=args.nb_scale, nbIter=10000, tolerance=0.05, minSize=min(H, W), scaleR=2.0,
                             variant="B", device=dev)
        args.no_cpu_baseline = True     # the CPU leg times the quick_start path
    else:
        pipe = AlignPipeline(sds, nbScale=args.nb_scale, nbIter=args.nb_iter, tolerance=0.05, minSize=max(H, W), scaleR=1.2,
                             variant="A", device=dev)
    # this rank's shard of the synthetic stream: pair i -> rank i mod world
    pairs = [synth.make_pair(H, W, seed=rank + world * i) for i in range(B)]
    log("weights packed, synthetic pairs made")
    if args.host_prep:
        prep = pipe.prepare(pairs)      # host PIL pyramid + upload: outside the timed region
        raw = None
    else:
        raw = pipe.upload_raw(pairs)    # raw uint8 images resident in HBM; the pyramid is part of the timed step
        prep = pipe.prepare_device(*raw)
    torch.manual_seed(123 + rank)
    log("inputs resident on %s" % dev)

    def step():
        p = prep if raw is None else pipe.prepare_device(*raw)
        if args.multi_h:
            outs = pipe.multi_h_batched(p, maxCoarse=10, maskRegionTh=0.01)
            rec = torch.zeros((B, 10 + 11 * 9), dtype=torch.float32, device=dev)   # [9 unused | status | nbH-1 H matrices]
            for b, o in enumerate(outs):
                rec[b, 9] = 0.0 if o["H"] else 1.0
                if o["H"]:
                    hs = torch.stack(o["H"]).reshape(-1)
                    rec[b, 10:10 + hs.numel()] = hs
            return rdist.gather_records(rec, dist)
        res = pipe.align_prepared(p, fine=True)
        rec = rdist.pack_records(res)                       # (B, 9 + 1 + 2*h8*w8) float32 on device
        return rdist.gather_records(rec, dist)              # ONE all_gather per step (no-op copy when world == 1)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warmup step %d done" % i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    with ops.Profiler() as prof:        # explicit, thread-local: two HIP events per conv / corr launch, on the launch stream
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    log("%d timed steps: %.3f s" % (args.steps, elapsed))
    conv_t, corr_t = prof.conv, prof.corr
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok_pairs = int((out[:, 9] == 0).sum().item())

    # ---- roofline of the dominant kernel (events were recorded on the launch stream inside the timed region)
    allc = [(f, e0.elapsed_time(e1) * 1e-3) for (v, f, e0, e1, _, _) in conv_t]
    by_kernel = {}
    for (v, f, e0, e1, _, nb) in conv_t:
        g = by_kernel.setdefault(v, [0, 0.0, 0.0, 0.0])
        g[0] += 1; g[1] += f; g[2] += e0.elapsed_time(e1) * 1e-3; g[3] += nb
    dom_id = max(by_kernel, key=lambda k: by_kernel[k][2])          # the kernel instance with the most GPU time
    dom_n, dom_f, dom_t, dom_b = by_kernel[dom_id]
    tmn = {0: "2, 2", 1: "1, 2", 2: "1, 1"}[dom_id & 3]
    tf = lambda b: "true" if b else "false"
    if dom_id & (32 | 512):   # direct 3x3 kernel; 512 = fused with the 1x1 expansion (Bottleneck tail)
        dom_name = "conv3x3_direct_kernel<%d, %d, %s>" % (1 if dom_id & 3 else 2, 4 if dom_id & 128 else (8 if dom_id & 64 else 16),
                                                          "true" if dom_id & 512 else "false")
    elif dom_id == 256:
        dom_name = "stem_conv_maxblur_kernel"
    elif dom_id == 257:
        dom_name = "stem7_conv_maxpool_kernel"
    else:
        dom_name = "conv2d_mfma_kernel<%s, %s, %s, %s>" % (tmn, tf(dom_id & 4), tf(dom_id & 8), tf(dom_id & 16))
    if os.environ.get("RFX_BENCH_DUMP") and rank == 0:
        agg = {}
        for (v, f, e0, e1, shp, _) in conv_t:
            a = agg.setdefault((v,) + shp, [0, 0.0, 0.0])
            a[0] += 1; a[1] += f; a[2] += e0.elapsed_time(e1) * 1e-3
        rows = sorted(((k, c, f, t) for k, (c, f, t) in agg.items()), key=lambda r: -r[3])
        with open(os.environ["RFX_BENCH_DUMP"], "w") as fh:
            fh.write("variant,N,Cin,H,W,Cout,k,stride,calls,total_ms,TFLOPs,share\n")
            tot = sum(r[3] for r in rows)
            for k, c, f, t in rows:
                fh.write(",".join(str(x) for x in k) + ",%d,%.3f,%.1f,%.3f\n" % (c, t * 1e3, f / t / 1e12, t / tot))
    flops_per_launch = dom_f / dom_n
    avg_dur = dom_t / dom_n
    ach = flops_per_launch / avg_dur / 1e12
    roofline = {"kernel": dom_name, "bound": "mfma", "achieved": round(ach, 2),
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": None, "launches": dom_n, "avg_launch_us": round(avg_dur * 1e6, 2),
                "flop_per_launch": flops_per_launch, "time_share": round(dom_t / elapsed, 3),
                "all_conv_tflops": round(sum(f for f, _ in allc) / sum(d for _, d in allc) / 1e12, 2),
                "conv_time_share": round(sum(d for _, d in allc) / elapsed, 3),
                "conv_kernels": {("%d" % k): {"launches": n, "tflops": round(f / t / 1e12, 1), "time_share": round(t / elapsed, 3)}
                                 for k, (n, f, t, _) in sorted(by_kernel.items())}}
    # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the bench)
    roofline["algorithmic_bytes_per_launch"] = round(dom_b / dom_n)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary_b64.json")))["kernels"]
        for name, e in pmc.items():
            if dom_name in name and B == 64:
                roofline["traffic"] = round(e["FETCH_SIZE_bytes_per_launch_raw"] + e["WRITE_SIZE_bytes_per_launch_raw"])
                roofline["traffic_note"] = ("FETCH_SIZE+WRITE_SIZE (x1024 B) per launch from profiles/r01_pmc_summary_b64.json, raw: "
                                            "4-byte/lane reads are uncalibrated on gfx950 (16-byte/lane reads under-count 2x)")
                roofline["mfma_busy_frac_pmc"] = round(e.get("mfma_busy_frac_at_2.4GHz", 0.0), 3)
            if "corr7_dma_kernel" in name and B == 64:
                corr_pmc = e
    except Exception:
        corr_pmc = None
    cb = sum(b for b, _, _ in corr_t) / len(corr_t)
    cd = sum(e0.elapsed_time(e1) * 1e-3 for _, e0, e1 in corr_t) / len(corr_t)
    roofline_corr = {"kernel": "corr7_dma_kernel", "bound": "hbm", "achieved": round(cb / cd / 1e9, 1),
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(cb / cd / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                     "bytes_per_launch": cb, "avg_launch_us": round(cd * 1e6, 2)}
    if locals().get("corr_pmc"):
        # 16-byte/lane LDS-DMA reads: FETCH_SIZE counts exactly half the bytes on gfx950 (MI355X_MICROARCH.md) -> x2
        roofline_corr["traffic"] = round(2 * corr_pmc["FETCH_SIZE_bytes_per_launch_raw"] + corr_pmc["WRITE_SIZE_bytes_per_launch_raw"])

    if rank == 0:
        total_pairs = B * args.steps * world
        line = {
            "metric": "aligned image-pairs/sec @480\u00d7640, 1/2/4/8 MI355X; max-abs flow \u0394 vs ref", "value": round(total_pairs / elapsed, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("MULTI-H (variant B, 10 000 RANSAC iterations and PredFlowMask per homography, up to 11 "
                                    "homographies per pair): " if args.multi_h else "") +
                                   "batch of %d %dx%d pairs per GPU per step, full pipeline: ResNet-50 conv4 feat x%d scales"
                                   " + mutual NN + RANSAC(nbIter=%d, 4-pt DLT) + FeatureExtractor + 7x7 corr + NetFlowCoarse"
                                   " + grid_sample (BASELINE configs 2+3 at the metric's 480x640)" % (B, H, W, args.nb_scale, args.nb_iter),
                       "pairs_per_step_per_gpu": B, "nbIter": args.nb_iter, "nbScale": args.nb_scale,
                       "parallelism": "pairs sharded over %d rank(s), one all_gather of result records per step" % world,
                       "weights": "random-init", "aligned_ok_last_step": ok_pairs,
                       "preprocessing": "host PIL, outside the timed region" if args.host_prep else
                       "device (bit-exact Pillow LANCZOS pyramid + ToTensor/Normalize), inside the timed step"},
            "roofline": roofline, "roofline_corr": roofline_corr,
        }
        if world == 1 and not args.no_cpu_baseline:
            # leave pair 0's GPU result for the checker (one extra, untimed pass with a recorded index draw)
            import tempfile
            r0 = pipe.align_prepared(prep, fine=True)[0]
            if r0["H"] is not None:
                args.parity_file = os.path.join(tempfile.mkdtemp(prefix="rfx_parity_"), "pair0.npz")
                np.savez(args.parity_file, seed=rank, samples=r0["samples"].numpy(), index1=r0["index1"].cpu().numpy(),
                         index2=r0["index2"].cpu().numpy(), inlier=r0["inlier"].cpu().numpy(), H=r0["H"].cpu().numpy(),
                         flow12=r0["flow12"].cpu().numpy())
            log("GPU leg done; timing the CPU oracle (bounded sample, child process)")
            cb = cpu_baseline_subprocess(args)
            if "parity" in cb:
                line["parity"] = cb.pop("parity")
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

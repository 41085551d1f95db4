#!/usr/bin/env python
"""TEST / EVIDENCE UTILITY (authoring container only: needs /root/reference) -- calibrates bench.py's cpu_baseline.

bench.py times ``oracle/restate.py`` (kind = "port") on the GPU box, because the reference tree does not exist there.  This
script times the REAL reference -- quick_start/coarseAlignFeatMatch.py + utils/outil.py + model/model.py executed from where
they lie under /root/reference by oracle/ref_loader.py (CUDA calls stubbed to CPU) -- and the port back to back on the same
cores, same pairs, same index draws, for the two workloads bench.py reports:

  qs   quick_start/align2images.py:53-97 semantics at 480x640 (7 scales x1.2, nbIter 1000, one homography, fine flow)
  ev   the multi-homography loop of evaluation/evalHpatch/evaluation.py:211-243 (variant B, 7 scales x2, coarseIter 10 000,
       maxCoarse 10, maskRegionTh 0.01) -- the reference's own ``while`` statement compiled out of the script

and writes profiles/r03_cpu_reference_vs_port.json with the reference/port time ratio, which is what turns a "port" pairs/s
measured on the GPU box into an estimate of the reference's own CPU pairs/s on those cores.

    python scripts/cpu_reference_vs_port.py [--pairs 3] [--threads 8] [--out profiles/r03_cpu_reference_vs_port.json]
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ransac-flow_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import ref_loader  # noqa: E402
import restate  # noqa: E402
from rfx import weights, synth  # noqa: E402

MATCH_STD = 3.0


def _load_trunk(ca):
    names = ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3"]
    remap = {}
    for k, v in weights.resnet50_trunk_sd(seed=0).items():
        top, rest = k.split(".", 1)
        remap["%d.%s" % (names.index(top), rest)] = v
    ca.net.load_state_dict(remap)
    ca.net.eval()
    return ca


def _ref_nets(R, match_std=None):
    model = R["model"]
    fe = ref_loader.quiet(model.FeatureExtractor)
    fe.load_state_dict(weights.feature_extractor_sd(seed=1))
    nf = ref_loader.quiet(model.NetFlowCoarse, 7)
    nf.load_state_dict(weights.net_flow_coarse_sd(seed=2))
    nm = ref_loader.quiet(model.NetMatchability, 7)
    nm.load_state_dict(weights.net_matchability_sd(3) if match_std is None else weights.net_matchability_sd(3, last_std=match_std))
    net = {"netFeatCoarse": fe, "netCorr": model.CorrNeigh(7), "netFlowCoarse": nf, "netMatch": nm}
    for m in net.values():
        m.eval()
    return net


def _grid(h, w):
    return torch.cat((torch.linspace(-1, 1, w).view(1, 1, -1, 1).expand(1, h, w, 1),
                      torch.linspace(-1, 1, h).view(1, -1, 1, 1).expand(1, h, w, 1)), dim=3)


def time_pairs(fn, seeds):
    fn(999)                                     # warm-up (oneDNN primitive caches, lazy imports)
    ts = []
    for s in seeds:
        t0 = time.perf_counter()
        fn(s)
        ts.append(time.perf_counter() - t0)
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cpu_reference_vs_port.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    R = ref_loader.load()
    H, W = 480, 640
    seeds = list(range(2000, 2000 + a.pairs))
    out = dict(threads=a.threads, pairs=a.pairs, size="%dx%d" % (H, W), host=os.popen("lscpu | grep 'Model name'").read().strip(),
               note="reference = the reference's own modules / script loop run in place by oracle/ref_loader.py (CUDA calls stubbed to "
                    "CPU, kornia warp_grid and torchvision stubs as in SURVEY A.2); port = oracle/restate.py; same pairs, same "
                    "generator seeds, back to back on the same cores")

    # ---------------- quick_start semantics
    net = _ref_nets(R)
    caA = _load_trunk(ref_loader.quiet(R["CoarseAlignA"], 7, 1000, 0.05, "Homography", 640, scaleR=1.2))
    warper = R["kornia_geometry"].HomographyWarper(H, W)
    grid = _grid(H, W)

    def ref_qs(seed):
        I1, I2 = synth.make_pair(H, W, seed=seed)
        torch.manual_seed(seed)
        with torch.no_grad():
            caA.setSource(I1)
            caA.setTarget(I2)
            bestPara, _ = caA.getCoarse(np.zeros((caA.It.size[1], caA.It.size[0])))
            flowCoarse = warper.warp_grid(torch.from_numpy(bestPara).unsqueeze(0))
            img1 = F.grid_sample(caA.IsTensor, flowCoarse)
            f1, f2 = F.normalize(net["netFeatCoarse"](img1)), F.normalize(net["netFeatCoarse"](caA.ItTensor))
            flowDown = net["netFlowCoarse"](net["netCorr"](f1, f2), False)
            flowUp = F.interpolate(flowDown, size=(H, W), mode="bilinear").permute(0, 2, 3, 1) + grid
            flow12 = F.grid_sample(flowCoarse.permute(0, 3, 1, 2), flowUp).permute(0, 2, 3, 1)
            F.grid_sample(caA.IsTensor, flow12)
        return bestPara

    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3, last_std=MATCH_STD))
    pA = restate.CoarseAlignOracle(sds["trunk"], 7, 1000, 0.05, 640, 1.2, variant="A")

    def port_qs(seed):
        I1, I2 = synth.make_pair(H, W, seed=seed)
        torch.manual_seed(seed)
        pA.setSource(I1)
        pA.setTarget(I2)
        r = pA.getCoarse(np.zeros((pA.It.size[1], pA.It.size[0])))
        with torch.no_grad():
            restate.fine_step_quickstart(dict(feat=sds["feat"], flow=sds["flow"]), pA.IsTensor, pA.ItTensor,
                                         restate.warp_grid(torch.from_numpy(r["H"])[None], H, W))
        return r["H"]

    hr, hp = ref_qs(seeds[0]), port_qs(seeds[0])
    out["qs_same_H"] = bool(np.abs(hr - hp).max() < 1e-6)
    tr, tp = time_pairs(ref_qs, seeds), time_pairs(port_qs, seeds)
    out["qs"] = dict(reference_s_per_pair=round(float(np.mean(tr)), 3), port_s_per_pair=round(float(np.mean(tp)), 3),
                     reference_over_port=round(float(np.mean(tr) / np.mean(tp)), 3), reference_pairs_per_s=round(1 / float(np.mean(tr)), 4),
                     port_pairs_per_s=round(1 / float(np.mean(tp)), 4))
    print("qs", out["qs"], flush=True)

    # ---------------- evaluation semantics: the reference's own multi-homography loop statement
    netm = _ref_nets(R, MATCH_STD)
    pfm = ref_loader.script_functions("evaluation/evalHpatch/evaluation.py", ["PredFlowMask"])["PredFlowMask"]
    loop = ref_loader.script_loop("evaluation/evalHpatch/evaluation.py", "nbCoarse <= args.maxCoarse")
    caB = _load_trunk(ref_loader.quiet(R["CoarseAlignB"], 7, 10000, 0.05, "Homography", 480, 2, False, 2.0, True, False))
    nh = {}

    def ref_ev(seed):
        I1, I2 = synth.make_pair(H, W, seed=seed, homography=True)
        torch.manual_seed(seed)
        with torch.no_grad():
            caB.setPair(I1, I2)
            Itw, Ith = caB.It.size
            featt = F.normalize(netm["netFeatCoarse"](caB.ItTensor))
            ns = dict(args=types.SimpleNamespace(maxCoarse=10, maskRegionTh=0.01), coarseModel=caB, network=netm, featt=featt,
                      grid=_grid(Ith, Itw), warper=R["kornia_geometry"].HomographyWarper(Ith, Itw),
                      It_bg=np.ones((Ith, Itw), dtype=np.float32), Mask=np.zeros((Ith, Itw), dtype=np.float32),
                      Coarse_Flow_Tensor=[], Fine_Flow_Tensor=[], Fine_Mask_Tensor=[], nbCoarse=0, PredFlowMask=pfm)
            loop(ns)
        nh[("ref", seed)] = ns["nbCoarse"]

    pB = restate.CoarseAlignOracle(sds["trunk"], 7, 10000, 0.05, 480, 2.0, variant="B")

    def port_ev(seed):
        I1, I2 = synth.make_pair(H, W, seed=seed, homography=True)
        torch.manual_seed(seed)
        pB.setPair(I1, I2)
        o = restate.multi_h_loop(pB, dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]), max_coarse=10, mask_region_th=0.01)
        nh[("port", seed)] = len(o["H"])

    tr, tp = time_pairs(ref_ev, seeds), time_pairs(port_ev, seeds)
    out["ev_same_nbH"] = all(nh[("ref", s)] == nh[("port", s)] for s in seeds)
    out["ev"] = dict(reference_s_per_pair=round(float(np.mean(tr)), 3), port_s_per_pair=round(float(np.mean(tp)), 3),
                     reference_over_port=round(float(np.mean(tr) / np.mean(tp)), 3), reference_pairs_per_s=round(1 / float(np.mean(tr)), 4),
                     port_pairs_per_s=round(1 / float(np.mean(tp)), 4),
                     homographies_per_pair=round(float(np.mean([nh[("ref", s)] for s in seeds])), 2))
    print("ev", out["ev"], flush=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()

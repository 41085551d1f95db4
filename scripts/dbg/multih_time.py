"""Experiment: throughput of BASELINE config 3 as literally described (variant B, multi-H loop, coarseIter 10 000,
minSize 480, 7 scales, scaleR 2, PredFlowMask per homography) with the per-pair device-resident driver pipeline.multi_h."""
import sys, time
sys.path.insert(0, "ransac-flow_amd")
import numpy as np, torch
from rfx import weights, synth
from rfx.pipeline import AlignPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
           match=weights.net_matchability_sd(3, last_std=0.02))
pipe = AlignPipeline(sds, nbScale=7, nbIter=10000, tolerance=0.05, minSize=480, scaleR=2.0, variant="B", device=dev)
pairs = [synth.make_pair(480, 640, seed=i) for i in range(B)]
MODE = sys.argv[2] if len(sys.argv) > 2 else "per-pair"
prep = pipe.prepare(pairs)
torch.manual_seed(0)
def run():
    feats = pipe.features(prep)
    nh = 0
    if MODE == "batched":
        nh = sum(len(o["H"]) for o in pipe.multi_h_batched(prep, maxCoarse=10, maskRegionTh=0.01, feats=feats))
    else:
        for b in range(B):
            out = pipe.multi_h(prep, b, maxCoarse=10, maskRegionTh=0.01, feats=feats)
            nh += len(out["H"])
    torch.cuda.synchronize()
    return nh
run()
t0 = time.perf_counter(); nh = run(); dt = time.perf_counter() - t0
print(MODE, "config-3-literal: %d pairs, %d accepted homographies, %.3f s -> %.1f pairs/s (nA=%d)" % (B, nh, dt, B / dt, pipe.features(prep)["nA"]))

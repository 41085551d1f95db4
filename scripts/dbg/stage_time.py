"""Experiment: wall time per pipeline stage at the default bench batch (sync between stages)."""
import sys, time
sys.path.insert(0, "ransac-flow_amd")
import torch
from rfx import weights, synth
from rfx.pipeline import AlignPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
           match=weights.net_matchability_sd(3))
pipe = AlignPipeline(sds, nbScale=7, nbIter=1000, tolerance=0.05, minSize=640, scaleR=1.2, variant="A", device=dev)
pairs = [synth.make_pair(480, 640, seed=i) for i in range(B)]
raw = pipe.upload_raw(pairs)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = sync(); prep = pipe.prepare_device(*raw)
    t1 = sync(); feats = pipe.features(prep)
    t2 = sync(); idx1, idx2, cnt = pipe._mutual_batched(feats, B)
    t3 = sync(); res = pipe.coarse(prep, feats=feats)
    t4 = sync()
    print("iter %d: prepare %.1f ms  features %.1f ms  mutualNN %.1f ms  coarse(given feats: mutualNN again + RANSAC) %.1f ms" % (
        it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
t0 = sync(); t1 = t0
eye = torch.eye(3, device=dev)
Hs = torch.stack([r["H"] if r["H"] is not None else eye for r in res])
t2 = sync(); f = pipe.fine_quickstart(prep, Hs); t3 = sync()
print("fine_quickstart %.1f ms" % ((t3 - t2) * 1e3))

"""Experiment: where a single-pair coarse step spends its time (graph replay of pyramid + trunk vs the eager mutual-NN / RANSAC part)."""
import sys, time
sys.path.insert(0, "ransac-flow_amd")
import torch
from rfx import weights, synth
from rfx.pipeline import AlignPipeline
dev = torch.device("cuda:0")
pipe = AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=7, nbIter=1000, tolerance=0.05, minSize=640, scaleR=1.2, variant="A",
                     device=dev, draw="device", seed=1)
raw = pipe.upload_raw([synth.make_pair(480, 640, seed=0)])
for _ in range(5):
    p, f = pipe.prepare_and_features(*raw); pipe.align_prepared(p, fine=False, feats=f)
torch.cuda.synchronize()
def timed(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("graph (pyramid + trunk + l2norm), incl. 2 input copies + 2 output clones: %.3f ms" % timed(lambda: pipe.prepare_and_features(*raw)))
p, f = pipe.prepare_and_features(*raw)
print("coarse part (mutual NN + draw + RANSAC + records, one sync): %.3f ms" % timed(lambda: pipe.align_prepared(p, fine=False, feats=f)))
print("whole step: %.3f ms" % timed(lambda: pipe.align_prepared(*((lambda pf: (pf[0],))(pipe.prepare_and_features(*raw))), fine=False)))
def step():
    p, f = pipe.prepare_and_features(*raw); return pipe.align_prepared(p, fine=False, feats=f)
print("whole step (feats passed): %.3f ms" % timed(step))
import torch.profiler as tp
with tp.profile(activities=[tp.ProfilerActivity.CPU, tp.ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))

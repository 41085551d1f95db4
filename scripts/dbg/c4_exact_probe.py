import sys, os, time, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ransac-flow_amd"))
import torch
from rfx import weights, synth, ops
from rfx.pipeline import AlignPipeline
dev = torch.device("cuda:0")
sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2), match=weights.net_matchability_sd(3, last_std=3.0))
for degen in ("device", "lapack"):
    pipe = AlignPipeline(sds, nbScale=5, nbIter=50000, tolerance=0.05, minSize=720, scaleR=2.0, variant="B", device=dev, draw="device", seed=1000, degenerate=degen, score_chunk="host")
    seeds = list(range(16))
    raw = pipe.upload_raw([synth.make_pair(720, 960, seed=s, homography=True) for s in seeds])
    def step():
        prep = pipe.prepare_device(*raw)
        R = ops.MultiHRecords(16, 90, 120, dev, max_h=11)
        pipe.multi_h_batched(prep, maxCoarse=10, maskRegionTh=0.01, records=R, want_lists=False, pair_ids=seeds)
        return R.rec
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out = {"mode": degen, "ms_per_step": round(ms, 1), "pairs_per_s": round(16 / ms * 1e3, 2)}
    if degen == "lapack":
        pipe.exact_log = []
        step(); torch.cuda.synchronize()
        L = pipe.exact_log
        out.update(flagged=int(sum(sum(r["n_degenerate"]) for r in L)), solved=int(sum(r["n_solved"] for r in L)), host_ms=round(sum(r["host_ms"] for r in L), 1),
                   rounds=[(r["round"], r["lo"], r["active"], int(sum(r["n_degenerate"])), int(r["n_solved"]), round(r["host_ms"], 1)) for r in L])
    print(json.dumps(out))

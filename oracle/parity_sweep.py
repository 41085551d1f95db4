"""TEST INFRASTRUCTURE ONLY -- end-to-end parity sweep of the HIP pipeline against the CPU oracle (oracle/restate.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU leg (a child process) may run this module; the product
path never imports it.  VERDICT r1 item 1: parity at the metric's own resolution must be a measured quantity over every
pair of the timed workload, with the oracle run END TO END ON ITS OWN HOMOGRAPHY, not a one-pair anecdote.

Protocol.  The GPU side (``dump_gpu_pair``) leaves, per pair, the match list, the RANSAC result and the final flow.  The
oracle side (``oracle_pair``) aligns the same synthetic pair from scratch on the CPU with the same index-draw rule --
``torch.randint(nMatch, (nbIter, 4), generator=Generator().manual_seed(DRAW_SEED + seed))``, what utils/outil.py:120
draws on a CPU run for a given generator state -- and ``compare`` reports

  * whether the two match lists are identical (utils/outil.py:32-45); if so: RANSAC inlier indices must be bit-identical
    (utils/outil.py:117-164), max |dH|, and the END-TO-END max |d flow12| (north-star bound 1e-3);
  * if not: every differing match with its float64 "tie evidence" computed from the ORACLE's features -- for a match only
    the oracle has, min(top1 - top2) over its row and its column of the score matrix (the device swapped the two best on
    one axis); for a match only the device has, max(best - score) over its row and column (the device saw it as the
    mutual maximum) -- i.e. the score margin that float32 round-off in ~40 convolution layers had to bridge; plus the
    end-to-end flow delta anyway (the two sides run DIFFERENT RANSACs from there on: nMatch enters the index draw) and the
    fine-stage-only delta with the device's H pushed through the oracle's fine stage.

Variants: "qs" = quick_start semantics (variant A, ResizeMaxSize, 7 scales x1.2, nA = 8 531 at 480x640, fine =
align2images.py:87-95); "ev" = evaluation semantics (variant B, ResizeMinSize, 7 scales x2, nA = 13 065, first
homography + PredFlowMask of evaluation/evalHpatch/evaluation.py:23-55).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "ransac-flow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

DRAW_SEED = 10_000
TIE_EPS = 2e-5          # feature round-off is <= 2e-5 per element: a flip whose evidence exceeds this is NOT explained by it -> failure

CONFIGS = {
    # name: (variant, nbScale, scaleR, minSize rule, nbIter, match head init)
    "qs": dict(variant="A", nbScale=7, scaleR=1.2, size="max", nbIter=1000),
    "ev": dict(variant="B", nbScale=7, scaleR=2.0, size="min", nbIter=10000),
}


def draw(seed, n, nb_iter):
    import torch
    g = torch.Generator().manual_seed(DRAW_SEED + int(seed))
    return torch.randint(int(n), (int(nb_iter), 4), generator=g)


def state_dicts():
    from rfx import weights
    return dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
                match=weights.net_matchability_sd(3))


# ------------------------------------------------------------------------------------------------ GPU side (called by tests / bench)


def gpu_pipeline(cfg_name, H, W, dev, sds=None):
    from rfx.pipeline import AlignPipeline
    c = CONFIGS[cfg_name]
    sds = sds or state_dicts()
    min_size = max(H, W) if c["size"] == "max" else min(H, W)
    return AlignPipeline(sds, nbScale=c["nbScale"], nbIter=c["nbIter"], tolerance=0.05, minSize=min_size, scaleR=c["scaleR"],
                         variant=c["variant"], device=dev)


def dump_gpu_pairs(cfg_name, seeds, H, W, dev, out_dir, pipe=None, batch=16):
    """Runs the HIP pipeline on synth.make_pair(H, W, seed) for every seed and writes out_dir/pair_<seed>.npz."""
    import torch
    from rfx import synth, ops
    pipe = pipe or gpu_pipeline(cfg_name, H, W, dev)
    os.makedirs(out_dir, exist_ok=True)
    seeds = list(seeds)
    for k in range(0, len(seeds), batch):
        sub = seeds[k:k + batch]
        pairs = [synth.make_pair(H, W, seed=s) for s in sub]
        prep = pipe.prepare_device(*pipe.upload_raw(pairs))
        feats = pipe.features(prep)
        res = pipe.coarse(prep, feats=feats, sample_fn=lambda b, n, it: draw(sub[b], n, it))
        eye = torch.eye(3, device=dev)
        Hs = torch.stack([r["H"] if r["H"] is not None else eye for r in res])
        if cfg_name == "qs":
            flow12 = pipe.fine_quickstart(prep, Hs)["flow12"]
            match = None
        else:
            h, w = prep["ItTensor"].shape[2], prep["ItTensor"].shape[3]
            pm = pipe.pred_flow_mask(prep["IsTensor"], ops.l2norm(pipe.feat(prep["ItTensor"])), ops.warp_grid(Hs, h, w))
            flow12, match = pm["flow12"], pm["match"]
        for b, s in enumerate(sub):
            r = res[b]
            ok = r["H"] is not None
            np.savez(os.path.join(out_dir, "pair_%d.npz" % s), seed=s, ok=ok, index1=r["index1"].cpu().numpy(),
                     index2=r["index2"].cpu().numpy(), H=(r["H"].cpu().numpy() if ok else np.zeros((3, 3), np.float32)),
                     inlier=(r["inlier"].cpu().numpy() if ok else np.zeros(0, bool)), flow12=flow12[b].cpu().numpy(),
                     match=(match[b, 0].cpu().numpy() if match is not None else np.zeros(0, np.float32)))
    return out_dir


# ------------------------------------------------------------------------------------------------ oracle side

_W = {}


def _worker_init(threads):
    import torch
    torch.set_num_threads(threads)
    _W["sds"] = state_dicts()


def oracle_pair(cfg_name, seed, H, W, sds=None):
    """The CPU oracle end to end on its own homography.  Returns (result dict, CoarseAlignOracle)."""
    import torch
    import restate
    from rfx import synth
    c = CONFIGS[cfg_name]
    sds = sds or _W.get("sds") or state_dicts()
    min_size = max(H, W) if c["size"] == "max" else min(H, W)
    ca = restate.CoarseAlignOracle(sds["trunk"], c["nbScale"], c["nbIter"], 0.05, min_size, c["scaleR"], variant=c["variant"],
                                   sample_fn=lambda n, it: draw(seed, n, it))
    I1, I2 = synth.make_pair(H, W, seed=seed)
    h = w = None
    if c["variant"] == "A":
        ca.setSource(I1)
        ca.setTarget(I2)
    else:
        ca.setPair(I1, I2)
    h, w = ca.It.size[1], ca.It.size[0]
    r = ca.getCoarse(np.zeros((h, w), dtype=np.float32))
    if r is None:           # sentinel path: recover the match list for the comparison
        if c["variant"] == "A":
            i1, i2 = restate.mutual_matching(ca.featsMultiScale, ca.featt.reshape(1024, -1))
        else:
            i1, i2 = ca.index1, ca.index2
        return dict(ok=False, index1=i1.numpy(), index2=i2.numpy()), ca
    nets = dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"])
    with torch.no_grad():
        fc = restate.warp_grid(torch.from_numpy(r["H"])[None], h, w)
        if cfg_name == "qs":
            r["flow12"] = restate.fine_step_quickstart(nets, ca.IsTensor, ca.ItTensor, fc)["flow12"][0].numpy()
        else:
            featt = torch.nn.functional.normalize(restate.feature_extractor(nets["feat"], ca.ItTensor))
            f12, match, _, _ = restate.pred_flow_mask(nets, ca.IsTensor, featt, fc, restate.identity_grid(h, w))
            r["flow12"], r["match"] = f12[0].numpy(), match
    r["ok"] = True
    return r, ca


def fine_given_h(cfg_name, ca, Hm, sds=None):
    """The oracle's fine stage with a homography handed over (isolates the fine stage from the match list)."""
    import torch
    import restate
    sds = sds or _W.get("sds") or state_dicts()
    nets = dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"])
    h, w = ca.It.size[1], ca.It.size[0]
    with torch.no_grad():
        fc = restate.warp_grid(torch.from_numpy(np.asarray(Hm, dtype=np.float32))[None], h, w)
        if cfg_name == "qs":
            return restate.fine_step_quickstart(nets, ca.IsTensor, ca.ItTensor, fc)["flow12"][0].numpy()
        featt = torch.nn.functional.normalize(restate.feature_extractor(nets["feat"], ca.ItTensor))
        return restate.pred_flow_mask(nets, ca.IsTensor, featt, fc, restate.identity_grid(h, w))[0][0].numpy()


def tie_evidence(ca, only_oracle, only_gpu):
    """float64 score margins (from the oracle's float32 features) behind every differing match."""
    A = ca.featsMultiScale.double()                  # (1024, nA)
    B = ca.featt.reshape(1024, -1).double()          # (1024, nB)
    out = []
    for (i, j) in only_oracle:
        row, col = A[:, i] @ B, A.t() @ B[:, j]
        tr, tc = row.topk(2).values, col.topk(2).values
        out.append(dict(match=[int(i), int(j)], side="oracle_only",
                        evidence=float(min(tr[0] - tr[1], tc[0] - tc[1])), score=float(row[j])))
    for (i, j) in only_gpu:
        row, col = A[:, i] @ B, A.t() @ B[:, j]
        s = float(row[j])
        out.append(dict(match=[int(i), int(j)], side="gpu_only", evidence=float(max(row.max() - s, col.max() - s)), score=s))
    return out


def compare(cfg_name, seed, H, W, gpu_npz):
    g = np.load(gpu_npz)
    r, ca = oracle_pair(cfg_name, seed, H, W)
    ref = set(zip(r["index1"].tolist(), r["index2"].tolist()))
    got = set(zip(g["index1"].tolist(), g["index2"].tolist()))
    same = bool(np.array_equal(r["index1"], g["index1"]) and np.array_equal(r["index2"], g["index2"]))
    rec = dict(seed=int(seed), n_matches_oracle=len(ref), n_matches_gpu=len(got), identical_list=same,
               n_differing=len(ref ^ got), oracle_ok=bool(r["ok"]), gpu_ok=bool(g["ok"]))
    if not same:
        rec["flips"] = tie_evidence(ca, sorted(ref - got), sorted(got - ref))
        rec["max_tie_evidence"] = max(f["evidence"] for f in rec["flips"])
    if r["ok"] and bool(g["ok"]):
        rec["max_abs_flow_delta_e2e"] = float(np.abs(r["flow12"] - g["flow12"]).max())
        rec["max_abs_H_delta"] = float(np.abs(r["H"] - g["H"]).max())
        if same:
            rec["inlier_indices_bit_exact"] = bool(np.array_equal(r["inlier"], g["inlier"]))
        else:
            # fine stage alone: the device's H through the oracle's fine stage
            rec["max_abs_flow_delta_given_gpu_H"] = float(np.abs(fine_given_h(cfg_name, ca, g["H"]) - g["flow12"]).max())
        if cfg_name == "ev":
            rec["max_abs_match_delta"] = float(np.abs(r["match"] - g["match"]).max())
    return rec


def _job(args):
    cfg_name, seed, H, W, path = args
    t0 = time.perf_counter()
    try:
        rec = compare(cfg_name, seed, H, W, path)
    except Exception as e:  # noqa: BLE001 -- a checker crash must surface in the summary, not kill the sweep
        rec = dict(seed=int(seed), error="%s: %s" % (type(e).__name__, e))
    rec["oracle_s"] = round(time.perf_counter() - t0, 2)
    return rec


def summarise(cfg_name, records, n_requested, elapsed, H, W):
    done = [r for r in records if "error" not in r]
    ident = [r for r in done if r["identical_list"]]
    diff = [r for r in done if not r["identical_list"]]
    flips = [f for r in diff for f in r["flips"]]
    both_ok = [r for r in done if "max_abs_flow_delta_e2e" in r]
    s = dict(config=cfg_name, size="%dx%d" % (H, W), pairs=len(done), pairs_requested=n_requested,
             errors=[r for r in records if "error" in r],
             identical_lists=len(ident),
             inlier_exact=sum(1 for r in ident if r.get("inlier_indices_bit_exact")),
             inlier_compared=sum(1 for r in ident if "inlier_indices_bit_exact" in r),
             inlier_indices_bit_exact=(all(r.get("inlier_indices_bit_exact", True) for r in ident) if ident else None),
             max_abs_H_delta_identical=max([r["max_abs_H_delta"] for r in ident if "max_abs_H_delta" in r], default=None),
             max_flow_delta_e2e_identical=max([r["max_abs_flow_delta_e2e"] for r in ident if "max_abs_flow_delta_e2e" in r], default=None),
             max_flow_delta_e2e=max([r["max_abs_flow_delta_e2e"] for r in ident if "max_abs_flow_delta_e2e" in r], default=None),   # = _identical (the bound 1e-3 applies where both sides ran the same RANSAC)
             pairs_with_flips=len(diff), total_flipped_matches=sum(r["n_differing"] for r in diff),
             total_matches=sum(r["n_matches_oracle"] for r in done),
             max_tie_evidence=max([f["evidence"] for f in flips], default=None),
             flips_all_near_ties=(all(f["evidence"] < TIE_EPS for f in flips) if flips else True), tie_eps=TIE_EPS,
             max_flow_delta_e2e_with_flips=max([r["max_abs_flow_delta_e2e"] for r in diff if "max_abs_flow_delta_e2e" in r], default=None),
             max_flow_delta_fine_stage_with_flips=max([r["max_abs_flow_delta_given_gpu_H"] for r in diff if "max_abs_flow_delta_given_gpu_H" in r], default=None),
             sentinel_agreement=all(r["oracle_ok"] == r["gpu_ok"] for r in ident),
             oracle_wall_s=round(elapsed, 1))
    if cfg_name == "ev" and both_ok:
        s["max_match_delta_identical"] = max([r["max_abs_match_delta"] for r in ident if "max_abs_match_delta" in r], default=None)
    return s


def sweep(cfg_name, dump_dir, seeds, H, W, workers=None, threads=8, budget_s=None):
    """Oracle + comparison for every dumped pair, ``workers`` processes x ``threads`` torch threads, optionally bounded
    by a wall-clock budget (pairs not reached are reported, not silently dropped).  Returns (summary, per-pair records)."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads, cores))
    workers = workers or max(1, cores // threads)
    jobs = [(cfg_name, s, H, W, os.path.join(dump_dir, "pair_%d.npz" % s)) for s in seeds]
    t0 = time.perf_counter()
    records = []
    if workers == 1:
        _worker_init(threads)
        for j in jobs:
            if budget_s and time.perf_counter() - t0 > budget_s:
                break
            records.append(_job(j))
    else:
        ctx = mp.get_context("spawn")
        with ctx.Pool(workers, initializer=_worker_init, initargs=(threads,)) as pool:
            it = pool.imap_unordered(_job, jobs)
            for _ in jobs:
                left = None if not budget_s else max(1.0, budget_s - (time.perf_counter() - t0))
                try:
                    records.append(it.next(timeout=left))
                except mp.TimeoutError:
                    break
            pool.terminate()
    records.sort(key=lambda r: r["seed"])
    return summarise(cfg_name, records, len(jobs), time.perf_counter() - t0, H, W), records


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="qs", choices=sorted(CONFIGS))
    ap.add_argument("--dump", required=True)
    ap.add_argument("--seeds", type=int, nargs="+", default=list(range(64)))
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--budget", type=float, default=None)
    ap.add_argument("--records", type=str, default=None, help="write the per-pair records to this JSON file")
    a = ap.parse_args()
    summary, records = sweep(a.config, a.dump, a.seeds, a.height, a.width, a.workers, a.threads, a.budget)
    if a.records:
        json.dump(dict(summary=summary, records=records), open(a.records, "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()

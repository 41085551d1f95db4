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

Round 3:
  * every pair whose match list differs is ALSO checked downstream of the arg-max: the oracle's RANSAC + fine stage on the
    DEVICE's match list with the same draw (inlier indices bit-exact, |dH|, flow) -> ``downstream_exact_given_matches``;
  * ``max_flow_delta_e2e`` is the maximum over ALL compared pairs (the identical-list subset keeps its ``_identical`` key);
  * ``--stability``: the oracle against ITSELF under a different thread count / oneDNN on-off -- how many of the pairs
    flip a match oracle-vs-oracle (is the reference CPU path stable at these margins?);
  * loop sweeps "ev_loop" (BASELINE config 3 as worded: the full multi-homography loop at 480x640, maxCoarse 10),
    "c4" (config 4: 960x720, 5 scales x2, nA = 21 675, coarseIter 50 000) and "c5" (config 5: the KITTI two-resolution loop,
    1242x376, nA = 25 747): the device driver leaves a per-round trace (``dump_gpu_loop``), the oracle replays EVERY ROUND
    from the device's own state (teacher-forced: same surviving-match count, RANSAC on the device's matches with the same
    draw, its own PredFlowMask on its own H, accept decision, next mask with a threshold-pixel proof) and also runs its
    own loop end to end (``free_run``).
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


def oracle_backend():
    """The checker the sweeps run: the REFERENCE ITSELF (oracle/ref_oracle.py: /root/reference, or its byte-compiled form
    oracle/_ref on the GPU box) whenever it is present, else the port (oracle/restate.py).  ``RFX_ORACLE=port`` forces the port."""
    import ref_loader
    if os.environ.get("RFX_ORACLE", "auto") != "port" and ref_loader.available():
        import ref_oracle
        ref_oracle.kind()
        return ref_oracle
    import restate
    return restate


def oracle_name(O=None):
    O = O or oracle_backend()
    return O.KIND if getattr(O, "KIND", None) else "port (oracle/restate.py)"

TIE_EPS = 2e-5          # feature round-off is <= 2e-5 per element: a flip whose evidence exceeds this is NOT explained by it -> failure

MULTIH_MATCH_STD = 3.0   # saturating matchability head of the multi-H workloads (bench.py, tests/golden/make_golden.py)
SAT_EPS = 1e-5           # a mask pixel that differs must have its matchability within this of the saturation threshold

CONFIGS = {
    # name: (variant, nbScale, scaleR, minSize rule, nbIter, match head init)
    "qs": dict(variant="A", nbScale=7, scaleR=1.2, size="max", nbIter=1000),
    "ev": dict(variant="B", nbScale=7, scaleR=2.0, size="min", nbIter=10000),
    # loop sweeps: BASELINE configs 3 / 4 / 5 at their own sizes (bench.py --config 3 / 4 / 5 times exactly these)
    "ev_loop": dict(variant="B", nbScale=7, scaleR=2.0, size="min", nbIter=10000, loop="hpatch", H=480, W=640, maxCoarse=10,
                    th=0.01, amp=0.05),
    "c4": dict(variant="B", nbScale=5, scaleR=2.0, size="min", nbIter=50000, loop="hpatch", H=720, W=960, maxCoarse=10, th=0.01,
               amp=0.05),
    "c5": dict(variant="B", nbScale=3, scaleR=1.2, size=800, nbIter=50000, loop="kitti", H=376, W=1242, fineSize=650, th=0.005,
               cc_th=0.01, amp=0.02),
}


def draw(seed, n, nb_iter):
    import torch
    g = torch.Generator().manual_seed(DRAW_SEED + int(seed))
    return torch.randint(int(n), (int(nb_iter), 4), generator=g)


def draw_round(seed, k, n, nb_iter):
    """Index draw of round k (the k-th RANSAC call) of pair ``seed`` in the loop sweeps."""
    import torch
    g = torch.Generator().manual_seed(DRAW_SEED + 1000 * int(seed) + int(k))
    return torch.randint(int(n), (int(nb_iter), 4), generator=g)


def state_dicts(match_std=None):
    from rfx import weights
    return dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
                match=weights.net_matchability_sd(3) if match_std is None else weights.net_matchability_sd(3, last_std=match_std))


def make_pair(cfg_name, seed, H, W):
    from rfx import synth
    c = CONFIGS[cfg_name]
    if "loop" in c:
        return synth.make_pair(H, W, seed=seed, homography=True, amp=c["amp"])
    return synth.make_pair(H, W, seed=seed)


def _min_size(c, H, W):
    return c["size"] if isinstance(c["size"], int) else (max(H, W) if c["size"] == "max" else min(H, W))


# ------------------------------------------------------------------------------------------------ GPU side (called by tests / bench)


DEVICE_DRAW_SEED = 1000     # seed of the device-side Philox draw in the "device" sweeps = bench.py's timed pipelines


def gpu_pipeline(cfg_name, H, W, dev, sds=None, draw="host", score_chunk="host", degenerate="auto"):
    """The device pipeline of a sweep.  ``draw="host"`` (+ degenerate "auto" = LAPACK patching): the exact mode, every draw handed
    over by the sweep; ``draw="device"``: the mode bench.py TIMES -- Philox draws on the device keyed by (DEVICE_DRAW_SEED, absolute
    pair id, round), rank-deficient samples solved by the device itself.  ``score_chunk="host"``: scores summed in the host
    sgemm's K blocks (the parity modes; resolved once, recorded in the dump's meta.json and in the summary)."""
    from rfx.pipeline import AlignPipeline
    c = CONFIGS[cfg_name]
    sds = sds or state_dicts(MULTIH_MATCH_STD if "loop" in c else None)
    return AlignPipeline(sds, nbScale=c["nbScale"], nbIter=c["nbIter"], tolerance=0.05, minSize=_min_size(c, H, W), scaleR=c["scaleR"],
                         variant=c["variant"], device=dev, draw=draw, seed=DEVICE_DRAW_SEED, score_chunk=score_chunk, degenerate=degenerate)


def write_meta(out_dir, pipe, **extra):
    """What the dumped results were computed WITH -- read back by sweep() into the summary (the numbers describe themselves)."""
    meta = dict(score_chunk_products=int(pipe.score_chunk), score_chunk_source=pipe.score_chunk_source, draw=pipe.draw,
                degenerate=pipe._degenerate_mode(pipe.draw == "host"), draw_seed=int(pipe.seed))
    meta.update(extra)
    json.dump(meta, open(os.path.join(out_dir, "meta.json"), "w"))
    return meta


def dump_gpu_pairs(cfg_name, seeds, H, W, dev, out_dir, pipe=None, batch=16):
    """Runs the HIP pipeline on synth.make_pair(H, W, seed) for every seed and writes out_dir/pair_<seed>.npz."""
    import torch
    from rfx import synth, ops
    pipe = pipe or gpu_pipeline(cfg_name, H, W, dev)
    os.makedirs(out_dir, exist_ok=True)
    write_meta(out_dir, pipe, draw="host (explicit per-pair draws: parity_sweep.draw)", degenerate=pipe._degenerate_mode(True))
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


def dump_gpu_loop(cfg_name, seeds, dev, out_dir, pipe=None, batch=4, sub=4, draw="host"):
    """Runs the lock-step multi-homography driver of BASELINE config 3 / 4 / 5 (``cfg_name`` = ev_loop / c4 / c5) on
    make_pair(seed) for every seed and writes out_dir/pair_<seed>.npz: the cached match list and, per round, the mask before the
    round, the surviving-match count, RANSAC status and H, the /8 outputs, the composed flow at every ``sub``-th pixel, the accept
    flag / gain and the mask after the round.
    ``draw="host"``: the explicit per-round draw ``draw_round`` (exact mode: LAPACK-patched rank-deficient samples).
    ``draw="device"``: THE MODE bench.py TIMES -- the device's own Philox draws keyed by (DEVICE_DRAW_SEED, pair id = seed, round),
    the device's own null vector on rank-deficient samples; the samples of every round are dumped (int32) so that the oracle
    replays each round with exactly the hypotheses the device scored (utils/outil.py:120 takes them through the randint proxy)."""
    import torch
    from rfx import ops
    c = CONFIGS[cfg_name]
    H, W = c["H"], c["W"]
    pipe = pipe or gpu_pipeline(cfg_name, H, W, dev, draw=draw)
    device_draw = draw == "device"
    if device_draw and pipe.draw != "device":
        raise ValueError("draw='device' needs a pipeline built with draw='device'")
    os.makedirs(out_dir, exist_ok=True)
    write_meta(out_dir, pipe, **(dict(draw_epoch=pipe.DRAW_TAG["multi_h" if c["loop"] == "hpatch" else "kitti"]) if device_draw else
                                 dict(draw="host (explicit per-round draws: parity_sweep.draw_round)", degenerate=pipe._degenerate_mode(True))))
    seeds = list(seeds)
    for k0 in range(0, len(seeds), batch):
        sb = seeds[k0:k0 + batch]
        raw = pipe.upload_raw([make_pair(cfg_name, s, H, W) for s in sb])
        calls = [0] * len(sb)

        def fn(b, n, it):
            calls[b] += 1
            return draw_round(sb[b], calls[b] - 1, n, it)
        kw = dict(pair_ids=sb) if device_draw else dict(sample_fn=fn)
        trace = []
        if c["loop"] == "hpatch":
            outs = pipe.multi_h_batched(pipe.prepare_device(*raw), maxCoarse=c["maxCoarse"], maskRegionTh=c["th"], trace=trace, **kw)
        else:
            outs = pipe.multi_h_kitti_batched(raw[0], raw[1], fineSize=c["fineSize"], maskRegionTh=c["th"], cc_th=c["cc_th"],
                                              trace=trace, **kw)
        for b, s in enumerate(sb):
            i1, i2, cnt = outs[b]["matches"]
            n = int(cnt.item())
            rows = [(t, t["active"].index(b)) for t in trace if b in t["active"]]
            pack = lambda m: np.packbits(m.cpu().numpy() > 0.5, axis=-1)
            d = dict(seed=s, index1=i1[:n].cpu().numpy(), index2=i2[:n].cpu().numpy(), nbH=outs[b]["nbH"], sub=sub,
                     mask_before=np.stack([pack(t["mask_before"][k]) for t, k in rows]),
                     mask_after=np.stack([pack(t["mask_after"][k]) for t, k in rows]),
                     n=np.asarray([int(t["n"][k]) for t, k in rows]), status=np.asarray([int(t["res"][k, 0]) for t, k in rows]),
                     winner=np.asarray([int(t["res"][k, 2]) for t, k in rows]),
                     inlier=np.stack([np.packbits(t["inlier"][k].cpu().numpy()) for t, k in rows]),
                     H=np.stack([t["H"][k].cpu().numpy() for t, k in rows]),
                     flowDown8=np.stack([t["pm"]["flowDown8"][k].cpu().numpy() for t, k in rows]),
                     matchDown8=np.stack([torch.cat((t["pm"]["match12Down8"][k], t["pm"]["match21Down8"][k])).cpu().numpy() for t, k in rows]),
                     flow12_sub=np.stack([t["pm"]["flow12"][k, ::sub, ::sub].cpu().numpy() for t, k in rows]),
                     accept=np.asarray([int(t["accept"][k]) for t, k in rows]), gain=np.asarray([float(t["gain"][k]) for t, k in rows]),
                     final_mask=pack(outs[b]["mask"]))
            if device_draw:
                d["samples"] = np.stack([t["samples"][k].cpu().numpy().astype(np.int32) for t, k in rows])
                d["round_id"] = np.asarray([int(t["round"]) for t, k in rows])
            if c["loop"] == "kitti":
                d["flowD2"] = np.stack([t["flowD2"][k].cpu().numpy() for t, k in rows])
            np.savez_compressed(os.path.join(out_dir, "pair_%d.npz" % s), **d)
        del trace, outs
        torch.cuda.empty_cache()
    return out_dir


# ------------------------------------------------------------------------------------------------ oracle side

_W = {}


def _worker_init(threads, match_std=None):
    import torch
    torch.set_num_threads(threads)
    _W["sds"] = state_dicts(match_std)


def oracle_pair(cfg_name, seed, H, W, sds=None):
    """The CPU oracle end to end on its own homography.  Returns (result dict, CoarseAlignOracle)."""
    import torch
    from rfx import synth
    restate = oracle_backend()
    c = CONFIGS[cfg_name]
    sds = sds or _W.get("sds") or state_dicts()
    ca = restate.CoarseAlignOracle(sds["trunk"], c["nbScale"], c["nbIter"], 0.05, _min_size(c, H, W), c["scaleR"], variant=c["variant"],
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
    r = dict(r)
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
    restate = oracle_backend()
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
    rec = dict(seed=int(seed), oracle=oracle_name(), n_matches_oracle=len(ref), n_matches_gpu=len(got), identical_list=same,
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
            # match = match12 * (-1 <= flow12 <= 1) (evaluation/evalHpatch/evaluation.py:51): a pixel whose flow sits on the
            # +-1 boundary may pass the test on one side only -- counted, and excluded from the value comparison
            ino = (np.abs(r["flow12"]) <= 1).all(axis=-1)
            ing = (np.abs(g["flow12"]) <= 1).all(axis=-1)
            agree = ino == ing
            rec["inbounds_mismatch_pixels"] = int((~agree).sum())
            rec["max_abs_match_delta"] = float(np.abs(r["match"] - g["match"])[agree].max())
    if not same and bool(g["ok"]):
        rec.update(downstream_given_matches(cfg_name, seed, ca, g))
    return rec


def matches_from_indices(ca, i1, i2):
    """match1 / match2 (n,3) of the oracle for a given index list (quick_start/coarseAlignFeatMatch.py:150-155)."""
    import torch
    i1, i2 = torch.as_tensor(np.asarray(i1), dtype=torch.long), torch.as_tensor(np.asarray(i2), dtype=torch.long)
    ones = torch.ones(len(i1))
    return (torch.stack((ca.HMultiScale[i1], ca.WMultiScale[i1], ones), dim=1),
            torch.stack((ca.Ht[i2], ca.Wt[i2], ones), dim=1))


def downstream_given_matches(cfg_name, seed, ca, g):
    """Everything downstream of the arg-max on the DEVICE's match list: the oracle's RANSAC with the same draw (the draw the
    device made: ``draw(seed, nMatch_device, nbIter)``) and the oracle's fine stage on the oracle's own H from it.  Shows that
    a pair whose match list differs by a near-tie is exact from there on."""
    restate = oracle_backend()
    c = CONFIGS[cfg_name]
    m1, m2 = matches_from_indices(ca, g["index1"], g["index2"])
    Hb, cnt, inl, _ = restate.ransac(m1, m2, 0.05, draw(seed, len(m1), c["nbIter"]))
    if Hb is None:
        return dict(downstream_ok=False, downstream_note="oracle RANSAC aborted on the device's match list")
    out = dict(downstream_ok=True, downstream_inlier_bit_exact=bool(np.array_equal(inl, g["inlier"])),
               downstream_H_delta=float(np.abs(Hb - g["H"]).max()),
               downstream_flow_delta=float(np.abs(fine_given_h(cfg_name, ca, Hb) - g["flow12"]).max()))
    return out


# ------------------------------------------------------------------------------------------------ loop sweeps (configs 3 / 4 / 5)


def _unpack(bits, w):
    return np.unpackbits(bits, axis=-1)[..., :w].astype(np.float32)


def compare_loop(cfg_name, seed, gpu_npz):
    """Oracle vs device for ONE pair of a loop sweep.  (1) match list with tie evidence; (2) every device round replayed on
    the oracle from the device's own state (teacher-forced); (3) the oracle's own loop end to end (free run)."""
    import torch
    import torch.nn.functional as F
    restate = oracle_backend()
    is_ref = hasattr(restate, "KIND")
    c = CONFIGS[cfg_name]
    H, W = c["H"], c["W"]
    g = np.load(gpu_npz)
    sds = _W.get("sds") or state_dicts(MULTIH_MATCH_STD)
    nets = dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"])
    state = {"k": 0}
    device_draw = "samples" in g.files
    if device_draw:
        # the TIMED mode: the device's own Philox draws.  Round k of pair ``seed`` = Philox(key = the pipeline's draw seed, counter =
        # (hypothesis, pair id = seed, stream = (epoch 0 << 32) | k)) % n (rfx_draw_samples_i64; tests/philox_ref.py pins the
        # generator on the published known-answer vectors).  The samples the device actually scored were dumped per round: they
        # are what the reference's RANSAC receives; the numpy restatement regenerates them (asserted equal) and supplies the
        # draws of the oracle's own free-running loop, whose n may differ
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import philox_ref
        meta = json.load(open(os.path.join(os.path.dirname(gpu_npz), "meta.json")))

        def draw_round(seed_, k, n, it):                              # noqa: F811 -- shadows the host-draw rule for this pair
            smp = philox_ref.draw_samples([int(n)], int(it), meta["draw_seed"], (int(meta.get("draw_epoch", 0)) << 32) | int(k),
                                          pair_ids=[int(seed_)])[0]
            if k < len(g["n"]) and int(n) == int(g["n"][k]) and int(g["round_id"][k]) == k:
                assert np.array_equal(smp, g["samples"][k].astype(np.int64)), "dumped device samples != Philox restatement"
            return torch.from_numpy(smp)
    else:
        draw_round = globals()["draw_round"]

    def sample_fn(n, it):
        state["k"] += 1
        return draw_round(seed, state["k"] - 1, n, it)
    ca = restate.CoarseAlignOracle(sds["trunk"], c["nbScale"], c["nbIter"], 0.05, _min_size(c, H, W), c["scaleR"], variant="B",
                                   sample_fn=sample_fn)
    Is, It = make_pair(cfg_name, seed, H, W)
    ca.setPair(Is, It)
    kitti = c["loop"] == "kitti"
    if kitti:
        T = restate.kitti_setup(Is, It, c["fineSize"])
        h, w = T["org"]
    else:
        h, w = ca.It.size[1], ca.It.size[0]
        with torch.no_grad():
            featt = F.normalize(restate.feature_extractor(nets["feat"], ca.ItTensor))
        grid = restate.identity_grid(h, w)
    # ---- (1) the cached match list
    ref = set(zip(ca.index1.tolist(), ca.index2.tolist()))
    got = set(zip(g["index1"].tolist(), g["index2"].tolist()))
    same = bool(np.array_equal(ca.index1.numpy(), g["index1"]) and np.array_equal(ca.index2.numpy(), g["index2"]))
    rec = dict(seed=int(seed), oracle=oracle_name(restate), nA=int(ca.featsMultiScale.shape[1]), nB=int(ca.featt.shape[2] * ca.featt.shape[3]),
               n_matches_oracle=len(ref), n_matches_gpu=len(got), identical_list=same, n_differing=len(ref ^ got),
               rounds=int(len(g["n"])), nbH_gpu=int(g["nbH"]), draw="device" if device_draw else "host")
    if not same:
        rec["flips"] = tie_evidence(ca, sorted(ref - got), sorted(got - ref))
        rec["max_tie_evidence"] = max(f["evidence"] for f in rec["flips"])
    # ---- (2) every device round from the device's own state
    WI, HI = restate.get_wh_int(ca.featt.shape[2], ca.featt.shape[3])
    i1d, i2d = torch.from_numpy(g["index1"]), torch.from_numpy(g["index2"])
    m1_all, m2_all = matches_from_indices(ca, i1d, i2d)
    sub = int(g["sub"])
    rr = []
    nb = 0
    for k in range(len(g["n"])):
        fg = _unpack(g["mask_before"][k], w)                       # It_bg = 1: the mask IS fgMask
        keep = ca._mask_to_feat(fg)[0, 0]
        valid = keep[WI[i2d], HI[i2d]]
        n_or = int(valid.sum())
        r = dict(k=k, n_gpu=int(g["n"][k]), n_oracle=n_or, count_equal=n_or == int(g["n"][k]), status_gpu=int(g["status"][k]))
        if n_or != int(g["n"][k]) or n_or < 4:
            rr.append(r)
            if n_or < 4 and int(g["n"][k]) < 4:
                r["both_below_4"] = True
            continue
        if is_ref:
            # the REFERENCE's own getCoarse(fgMask) on the device's cached match list: its mask interpolation, its selection of
            # the surviving matches, its outil.RANSAC -- with this round's draw handed to utils/outil.py:120
            ca.set_matches(i1d, i2d)
            res = ca.getCoarse(fg, sample_fn=lambda n, it, k=k: draw_round(seed, k, n, it))
            ca.set_matches(None, None)
            assert ca.last["n"] == n_or, "the reference kept %d matches, the restated mask lines %d" % (ca.last["n"], n_or)
            Hb, inl = (None, None) if res is None else (res["H"], res["inlier"])
        else:
            Hb, cnt, inl, _ = restate.ransac(m1_all[valid], m2_all[valid], 0.05, draw_round(seed, k, n_or, c["nbIter"]))
        r["status_equal"] = (Hb is None) == (int(g["status"][k]) != 0)
        if Hb is None or int(g["status"][k]) != 0:
            rr.append(r)
            continue
        r["H_delta"] = float(np.abs(Hb - g["H"][k]).max())
        if "inlier" in g.files:
            r["inlier_bit_exact"] = bool(np.array_equal(inl, np.unpackbits(g["inlier"][k])[:n_or].astype(bool)))
        if r["H_delta"] > 2e-6 and "winner" in g.files:
            # a late round with a handful of matches can only choose near-degenerate 4-point samples: the DLT null vector is
            # then ill-conditioned (error ~ eps64 * sigma_1 / sigma_8 on BOTH sides) -- report the conditioning of the winner
            smp = draw_round(seed, k, n_or, c["nbIter"])[int(g["winner"][k])]
            sv = np.linalg.svd(restate.dlt_matrix(m1_all[valid][smp][None].numpy(), m2_all[valid][smp][None].numpy()), compute_uv=False)[0]
            r["dlt_sigma8_over_sigma1"] = float(sv[7] / sv[0])
            # which hypothesis won on each side?  (the oracle's winner = first maximum of its per-hypothesis counts)
            allsmp = draw_round(seed, k, n_or, c["nbIter"])
            uniq = restate.filter_samples(allsmp)
            Hs_o, cnt_o = restate.score_ransac(m1_all[valid], m2_all[valid], 0.05, uniq)
            wo = uniq[int(torch.argmax(cnt_o))]
            wg = allsmp[int(g["winner"][k])]
            r["winner_same_sample"] = bool(torch.equal(wo, wg))
            r["winner_count_oracle"] = int(cnt_o.max())
            gi = int((uniq == wg).all(dim=1).nonzero()[0, 0]) if (uniq == wg).all(dim=1).any() else -1
            r["oracle_count_of_gpu_winner"] = int(cnt_o[gi]) if gi >= 0 else None
            # the round is decided by a rank-deficient 4-point sample when EITHER side's winner is one (device-draw mode: the
            # device keeps its own null vector for such a sample, the reference's LAPACK another -- either can come out on top)
            svo = np.linalg.svd(restate.dlt_matrix(m1_all[valid][wo][None].numpy(), m2_all[valid][wo][None].numpy()), compute_uv=False)[0]
            r["dlt_sigma8_over_sigma1_oracle_winner"] = float(svo[7] / svo[0])
            r["degenerate_winner"] = bool(sv[7] / sv[0] < 1e-10 or svo[7] / svo[0] < 1e-10)
        Hm = torch.from_numpy(np.asarray(Hb, dtype=np.float32))[None]
        if kitti:
            match, flow_d2, fd8, md8, flow12 = restate.kitti_fine_round(nets, T, Hm, c["cc_th"])
            fd8, md8 = fd8.numpy(), md8.numpy()
            r["flowD2_delta"] = float(np.abs(flow_d2.numpy()[0] - g["flowD2"][k]).max())
            stat = (match > 0.9999) * (1 - fg)
        else:
            with torch.no_grad():
                flow12, match, fd8, md8 = restate.pred_flow_mask(nets, ca.IsTensor, featt, restate.warp_grid(Hm, h, w), grid)
            stat = match * (1 - fg)
        fo, fd = flow12[0, ::sub, ::sub].numpy(), g["flow12_sub"][k]
        r["flow12_delta_raw"] = float(np.abs(fo - fd).max())
        # flow12 is unbounded where the homography sends a pixel towards its horizon line (x'/z' with z' -> 0: a late round
        # with a handful of matches can return such an H); there a 1e-6 difference in H or one float32 rounding of z' moves the
        # quotient by whole units on either side.  Downstream only the in-bounds part is used (match = match12 *
        # (-1 <= flow12 <= 1), evaluation/evalHpatch/evaluation.py:51): the bound applies where BOTH sides are in bounds, and
        # the pixels whose in-bounds decision differs are counted separately (a threshold effect, like the mask pixels)
        ino, ind = np.abs(fo).max(axis=-1) <= 1, np.abs(fd).max(axis=-1) <= 1
        both = ino & ind
        r["flow12_delta"] = float(np.abs(fo - fd)[both].max()) if both.any() else 0.0
        r["inbounds_frac"] = float(both.mean())
        r["inbounds_mismatch_frac"] = float((ino != ind).mean())
        r["flowDown8_delta"] = float(np.abs(fd8[0] - g["flowDown8"][k]).max())
        r["matchDown8_frac_over_1e-3"] = float(np.mean(np.abs(md8[0] - g["matchDown8"][k]) > 1e-3))
        gain = float(stat.mean())
        acc = bool(gain > c["th"] or nb == 0)
        r.update(gain_oracle=gain, gain_gpu=float(g["gain"][k]), accept_oracle=acc, accept_gpu=bool(g["accept"][k]),
                 accept_equal=acc == bool(g["accept"][k]))
        if bool(g["accept"][k]):
            nb += 1
        if acc and bool(g["accept"][k]):
            upd = fg + match * (1 - fg)
            new = ((upd > 0.9999) if kitti else (upd >= 1.0)).astype(np.float32)
            diff = new != _unpack(g["mask_after"][k], w)
            r["mask_diff_frac"] = float(diff.mean())
            if diff.any() and not kitti:
                # a differing pixel must sit at a threshold on the oracle's side too: the saturation of the sigmoid
                # (match >= 1 - eps: the mask rule is ``>= 1``), or the in-bounds test of evalHpatch/evaluation.py:51
                # (|flow12| within 1e-4 of 1, where ``-1 <= flow12 <= 1`` flips and zeroes the matchability)
                f12 = np.abs(flow12[0].numpy())
                at_sat = match > 1 - SAT_EPS
                at_inb = (np.abs(f12 - 1) < 1e-4).any(axis=-1)
                r["mask_diff_min_match"] = float(match[diff].min())
                r["mask_diff_at_threshold"] = bool((at_sat | at_inb)[diff].all())
        rr.append(r)
    rec["round_records"] = rr
    # ---- (3) the oracle's own loop, end to end.  When the lists are identical and every round above reproduced the device's
    # count, status, accept decision and NEXT MASK exactly, the oracle's own loop visits the same states by induction (same
    # state + same draw -> the computations of (2)): it is not run a second time.
    if same and rr and all(q["count_equal"] and q.get("status_equal", True) and q.get("accept_equal", True)
                           and q.get("mask_diff_frac", 0.0) == 0.0 for q in rr):
        rec["free_run"] = dict(implied_by_exact_rounds=True, nbH_oracle=int(g["nbH"]), nbH_gpu=int(g["nbH"]), same_nbH=True,
                               max_H_delta=max([q["H_delta"] for q in rr if "H_delta" in q and q.get("accept_gpu")], default=0.0),
                               final_mask_diff_frac=0.0)
        return rec
    state["k"] = 0
    if kitti:
        o = restate.multi_h_loop_kitti(ca, nets, Is, It, c["fineSize"], mask_region_th=c["th"], cc_th=c["cc_th"])
    else:
        o = restate.multi_h_loop(ca, nets, max_coarse=c["maxCoarse"], mask_region_th=c["th"])
    Hg = g["H"][g["accept"] > 0]
    rec["free_run"] = dict(nbH_oracle=len(o["H"]), nbH_gpu=int(g["nbH"]), same_nbH=len(o["H"]) == int(g["nbH"]))
    if len(o["H"]) == int(g["nbH"]) and len(o["H"]):
        rec["free_run"]["max_H_delta"] = float(np.abs(np.stack(o["H"]) - Hg).max())
        rec["free_run"]["final_mask_diff_frac"] = float((o["masks"][-1] != _unpack(g["final_mask"], w)).mean())
    return rec


def summarise_loop(cfg_name, records, n_requested, elapsed):
    c = CONFIGS[cfg_name]
    done = [r for r in records if "error" not in r]
    rounds = [q for r in done for q in r["round_records"]]
    allfull = [q for q in rounds if "H_delta" in q]
    # a winning 4-point sample whose 8x9 DLT system is rank deficient (sigma_8 / sigma_1 < 1e-10: collinear lattice points) has a
    # TWO-dimensional null space; which unit vector of it LAPACK's SVD returns is decided by rounding noise inside dgesdd, not
    # by the geometry -- it differs between LAPACK builds on the CPU, too.  Such rounds are listed apart, not hidden.
    degen = [q for q in allfull if q.get("degenerate_winner")]
    full = [q for q in allfull if not q.get("degenerate_winner")]
    flips = [f for r in done for f in r.get("flips", [])]
    fr = [r["free_run"] for r in done]
    mx = lambda key, rows=full: max([q[key] for q in rows if key in q], default=None)
    thr = [q for q in full if "mask_diff_at_threshold" in q]
    s = dict(config=cfg_name, oracle=(done[0].get("oracle") if done else oracle_name()), size="%dx%d" % (c["H"], c["W"]), pairs=len(done),
             pairs_requested=n_requested,
             errors=[r for r in records if "error" in r],
             nA=done[0]["nA"] if done else None, nB=done[0]["nB"] if done else None,
             identical_lists=sum(1 for r in done if r["identical_list"]), total_matches=sum(r["n_matches_oracle"] for r in done),
             total_flipped_matches=sum(r["n_differing"] for r in done if not r["identical_list"]),
             max_tie_evidence=max([f["evidence"] for f in flips], default=None),
             flips_all_near_ties=(all(f["evidence"] < TIE_EPS for f in flips) if flips else True), tie_eps=TIE_EPS,
             rounds=len(rounds), rounds_count_equal=sum(1 for q in rounds if q["count_equal"]),
             rounds_compared=len(allfull), rounds_status_equal=all(q.get("status_equal", True) for q in rounds),
             rounds_degenerate_winner=len(degen),
             degenerate_winner_rounds=[dict(k=q["k"], n=q["n_gpu"], H_delta=q["H_delta"], sigma8_over_sigma1=q.get("dlt_sigma8_over_sigma1"),
                                            accept_equal=q.get("accept_equal")) for q in degen],
             homographies_per_pair_gpu=round(float(np.mean([r["nbH_gpu"] for r in done])), 2) if done else None,
             max_H_delta=mx("H_delta"), rounds_H_within_2e6=sum(1 for q in full if q["H_delta"] <= 2e-6),
             max_dlt_conditioning_of_rounds_above_2e6=max([q["dlt_sigma8_over_sigma1"] for q in full if "dlt_sigma8_over_sigma1" in q], default=None),
             inlier_bit_exact="%d/%d" % (sum(1 for q in full if q.get("inlier_bit_exact")), sum(1 for q in full if "inlier_bit_exact" in q)),
             max_flow12_delta=mx("flow12_delta"), max_flow12_delta_incl_out_of_bounds=mx("flow12_delta_raw"),
             max_inbounds_mismatch_frac=mx("inbounds_mismatch_frac"),
             max_flowDown8_delta=mx("flowDown8_delta"),
             max_flowD2_delta=mx("flowD2_delta"), max_matchDown8_frac_over_1e3=mx("matchDown8_frac_over_1e-3"),
             accept_equal="%d/%d" % (sum(1 for q in allfull if q["accept_equal"]), len(allfull)),
             max_gain_delta=max([abs(q["gain_oracle"] - q["gain_gpu"]) for q in full], default=None),
             max_mask_diff_frac=mx("mask_diff_frac"),
             mask_diffs_all_at_threshold=(all(q["mask_diff_at_threshold"] for q in thr) if thr else None),
             free_run_same_nbH="%d/%d" % (sum(1 for f in fr if f["same_nbH"]), len(fr)),
             free_run_max_H_delta=max([f["max_H_delta"] for f in fr if "max_H_delta" in f], default=None),
             free_run_max_final_mask_diff=max([f["final_mask_diff_frac"] for f in fr if "final_mask_diff_frac" in f], default=None),
             oracle_wall_s=round(elapsed, 1))
    # a round is exact given the device's state when: same surviving-match count (precondition of being compared), bit-exact
    # inlier indices, H to float32 round-off, in-bounds flow within the north-star bound and the same accept decision
    ok = lambda q: q.get("inlier_bit_exact", True) and q["H_delta"] <= 2e-6 and q["flow12_delta"] < 1e-3 and q["accept_equal"]
    s["rounds_exact_given_state"] = "%d/%d" % (sum(1 for q in full if ok(q)), len(full))
    if degen:
        s["rounds_exact_given_state"] += " (+ %d round(s) won by a rank-deficient sample, listed in degenerate_winner_rounds)" % len(degen)
    return s


# ------------------------------------------------------------------------------------------------ oracle vs oracle


def oracle_stability(cfg_name, seed, H, W, threads_a, threads_b):
    """The oracle's match list and end-to-end flow under two CPU execution settings: (threads_a, oneDNN on) vs
    (threads_b, oneDNN off).  Is the reference CPU path itself stable at the arg-max margins where the device flips?"""
    import torch
    runs = []
    for threads, mkldnn in ((threads_a, True), (threads_b, False)):
        torch.set_num_threads(threads)
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            r, ca = oracle_pair(cfg_name, seed, H, W)
        runs.append((r, ca))
    (ra, ca), (rb, _) = runs
    A = set(zip(ra["index1"].tolist(), ra["index2"].tolist()))
    B = set(zip(rb["index1"].tolist(), rb["index2"].tolist()))
    rec = dict(seed=int(seed), identical_list=A == B, n_differing=len(A ^ B), n_matches=len(A))
    if A != B:
        rec["flips"] = tie_evidence(ca, sorted(A - B), sorted(B - A))
    if ra.get("ok") and rb.get("ok"):
        rec["flow_delta"] = float(np.abs(ra["flow12"] - rb["flow12"]).max())
        rec["H_delta"] = float(np.abs(ra["H"] - rb["H"]).max())
    return rec


def _stab_job(args):
    t0 = time.perf_counter()
    try:
        rec = oracle_stability(*args)
    except Exception as e:  # noqa: BLE001
        rec = dict(seed=int(args[1]), error="%s: %s" % (type(e).__name__, e))
    rec["oracle_s"] = round(time.perf_counter() - t0, 2)
    return rec


def summarise_stability(cfg_name, records, H, W, ta, tb):
    done = [r for r in records if "error" not in r]
    flips = [f for r in done for f in r.get("flips", [])]
    return dict(kind="oracle_vs_oracle", oracle=oracle_name(), config=cfg_name, size="%dx%d" % (H, W),
                settings=["%d threads, oneDNN on" % ta, "%d thread(s), oneDNN off" % tb], pairs=len(done),
                errors=[r for r in records if "error" in r],
                pairs_with_flips=sum(1 for r in done if not r["identical_list"]),
                total_flipped_matches=sum(r["n_differing"] for r in done), total_matches=sum(r["n_matches"] for r in done),
                max_tie_evidence=max([f["evidence"] for f in flips], default=None),
                max_flow_delta_identical=max([r["flow_delta"] for r in done if r["identical_list"] and "flow_delta" in r], default=None),
                max_flow_delta_with_flips=max([r["flow_delta"] for r in done if not r["identical_list"] and "flow_delta" in r], default=None))


def _job(args):
    cfg_name, seed, H, W, path = args
    t0 = time.perf_counter()
    try:
        rec = compare_loop(cfg_name, seed, path) if "loop" in CONFIGS[cfg_name] else compare(cfg_name, seed, H, W, path)
    except Exception as e:  # noqa: BLE001 -- a checker crash must surface in the summary, not kill the sweep
        import traceback
        rec = dict(seed=int(seed), error="%s: %s" % (type(e).__name__, e), trace=traceback.format_exc()[-800:])
    rec["oracle_s"] = round(time.perf_counter() - t0, 2)
    return rec


def summarise(cfg_name, records, n_requested, elapsed, H, W):
    done = [r for r in records if "error" not in r]
    ident = [r for r in done if r["identical_list"]]
    diff = [r for r in done if not r["identical_list"]]
    flips = [f for r in diff for f in r["flips"]]
    both_ok = [r for r in done if "max_abs_flow_delta_e2e" in r]
    s = dict(config=cfg_name, oracle=(done[0].get("oracle") if done else oracle_name()), size="%dx%d" % (H, W), pairs=len(done),
             pairs_requested=n_requested,
             errors=[r for r in records if "error" in r],
             identical_lists=len(ident),
             inlier_exact=sum(1 for r in ident if r.get("inlier_indices_bit_exact")),
             inlier_compared=sum(1 for r in ident if "inlier_indices_bit_exact" in r),
             inlier_indices_bit_exact=(all(r.get("inlier_indices_bit_exact", True) for r in ident) if ident else None),
             max_abs_H_delta_identical=max([r["max_abs_H_delta"] for r in ident if "max_abs_H_delta" in r], default=None),
             max_flow_delta_e2e_identical=max([r["max_abs_flow_delta_e2e"] for r in ident if "max_abs_flow_delta_e2e" in r], default=None),
             # maximum over ALL compared pairs -- incl. the pairs where a near-tie flip made the two sides run different RANSACs
             max_flow_delta_e2e=max([r["max_abs_flow_delta_e2e"] for r in both_ok], default=None),
             pairs_with_flips=len(diff), total_flipped_matches=sum(r["n_differing"] for r in diff),
             total_matches=sum(r["n_matches_oracle"] for r in done),
             max_tie_evidence=max([f["evidence"] for f in flips], default=None),
             flips_all_near_ties=(all(f["evidence"] < TIE_EPS for f in flips) if flips else True), tie_eps=TIE_EPS,
             max_flow_delta_e2e_with_flips=max([r["max_abs_flow_delta_e2e"] for r in diff if "max_abs_flow_delta_e2e" in r], default=None),
             max_flow_delta_fine_stage_with_flips=max([r["max_abs_flow_delta_given_gpu_H"] for r in diff if "max_abs_flow_delta_given_gpu_H" in r], default=None),
             sentinel_agreement=all(r["oracle_ok"] == r["gpu_ok"] for r in ident),
             oracle_wall_s=round(elapsed, 1))
    # downstream of the arg-max: identical-list pairs are exact end to end; flipped pairs are checked on the device's list
    dn = [r for r in diff if r.get("downstream_ok")]
    exact = lambda r: r["downstream_inlier_bit_exact"] and r["downstream_H_delta"] <= 2e-6 and r["downstream_flow_delta"] < 1e-3
    ident_exact = [r for r in ident if r.get("inlier_indices_bit_exact") and r.get("max_abs_H_delta", 1) <= 2e-6
                   and r.get("max_abs_flow_delta_e2e", 1) < 1e-3]
    ident_sentinel = [r for r in ident if "max_abs_flow_delta_e2e" not in r and r["oracle_ok"] == r["gpu_ok"]]
    s["downstream_exact_given_matches"] = "%d/%d" % (len(ident_exact) + len(ident_sentinel) + sum(1 for r in dn if exact(r)), len(done))
    s["downstream_checked_flipped_pairs"] = len(dn)
    s["downstream_max_H_delta_flipped"] = max([r["downstream_H_delta"] for r in dn], default=None)
    s["downstream_max_flow_delta_flipped"] = max([r["downstream_flow_delta"] for r in dn], default=None)
    s["downstream_inlier_bit_exact_flipped"] = all(r["downstream_inlier_bit_exact"] for r in dn) if dn else None
    if cfg_name == "ev" and both_ok:
        s["max_match_delta_identical"] = max([r["max_abs_match_delta"] for r in ident if "max_abs_match_delta" in r], default=None)
        s["max_inbounds_mismatch_pixels_identical"] = max([r.get("inbounds_mismatch_pixels", 0) for r in ident], default=None)
    return s


def sweep(cfg_name, dump_dir, seeds, H, W, workers=None, threads=8, budget_s=None):
    """Oracle + comparison for every dumped pair, ``workers`` processes x ``threads`` torch threads, optionally bounded
    by a wall-clock budget (pairs not reached are reported, not silently dropped).  Returns (summary, per-pair records)."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads, cores))
    # one OpenMP thread per PHYSICAL core: on the GPU box (2 x 64 cores, SMT: 256 logical CPUs) 32 workers x 8 spinning threads
    # ran each oracle pair 4x slower than 2 workers x 4 threads do on an 8-core host
    phys = cores // 2 if cores >= 64 else cores
    workers = workers or max(1, phys // threads)
    jobs = [(cfg_name, s, H, W, os.path.join(dump_dir, "pair_%d.npz" % s)) for s in seeds]
    loop = "loop" in CONFIGS[cfg_name]
    mstd = MULTIH_MATCH_STD if loop else None
    t0 = time.perf_counter()
    records = []
    if workers == 1:
        _worker_init(threads, mstd)
        for j in jobs:
            if budget_s and time.perf_counter() - t0 > budget_s:
                break
            records.append(_job(j))
    else:
        ctx = mp.get_context("spawn")
        with ctx.Pool(workers, initializer=_worker_init, initargs=(threads, mstd)) as pool:
            it = pool.imap_unordered(_job, jobs)
            for _ in jobs:
                left = None if not budget_s else max(1.0, budget_s - (time.perf_counter() - t0))
                try:
                    records.append(it.next(timeout=left))
                except mp.TimeoutError:
                    break
            pool.terminate()
    records.sort(key=lambda r: r["seed"])
    summary = (summarise_loop(cfg_name, records, len(jobs), time.perf_counter() - t0) if loop else
               summarise(cfg_name, records, len(jobs), time.perf_counter() - t0, H, W))
    try:                                    # what the device results were computed with (dump_gpu_*: write_meta)
        summary["device"] = json.load(open(os.path.join(dump_dir, "meta.json")))
    except (OSError, ValueError):
        summary["device"] = None
    return summary, records


def stability_sweep(cfg_name, seeds, H, W, threads_a, threads_b, budget_s=None, workers=None):
    """Oracle vs oracle over ``seeds`` (worker processes like sweep(); pairs not reached within the budget are reported)."""
    import multiprocessing as mp
    t0 = time.perf_counter()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    phys = cores // 2 if cores >= 64 else cores
    workers = workers or max(1, phys // max(threads_a, 1))
    jobs = [(cfg_name, s, H, W, threads_a, threads_b) for s in seeds]
    records = []
    if workers == 1:
        _W["sds"] = state_dicts()
        for j in jobs:
            if budget_s and time.perf_counter() - t0 > budget_s:
                break
            records.append(_stab_job(j))
    else:
        ctx = mp.get_context("spawn")
        with ctx.Pool(workers, initializer=_worker_init, initargs=(threads_a, None)) as pool:
            it = pool.imap_unordered(_stab_job, jobs)
            for _ in jobs:
                left = None if not budget_s else max(1.0, budget_s - (time.perf_counter() - t0))
                try:
                    records.append(it.next(timeout=left))
                except mp.TimeoutError:
                    break
            pool.terminate()
    records.sort(key=lambda r: r["seed"])
    s = summarise_stability(cfg_name, records, H, W, threads_a, threads_b)
    s.update(pairs_requested=len(jobs), oracle_wall_s=round(time.perf_counter() - t0, 1))
    return s, records


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="qs", choices=sorted(CONFIGS))
    ap.add_argument("--dump", default=None)
    ap.add_argument("--stability", action="store_true", help="oracle vs oracle (thread count / oneDNN), no GPU dump needed")
    ap.add_argument("--threads-b", type=int, default=1)
    ap.add_argument("--seeds", type=int, nargs="+", default=list(range(64)))
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--budget", type=float, default=None)
    ap.add_argument("--records", type=str, default=None, help="write the per-pair records to this JSON file")
    ap.add_argument("--resummarise", type=str, default=None, help="recompute the summary of a saved records file (no oracle run)")
    a = ap.parse_args()
    if a.resummarise:
        d = json.load(open(a.resummarise))
        loop = "loop" in CONFIGS[a.config]
        old = d["summary"]
        d["summary"] = (summarise_loop(a.config, d["records"], old["pairs_requested"], old["oracle_wall_s"]) if loop else
                        summarise(a.config, d["records"], old["pairs_requested"], old["oracle_wall_s"], a.height or 480, a.width or 640))
        json.dump(d, open(a.resummarise, "w"), indent=1)
        print(json.dumps(d["summary"]))
        return
    a.height = a.height or CONFIGS[a.config].get("H", 480)
    a.width = a.width or CONFIGS[a.config].get("W", 640)
    if a.stability:
        summary, records = stability_sweep(a.config, a.seeds, a.height, a.width, a.threads, a.threads_b, a.budget, a.workers)
    else:
        if not a.dump:
            ap.error("--dump is required unless --stability")
        summary, records = sweep(a.config, a.dump, a.seeds, a.height, a.width, a.workers, a.threads, a.budget)
    if a.records:
        json.dump(dict(summary=summary, records=records), open(a.records, "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()

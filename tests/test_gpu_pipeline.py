"""End-to-end parity of the HIP pipeline against the reference's golden outputs (BASELINE config 1:
quick_start path on one 240x320 synthetic pair, nbIter=100) and against the CPU oracle run on this host
(config 2 shape, reduced), plus size-independent properties at the full 480x640 size.  ``-m gpu``."""
import os

import numpy as np
import pytest
import torch

import restate
from rfx import weights, synth, ops
from rfx.pipeline import AlignPipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sds():
    return dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1),
                flow=weights.net_flow_coarse_sd(2), match=weights.net_matchability_sd(3))


def _match_sets(i1a, i2a, i1b, i2b):
    a, b = set(zip(i1a.tolist(), i2a.tolist())), set(zip(i1b.tolist(), i2b.tolist()))
    return len(a ^ b), len(a)


def test_config1_against_reference_golden(dev):
    """The reference itself (CPU, /root/reference) produced tests/golden/config1.npz."""
    g = np.load(os.path.join(GOLD, "config1.npz"))
    I1, I2 = synth.make_pair(240, 320, seed=0)
    pipe = AlignPipeline(_sds(), nbScale=7, nbIter=100, tolerance=0.05, minSize=320, scaleR=1.2, device=dev, draw="host")
    prep = pipe.prepare([(I1, I2)])
    feats = pipe.features(prep)
    assert feats["nA"] == 2107 and feats["nB"] == 300
    ft = feats["featB"].view(1, 1024, 15, 20)[0, ::64].cpu().numpy()
    assert np.abs(ft - g["feat_t_sub"]).max() < 2e-5                                  # trunk + L2 norm
    torch.manual_seed(123)                                                            # same CPU index draw as the reference run
    r = pipe.align_prepared(prep, fine=True)[0]
    ndiff, nref = _match_sets(r["index1"].cpu().numpy(), r["index2"].cpu().numpy(), g["index1"], g["index2"])
    print("config1: %d matches, %d differ from the reference" % (nref, ndiff))
    # no escape hatch: on this pair the device's match list IS the reference's (a flip here is a regression to look at;
    # full-size pairs, where float near-ties do flip matches, are covered by tests/test_gpu_parity_sweep.py with a proof
    # obligation per flipped match)
    assert ndiff == 0
    assert np.array_equal(r["index1"].cpu().numpy(), g["index1"]) and np.array_equal(r["index2"].cpu().numpy(), g["index2"])
    # identical match list + identical index draw -> bit-exact inliers, H to float32 round-off
    assert r["status"] == 0
    inl = r["inlier"].cpu().numpy()
    i2 = r["index2"].cpu().numpy()[inl]
    mask = np.zeros((15, 20), dtype=np.float32)
    mask[i2 // 20, i2 % 20] = 1
    assert np.array_equal(mask, g["inlierMask"])
    assert np.abs(r["H"].cpu().numpy() - g["H"]).max() <= 1e-6
    assert np.abs(r["flowDown"].cpu().numpy() - g["flowDown"]).max() < 1e-4
    d = np.abs(r["flow12"][:, ::8, ::8].cpu().numpy() - g["flow12_sub"]).max()
    print("config1: max-abs flow delta vs reference = %.3e" % d)
    assert d < 1e-3                                                                   # north-star tolerance


def test_fine_stage_against_oracle_given_same_homography(dev):
    """Flow parity isolated from the match list: feed the oracle's H to both fine stages (320x240)."""
    I1, I2 = synth.make_pair(240, 320, seed=3)
    sds = _sds()
    pipe = AlignPipeline(sds, nbScale=3, nbIter=100, tolerance=0.05, minSize=320, scaleR=1.2, device=dev)
    prep = pipe.prepare([(I1, I2), (I2, I1)])
    Hs = torch.tensor([[[1.01, 0.02, 0.03], [-0.015, 0.99, -0.02], [0.01, 0.005, 1.0]],
                       [[0.98, -0.01, -0.04], [0.02, 1.02, 0.01], [-0.004, 0.01, 1.0]]])
    f = pipe.fine_quickstart(prep, Hs.to(dev))
    for b in range(2):
        with torch.no_grad():
            fc = restate.warp_grid(Hs[b:b + 1], 240, 320)
            st = restate.fine_step_quickstart(dict(feat=sds["feat"], flow=sds["flow"]), prep["IsTensor"][b:b + 1].cpu(),
                                              prep["ItTensor"][b:b + 1].cpu(), fc)
        assert (f["corr12"][b].cpu() - st["corr12"][0]).abs().max() < 2e-5
        assert (f["flowDown"][b].cpu() - st["flowDown"][0]).abs().max() < 1e-4
        d = (f["flow12"][b].cpu() - st["flow12"][0]).abs().max().item()
        assert d < 1e-3, d
        assert (f["img1_fine"][b].cpu() - st["img1_fine"][0]).abs().max() < 2e-3
    # PredFlowMask (evalHpatch variant) on the same inputs
    featt = None
    from rfx import ops
    featt = ops.l2norm(pipe.feat(prep["ItTensor"]))
    fc_d = ops.warp_grid(Hs.to(dev), 240, 320)
    pm = pipe.pred_flow_mask(prep["IsTensor"], featt, fc_d)
    with torch.no_grad():
        import torch.nn.functional as F
        ft = F.normalize(restate.feature_extractor(sds["feat"], prep["ItTensor"][:1].cpu()))
        flow12, match, fd8, md8 = restate.pred_flow_mask(dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]),
                                                         prep["IsTensor"][:1].cpu(), ft, restate.warp_grid(Hs[:1], 240, 320),
                                                         restate.identity_grid(240, 320))
    assert (pm["flow12"][0].cpu() - flow12[0]).abs().max() < 1e-3
    assert np.abs(pm["match"][0, 0].cpu().numpy() - match).max() < 1e-3
    assert np.abs(pm["match21Down8"][0].cpu().numpy() - md8[0, 1:2]).max() < 1e-4
    # KITTI variant (evaluation/evalKITTI/evaluation.py:49-81): cycle-checked matchability
    IsSample = ops.grid_sample(prep["IsTensor"], fc_d)
    pk = pipe.pred_flow_mask_kitti(IsSample, prep["ItTensor"], fc_d)
    with torch.no_grad():
        f12k, mk, fdk, mdk = restate.pred_flow_mask_kitti(dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]),
                                                          IsSample[:1].cpu(), prep["ItTensor"][:1].cpu(),
                                                          restate.warp_grid(Hs[:1], 240, 320), restate.identity_grid(240, 320))
    assert (pk["flow12"][0].cpu() - f12k[0]).abs().max() < 1e-3
    assert np.abs(pk["match"][0, 0].cpu().numpy() - mk).max() < 1e-3


def test_config2_coarse_480x640_properties(dev):
    """BASELINE config 2 (one 480x640 pair, coarse only, nbIter=1000): shapes, determinism, known answer.
    The synthetic target is the source shifted by (12, 8) px, so the recovered homography must be close to
    that translation in normalised coordinates and explain most matches."""
    I1, I2 = synth.make_pair(480, 640, seed=0)
    pipe = AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=7, nbIter=1000, tolerance=0.05, minSize=640,
                         scaleR=1.2, device=dev, draw="host")
    prep = pipe.prepare([(I1, I2)])
    feats = pipe.features(prep)
    assert feats["nA"] == 8531 and feats["nB"] == 1200
    torch.manual_seed(123)
    r1 = pipe.coarse(prep, feats=feats)[0]
    torch.manual_seed(123)
    r2 = pipe.coarse(prep, feats=feats)[0]
    assert r1["status"] == 0
    assert torch.equal(r1["index1"], r2["index1"]) and torch.equal(r1["inlier"], r2["inlier"])   # run-to-run determinism
    assert torch.equal(r1["H"], r2["H"])
    H = r1["H"].cpu().numpy()
    H = H / H[2, 2]
    assert abs(H[0, 0] - 1) < 0.05 and abs(H[1, 1] - 1) < 0.05
    assert abs(H[0, 2] - 2 * 12 / 640) < 0.02 and abs(H[1, 2] - 2 * 8 / 480) < 0.02
    assert r1["count"] >= 0.5 * r1["n"]
    # oracle RANSAC on the device's match list with the same index draw: bit-exact inliers
    Hb, cnt, inl, _ = restate.ransac(r1["match1"].cpu(), r1["match2"].cpu(), 0.05, r1["samples"])
    assert int(cnt) == r1["count"] and np.array_equal(inl, r1["inlier"].cpu().numpy())
    assert np.abs(Hb - r1["H"].cpu().numpy()).max() <= 1.2e-7


def test_multi_homography_driver_matches_oracle(dev):
    """Device-resident multi-H loop (pipeline.multi_h) vs the oracle's restatement of
    evaluation/evalHpatch/evaluation.py:184-243 on the same pair and the same index draws."""
    I1, I2 = synth.make_pair(240, 320, seed=9)
    sds = _sds()
    sds["match"] = weights.net_matchability_sd(3, last_std=0.02)
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev, draw="host")
    prep = pipe.prepare([(I1, I2)])
    torch.manual_seed(11)
    out = pipe.multi_h(prep, 0, maxCoarse=2, maskRegionTh=0.01)
    ca = restate.CoarseAlignOracle(sds["trunk"], 3, 300, 0.05, 240, 1.2, variant="B")
    ca.setPair(I1, I2)
    torch.manual_seed(11)
    o = restate.multi_h_loop(ca, dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]), max_coarse=2,
                             mask_region_th=0.01)
    assert len(out["H"]) == len(o["H"]) >= 1
    assert np.abs(out["H"][0].cpu().numpy() - o["H"][0]).max() < 1e-5
    assert np.abs(out["flowDown8"][0].cpu().numpy() - o["flowDown8"][0]).max() < 1e-3
    assert np.abs(out["matchDown8"][0].cpu().numpy() - o["matchDown8"][0]).max() < 1e-3
    # explained-region mask: thresholded matchability -> identical except where the value sits at the threshold
    diff = (out["mask"].cpu().numpy() != o["masks"][-1]).mean()
    assert diff < 1e-3


@pytest.mark.parametrize("cfg", ["config4", "config5"])
def test_large_configs_shapes_and_invariants(dev, cfg):
    """BASELINE configs 4 (960x720, 5 scales x2, minSize 720, variant B) and 5 (KITTI 1242x376, 3 scales x1.2,
    coarseSize 800): the cell counts of SURVEY Appendix C, a valid run of the coarse stage at those sizes (the
    850 MB score matrix of config 5 is never materialised), run-to-run determinism and oracle-exact RANSAC on the
    device's own match list."""
    if cfg == "config4":
        H, W, nbScale, scaleR, minSize, nA, nB = 720, 960, 5, 2.0, 720, 21675, 2700
    else:
        H, W, nbScale, scaleR, minSize, nA, nB = 376, 1242, 3, 1.2, 800, 25747, 8250
    I1, I2 = synth.make_pair(H, W, seed=4)
    pipe = AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=nbScale, nbIter=500, tolerance=0.05,
                         minSize=minSize, scaleR=scaleR, variant="B", device=dev, draw="host")
    prep = pipe.prepare([(I1, I2)])
    feats = pipe.features(prep)
    assert feats["nA"] == nA and feats["nB"] == nB
    torch.manual_seed(5)
    r1 = pipe.coarse(prep, feats=feats)[0]
    torch.manual_seed(5)
    r2 = pipe.coarse(prep, feats=feats)[0]
    assert r1["n"] >= 4 and r1["status"] == 0
    assert torch.equal(r1["index1"], r2["index1"]) and torch.equal(r1["index2"], r2["index2"])
    assert torch.equal(r1["inlier"], r2["inlier"]) and torch.equal(r1["H"], r2["H"])
    i1 = r1["index1"].cpu()
    assert (i1[1:] > i1[:-1]).all() and int(r1["index2"].max()) < nB          # ordered, in range, one match per source cell
    assert len(torch.unique(r1["index2"])) == r1["n"]                          # mutual => target cells unique too
    Hb, cnt, inl, _ = restate.ransac(r1["match1"].cpu(), r1["match2"].cpu(), 0.05, r1["samples"])
    assert int(cnt) == r1["count"] and np.array_equal(inl, r1["inlier"].cpu().numpy())


def test_failed_and_mixed_batches_keep_the_reference_sentinels(dev):
    """A batch holding an alignable pair, a pair of unrelated noise images and a pair whose target is blank: the
    pipeline must neither crash nor let one pair disturb another.  The good pair is bit-identical to the same pair run
    alone; whatever the hopeless pairs end as (a homography -- with a random-init trunk the zero-padding makes
    features position-dependent, so even noise "aligns" to the identity -- or the reference's None sentinel: RANSAC
    abort, utils/outil.py:145-146 / fewer than 4 matches, quick_start/coarseAlignFeatMatch.py:157-158), flows stay
    finite and the result records carry a failure flag exactly where H is None."""
    import PIL.Image as Image
    from rfx import dist as rdist
    rng = np.random.RandomState(3)
    good = synth.make_pair(128, 160, seed=5)
    noise = (Image.fromarray((rng.rand(128, 160, 3) * 255).astype(np.uint8)),
             Image.fromarray((rng.rand(128, 160, 3) * 255).astype(np.uint8)))
    blank = (good[0], Image.fromarray(np.full((128, 160, 3), 127, dtype=np.uint8)))
    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3))
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.01, minSize=160, scaleR=1.2, device=dev, draw="host")
    torch.manual_seed(7)
    alone = pipe.align_pairs([good])[0]
    torch.manual_seed(7)
    res = pipe.align_pairs([good, noise, blank])
    assert len(res) == 3
    assert res[0]["H"] is not None and torch.equal(res[0]["H"], alone["H"]) and torch.equal(res[0]["flow12"], alone["flow12"])
    rec = rdist.pack_records(res)
    assert rec.shape[0] == 3 and rec[0, 9].item() == 0.0
    for b, r in enumerate(res):
        assert torch.isfinite(r["flow12"]).all()
        assert (r["H"] is None) == (rec[b, 9].item() == 1.0)
        if r["H"] is None:
            assert float(rec[b, :9].abs().max()) == 0.0 and (r["n"] < 4 or r["status"] in (1, 2))
        else:
            assert r["status"] == 0 and r["count"] >= 4 and int(r["inlier"].sum()) == r["count"]



def test_batched_multi_homography_equals_per_pair_driver(dev):
    """pipeline.multi_h_batched (all pairs in lock-step) against pipeline.multi_h (one pair at a time) with the same index
    draws per (pair, homography): same number of homographies, same H / flows / masks."""
    sds = _sds()
    sds["match"] = weights.net_matchability_sd(3, last_std=0.02)
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev)
    pairs = [synth.make_pair(240, 320, seed=s) for s in (9, 10, 11)]
    prep = pipe.prepare(pairs)
    feats = pipe.features(prep)

    def draws(b):
        state = {"k": 0}
        def fn(n, it):
            g = torch.Generator().manual_seed(1000 * b + state["k"])
            state["k"] += 1
            return torch.randint(n, (it, 4), generator=g)
        return fn
    single = [pipe.multi_h(prep, b, maxCoarse=2, maskRegionTh=0.01, feats=feats, sample_fn=draws(b)) for b in range(3)]
    per_pair = [draws(b) for b in range(3)]
    batched = pipe.multi_h_batched(prep, maxCoarse=2, maskRegionTh=0.01, feats=feats,
                                   sample_fn=lambda b, n, it: per_pair[b](n, it))
    for b in range(3):
        s, m = single[b], batched[b]
        assert len(s["H"]) == len(m["H"]) >= 1, (b, len(s["H"]), len(m["H"]))
        for k in range(len(s["H"])):
            assert (s["H"][k] - m["H"][k]).abs().max() < 1e-6
            assert (s["flowDown8"][k] - m["flowDown8"][k]).abs().max() < 1e-4
            assert (s["matchDown8"][k] - m["matchDown8"][k]).abs().max() < 1e-4
        assert float((s["mask"] != m["mask"]).float().mean()) < 1e-3


@pytest.mark.parametrize("tag", ["a", "b"])
def test_kitti_two_resolution_driver_matches_reference_golden(dev, tag):
    """pipeline.multi_h_kitti (SURVEY 8f1 / BASELINE config 5) against tests/golden/kitti_loop.npz -- the outputs of the
    reference's own ``while True`` loop (evaluation/evalKITTI/evaluation.py:270-336) with its PredFlowMask / get_info /
    remove_small_cc / resizeImg and CoarseAlign, produced by tests/golden/make_golden.py -- and against the oracle
    restatement run here with the same index draws.  Half-resolution features are 8 x 26 / 7 x 23: W % 4 != 0 exercises
    the plain correlation fallback; the cc-filter (case b) runs on the host as in the reference."""
    g = np.load(os.path.join(GOLD, "kitti_loop.npz"))
    seed, fine, cc_th, th, draw_seed = g["%s_cfg" % tag]
    sds = _sds()
    sds["match"] = weights.net_matchability_sd(3, last_std=float(g["match_std"]))
    Is, It = synth.make_pair(96, 312, seed=int(seed), homography=True, amp=0.03)
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=160, scaleR=1.2, variant="B", device=dev, draw="host")
    raw = pipe.upload_raw([(Is, It)])
    h_org, w_org, h_r, w_r, h_d2, w_d2 = (int(x) for x in g["%s_sizes" % tag])
    assert pipe.resize_img_dims(w_org, h_org, 8, int(fine)) == (w_r, h_r)
    assert pipe.resize_img_dims(w_org, h_org, 8, int(fine) // 2) == (w_d2, h_d2)
    torch.manual_seed(int(draw_seed))
    out = pipe.multi_h_kitti(raw[0], raw[1], fineSize=int(fine), maskRegionTh=float(th), cc_th=float(cc_th))
    nb = int(g["%s_nb" % tag])
    assert len(out["H"]) == nb >= 2
    Hs = torch.stack(out["H"]).cpu().numpy()
    assert np.abs(Hs[0] - g["%s_H" % tag][0]).max() < 1e-5                      # same matches, same draw -> same first H
    assert np.abs(torch.cat(out["flowD2"]).cpu().numpy()[0] - g["%s_flowD2" % tag][0]).max() < 1e-3
    assert np.abs(torch.cat(out["flowDown8"]).cpu().numpy()[0] - g["%s_flowDown8" % tag][0]).max() < 1e-3
    # free-running loop: the final explained-region mask (later rounds depend on maps thresholded at 0.9999 -- a saturating
    # sigmoid -- so a pixel at the threshold may differ; every round is compared UNCONDITIONALLY below, teacher-forced)
    assert float((out["mask"].cpu().numpy() != g["%s_mask" % tag]).mean()) < 5e-3
    _teacher_forced_rounds(pipe, g, tag, dev, kitti=dict(raw=raw, fine=int(fine), cc_th=float(cc_th)), th=float(th),
                           draw_seed=int(draw_seed))
    # the oracle restatement on this host agrees with the golden too (same code path as tests/test_oracle_pins.py)
    ca = restate.CoarseAlignOracle(sds["trunk"], 3, 300, 0.05, 160, 1.2, variant="B")
    ca.setPair(Is, It)
    torch.manual_seed(int(draw_seed))
    o = restate.multi_h_loop_kitti(ca, dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]), Is, It, int(fine),
                                   mask_region_th=float(th), cc_th=float(cc_th))
    assert len(o["H"]) == nb and np.abs(np.stack(o["H"]) - g["%s_H" % tag]).max() < 1e-5


def test_kitti_lock_step_driver_equals_the_per_pair_driver(dev):
    """pipeline.multi_h_kitti_batched (throughput form of the KITTI driver: the k-th homography of all active pairs in one
    launch chain) == pipeline.multi_h_kitti pair by pair, given the same index draws: same number of homographies per
    pair, same H, flows and masks up to batch-size-dependent float32 round-off (none: the kernels are batch-invariant)."""
    sds = _sds()
    sds["match"] = weights.net_matchability_sd(3, last_std=3.0)
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=160, scaleR=1.2, variant="B", device=dev)
    pairs = [synth.make_pair(96, 312, seed=s, homography=True, amp=0.03) for s in (11, 12, 13)]
    raw = pipe.upload_raw(pairs)

    def draws(b, k, n, it):
        return torch.randint(n, (it, 4), generator=torch.Generator().manual_seed(1000 * b + k))
    single = []
    for b in range(len(pairs)):
        calls = [0]

        def fn(n, it, b=b, calls=calls):
            calls[0] += 1
            return draws(b, calls[0], n, it)
        single.append(pipe.multi_h_kitti(raw[0][b:b + 1], raw[1][b:b + 1], fineSize=200, maskRegionTh=0.005, cc_th=0.01, sample_fn=fn))
    ncall = [0] * len(pairs)

    def fnb(b, n, it):
        ncall[b] += 1
        return draws(b, ncall[b], n, it)
    batched = pipe.multi_h_kitti_batched(raw[0], raw[1], fineSize=200, maskRegionTh=0.005, cc_th=0.01, sample_fn=fnb)
    assert max(len(o["H"]) for o in single) >= 2
    for b, (s1, m) in enumerate(zip(single, batched)):
        assert len(s1["H"]) == len(m["H"]), (b, len(s1["H"]), len(m["H"]))
        for k in range(len(s1["H"])):
            assert (s1["H"][k] - m["H"][k]).abs().max() < 1e-6
            assert (s1["flowD2"][k] - m["flowD2"][k]).abs().max() < 1e-5
            assert (s1["flowDown8"][k] - m["flowDown8"][k]).abs().max() < 1e-5
            assert (s1["matchDown8"][k] - m["matchDown8"][k]).abs().max() < 1e-5
        assert float((s1["mask"] != m["mask"]).float().mean()) < 1e-4


def test_kitti_lock_step_groups_on_streams_give_the_one_group_records_bit_for_bit(dev):
    """multi_h_kitti_batched(split=k): the KITTI rounds run on the Hpatch driver's round generator (lock-step groups on HIP
    streams, ready-first scheduling, the exact mode's host stage under the other groups' kernels).  Device draws are keyed by
    pair position / id, every kernel computes a pair independently: records, masks and lists equal the one-group run's bit for
    bit, in the exact mode and the device-null-vector mode; the capacity stop (status 4) too."""
    sds = _sds()
    sds["match"] = weights.net_matchability_sd(3, last_std=3.0)
    pairs = [synth.make_pair(96, 312, seed=s, homography=True, amp=0.03) for s in (11, 12, 13, 14, 15)]
    for degen in ("lapack", "device"):
        pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=160, scaleR=1.2, variant="B", device=dev, seed=7,
                             degenerate=degen)
        raw = pipe.upload_raw(pairs)
        w_r, h_r = pipe.resize_img_dims(312, 96, 8, 200)
        w_d2, h_d2 = pipe.resize_img_dims(312, 96, 8, 100)

        def run(split, max_h=8):
            pipe.reseed(7)
            R = ops.MultiHRecords(len(pairs), h_r // 8, w_r // 8, dev, max_h=max_h, hd2=h_d2 // 8, wd2=w_d2 // 8)
            outs = pipe.multi_h_kitti_batched(raw[0], raw[1], fineSize=200, maskRegionTh=0.005, cc_th=0.01, records=R, split=split)
            torch.cuda.synchronize()
            return R, outs
        R1, outs1 = run(1)
        assert int(R1.rec[:, 0].sum()) >= len(pairs) + 1
        for k in (2, 3):
            Rk, outs = run(k)
            assert torch.equal(Rk.rec, R1.rec), (degen, k)
            for a, b in zip(outs, outs1):
                assert a["nbH"] == b["nbH"] and torch.equal(a["mask"], b["mask"])
                assert all(torch.equal(x, y) for x, y in zip(a["H"], b["H"]))
                assert all(torch.equal(x, y) for x, y in zip(a["flowD2"], b["flowD2"]))
        Rc1, _ = run(1, max_h=1)
        Rc2, _ = run(2, max_h=1)
        assert torch.equal(Rc1.rec, Rc2.rec) and bool((Rc1.rec[:, 1] == 4.0).any())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_multi_homography_driver_matches_reference_golden(dev, tag):
    """pipeline.multi_h / multi_h_batched against tests/golden/multi_h.npz: the outputs of the reference's own
    ``while nbCoarse <= args.maxCoarse`` loop (evaluation/evalHpatch/evaluation.py:211-243) with its CoarseAlign-B and
    PredFlowMask, on a homography-warped pair with a saturating matchability head (the explained-region mask grows)."""
    g = np.load(os.path.join(GOLD, "multi_h.npz"))
    seed, maxCoarse, th, draw_seed = g["%s_cfg" % tag]
    sds = _sds()
    sds["match"] = weights.net_matchability_sd(3, last_std=float(g["match_std"]))
    I1, I2 = synth.make_pair(240, 320, seed=int(seed), homography=True)
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev, draw="host")
    prep = pipe.prepare([(I1, I2)])
    feats = pipe.features(prep)
    nb = int(g["%s_nb" % tag])
    for driver in ("single", "batched"):
        torch.manual_seed(int(draw_seed))
        if driver == "single":
            out = pipe.multi_h(prep, 0, maxCoarse=int(maxCoarse), maskRegionTh=float(th), feats=feats)
        else:
            out = pipe.multi_h_batched(prep, maxCoarse=int(maxCoarse), maskRegionTh=float(th), feats=feats)[0]
        assert len(out["H"]) == nb, (driver, len(out["H"]), nb)
        Hs = torch.stack(out["H"]).cpu().numpy()
        assert np.abs(Hs[0] - g["%s_H" % tag][0]).max() < 1e-5
        assert np.abs(out["flowDown8"][0].cpu().numpy()[0] - g["%s_flowDown8" % tag][0]).max() < 1e-3
        assert float((out["mask"].cpu().numpy() != g["%s_mask" % tag]).mean()) < 5e-3
    # every round against the reference's own round, from the reference's own state (no "if the Hs happen to agree")
    _teacher_forced_rounds(pipe, g, tag, dev, prep=prep, feats=feats, th=float(th), draw_seed=int(draw_seed))


def _teacher_forced_rounds(pipe, g, tag, dev, th, draw_seed, prep=None, feats=None, kitti=None):
    """Round k of the device path from the REFERENCE's state at round k (tests/golden/make_golden.py records, for every
    round of the reference's own loop, the mask getCoarse received, the number of matches RANSAC saw, the H it returned and
    the matchability map the accept test read).  Unconditional per round: same surviving-match count, H within 1e-5, /8
    flows within 1e-3, same accept decision, and a next mask that differs from the reference's only at pixels whose
    matchability sits at the saturation threshold on BOTH sides."""
    from rfx import ops
    fgs = np.unpackbits(g["%s_round_fg" % tag], axis=-1).astype(np.float32)
    ns, Hs, matches, nb = g["%s_round_n" % tag], g["%s_round_H" % tag], g["%s_round_match" % tag], int(g["%s_nb" % tag])
    if kitti is not None:
        src_u8, tgt_u8 = kitti["raw"]
        prep = pipe.prepare_device(src_u8, tgt_u8)
        feats = pipe.features(prep)
        h, w = tgt_u8.shape[1], tgt_u8.shape[2]
        w_r, h_r = pipe.resize_img_dims(w, h, 8, kitti["fine"])
        w_d2, h_d2 = pipe.resize_img_dims(w, h, 8, kitti["fine"] // 2)
        tensor_s, _ = ops.u8_to_f32(src_u8)
        tensor_resize, _ = ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, w_r, h_r))
        tensor_d2, _ = ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, w_d2, h_d2))
    else:
        h, w = prep["ItTensor"].shape[2], prep["ItTensor"].shape[3]
        featt = ops.l2norm(pipe.feat(prep["ItTensor"]))
    fgs = fgs[:, :, :w]
    idx1, idx2, cnt = pipe._mutual_batched(feats, 1)
    torch.manual_seed(draw_seed)                      # the reference drew round k's indices from this generator, in order
    rounds = len(ns)
    assert rounds >= nb >= 2
    for k in range(rounds):
        Mask = torch.from_numpy(fgs[k:k + 1].copy()).to(dev)           # bg = 1 -> fg IS the mask
        M1, M2, n_dev = ops.filter_matches(idx1, idx2, cnt, None, Mask, None, feats["rt"], feats["ct"], feats["HA"], feats["WA"],
                                           feats["Ht"], feats["Wt"])
        n = int(n_dev.item())
        if ns[k] < 0:                                                   # the reference returned None before RANSAC (:165-166)
            assert n < 4
            break
        assert n == int(ns[k]), (k, n, int(ns[k]))
        smp = torch.randint(n, (pipe.nbIter, 4))
        bestH, _, res = ops.ransac_h4_batched(M1, M2, n_dev, smp[None].to(dev), pipe.tol)
        assert int(res[0, 0]) == 0
        assert np.abs(bestH[0].cpu().numpy() - Hs[k]).max() < 1e-5, (k, bestH[0].cpu().numpy(), Hs[k])
        if kitti is not None:
            flow_d2, pm, match = pipe.kitti_fine_round(bestH, tensor_s, tensor_d2, tensor_resize, (h, w), kitti["cc_th"])
            mode = 1
        else:
            pm = pipe.pred_flow_mask(prep["IsTensor"], featt, ops.warp_grid(bestH, h, w))
            match, mode, flow_d2 = pm["match"][:, 0], 0, None
        accepted = k < nb
        if accepted:
            assert np.abs(pm["flowDown8"][0].cpu().numpy() - g["%s_flowDown8" % tag][k]).max() < 1e-3
            md = torch.cat((pm["match12Down8"], pm["match21Down8"]), dim=1)[0].cpu().numpy()
            assert np.mean(np.abs(md - g["%s_matchDown8" % tag][k]) > 1e-3) < 2e-3     # sigmoid of std-3 logits amplifies round-off
            if kitti is not None:
                assert np.abs(flow_d2[0].cpu().numpy() - g["%s_flowD2" % tag][k]).max() < 1e-3
        mdev, mref = match[0].cpu().numpy(), matches[k]
        assert np.mean(np.abs(mdev - mref) > 1e-3) < 5e-3
        nbH = torch.full((1,), k, dtype=torch.int32, device=dev)
        acc, gain = ops.multih_accept(match, Mask, None, None, res, n_dev, nbH, th, mode)
        assert int(acc[0]) == int(accepted), (k, float(gain[0]))
        if accepted:
            assert int(nbH[0]) == k + 1
        if accepted and k + 1 < rounds:
            new_dev, new_ref = Mask[0].cpu().numpy(), fgs[k + 1]
            diff = new_dev != new_ref
            assert diff.mean() < 5e-3
            if kitti is None:
                # a differing pixel is a saturation-threshold pixel: the sigmoid is 1.0 on one side and 1 - ulp on the other
                assert diff.sum() == 0 or min(mdev[diff].min(), mref[diff].min()) > 1 - 1e-5


def test_rccl_transport_with_a_world_of_one_rank(dev):
    """The 1-GPU box cannot run N > 1 over RCCL, but it can run the RCCL path itself: bench.py with RFX_BENCH_FORCE_DIST=1
    initialises the "nccl" (= RCCL) process group for ONE rank and pushes the per-step result records through
    all_gather_into_tensor and the elapsed time through all_reduce(MAX) -- library load, communicator init (with
    HSA_ENABLE_IPC_MODE_LEGACY=0) and both collectives on the device, as the driver's N > 1 runs will."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RFX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--no-config3-leg"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["gathered_records"] == 8 and j["config"]["aligned_ok_last_step"] == 8
    assert "nccl" in j["config"]["collective"]


def test_two_rccl_ranks_give_every_pair_the_one_rank_record(dev, tmp_path):
    """the first REAL N > 1 execution, wherever the box exposes >= 2 GPUs (skipped on the 1-GPU boxes of this pool):
    ``bench.py --gpus 2`` spawns two RCCL ranks (one per GPU), every rank aligns its shard (pair i -> rank i mod 2) and the per-step
    result records go through ONE pipelined all_gather over xGMI.  Checked: both ranks' records arrive (``ranks_seen_in_gather``), and
    the record of every pair -- homography count, homographies, /8 flows, matchability maps -- equals, bit for bit, the record the
    same pair gets in a 1-rank run of the same stream (device draws are keyed by the ABSOLUTE pair id, the exact mode's LAPACK
    stage is per pair: nothing depends on the sharding)."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device: profiles/r03_bench_2ranks_on_1gpu_rccl_refused.log)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-exact-leg", "--no-qs-leg", "--score-chunk", "256"]
    two, one = str(tmp_path / "two.pt"), str(tmp_path / "one.pt")
    o2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "4", "--dump-records", two] + common,
                        capture_output=True, text=True, timeout=900, env=env)
    assert o2.returncode == 0, o2.stderr[-2000:]
    j2 = json.loads([ln for ln in o2.stdout.splitlines() if ln.startswith("{")][0])
    assert j2["n_gpus"] == 2 and j2["config"]["ranks_seen_in_gather"] == [0, 1] and j2["config"]["gathered_records"] == 8
    assert j2["config"]["n_ranks_in_rccl"] == 2 and "nccl" in j2["config"]["collective"]
    o1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "8", "--dump-records", one] + common,
                        capture_output=True, text=True, timeout=900, env=env)
    assert o1.returncode == 0, o1.stderr[-2000:]
    d2, d1 = torch.load(two), torch.load(one)
    assert sorted(d2["pair_ids"]) == list(range(8)) and d1["pair_ids"] == list(range(8))
    keep = [c for c in range(d1["records"].shape[1]) if c != 2]            # column 2 = the producer's rank
    for row, pid in enumerate(d2["pair_ids"]):
        assert torch.equal(d2["records"][row, keep], d1["records"][pid, keep]), pid
